#!/usr/bin/env python
"""Lab for csrc/gemm128x.hip (128 x 256 tiles, the previous tile's epilogue under the current tile's K loop; round 6).
For each case: the 256-row kernels (kbner_gemm_set_variant without bit 4) against the same launch with bit 4 set -- results must
be EQUAL (same MFMA order per element, same epilogue arithmetic) -- then TFLOP/s of both, alternating.
    python tools/gemm128x_lab.py [--sentences 256] [--reps 20] [--skip-bench] [--cases ffn_up,oproj,ffn_down_dgrad,qkv]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import labenv; labenv.apply()   # KBNER_LIB / KBNER_GEMM_VARIANT (lab switches live in tools/, not in the product binding)
import torch
from kbner import ops
from kbner import lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--sentences", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--skip-bench", action="store_true")
ap.add_argument("--cases", default="ffn_up")
ap.add_argument("--base", type=int, default=3, help="variant bits of the reference launch")
ap.add_argument("--bit", type=int, default=16, help="variant bit of the kernel under test: 16 = gemm128x (bit-identical), 32 = gemm128s (wave-specialised epilogue: the pre-activation is rounded to bf16 before GELU, compared with a tolerance)")
ap.add_argument("--tol", type=float, default=4e-3, help="--bit 32: relative-L2 tolerance of the outputs")
ap.add_argument("--variants", default="", help="extra kbner_gemm_set_variant values to time next to --base (e.g. 35,67,99: phased start)")
ap.add_argument("--strace", default="", help="gemm128s lab trace ids (7 idle epilogue, 8 full): cycles of wave 0 per K step and around the hand-off")
ap.add_argument("--trace", default="", help="lab library: ablation ids built with the cycle trace (11 full, 13 no stores, 14 neither): print per-segment cycles")
ap.add_argument("--abl", default="", help="lab library (KBNER_LIB=kb-ner_amd/kbner/_exp/libkbner_lab.so): ablation ids 1-8 to time next to the full kernel")
a = ap.parse_args()
dev = "cuda"
H, F, S = 1024, 4096, 512


def make(case, M, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    if case == "ffn_up":      # NT: act = gelu(x W1^T + b), out2 = gelu'
        A = (rn(M, H) * 0.7).to(torch.bfloat16); B = (rn(F, H) * 0.05).to(torch.bfloat16)
        kw = dict(layout=L.GEMM_NT, N=F, K=H, bias=rn(F) * 0.3, epi=L.EPI_BIAS | L.EPI_GELU, outs=("C", "out2"))
    elif case == "qkv":       # NT, bias only
        A = (rn(M, H) * 0.7).to(torch.bfloat16); B = (rn(3 * H, H) * 0.05).to(torch.bfloat16)
        kw = dict(layout=L.GEMM_NT, N=3 * H, K=H, bias=rn(3 * H) * 0.3, epi=L.EPI_BIAS, outs=("C",))
    elif case == "oproj":     # NT: h1 = ctx Wo^T + b + x
        A = (rn(M, H) * 0.7).to(torch.bfloat16); B = (rn(H, H) * 0.05).to(torch.bfloat16)
        kw = dict(layout=L.GEMM_NT, N=H, K=H, bias=rn(H) * 0.3, addend=(rn(M, H)).to(torch.bfloat16), epi=L.EPI_BIAS | L.EPI_ADD, outs=("C",))
    elif case == "ffn_down_dgrad":   # NN: dpre = (dh2 W2) * gelu', column sums -> b1 gradient
        A = (rn(M, H) * 0.7).to(torch.bfloat16); B = (rn(H, F) * 0.05).to(torch.bfloat16)
        kw = dict(layout=L.GEMM_NN, N=F, K=H, aux=(rn(M, F) * 0.5).to(torch.bfloat16),
                  epi=L.EPI_DGELU | L.EPI_COLSUM | L.EPI_COLSUM_WS, outs=("C", "colsum"))
    else:
        raise SystemExit("unknown case " + case)
    return A, B, kw


def run(case, M, A, B, kw, variant, bufs=None):
    ops.gemm_variant(variant)
    N, K = kw["N"], kw["K"]
    if bufs is None:
        bufs = {"C": torch.empty(M, N, dtype=torch.bfloat16, device=dev)}
        if "out2" in kw["outs"]:
            bufs["out2"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        if "colsum" in kw["outs"]:
            bufs["colsum"] = torch.empty(2 * (M // 128), N, dtype=torch.float32, device=dev)
    rows = ops.gemm(kw["layout"], A, B, M, N, K, C=bufs["C"], bias=kw.get("bias"), addend=kw.get("addend"), aux=kw.get("aux"),
                    out2=bufs.get("out2"), epi=kw["epi"], colsum=bufs.get("colsum"))
    return bufs, rows


def main():
    prev = ops.gemm_variant()
    M = a.sentences * S
    rc = 0
    try:
        for case in a.cases.split(","):
            for Mc in (M,):
                A, B, kw = make(case, Mc)
                r0, rows0 = run(case, Mc, A, B, kw, a.base)
                r1, rows1 = run(case, Mc, A, B, kw, a.base | a.bit)
                torch.cuda.synchronize()
                ok = True
                for k in kw["outs"]:
                    if k == "colsum":
                        n0_ = 2 * (Mc // rows0); n1_ = 2 * (Mc // rows1)
                        s0 = r0[k][:n0_].double().sum(0); s1 = r1[k][:n1_].double().sum(0)
                        same = float((s0 - s1).abs().max()) <= 1e-5 * float(s0.abs().max())
                    elif a.bit == 32:
                        # one bf16 rounding of the pre-activation apart: relative L2 of the outputs, and no element further than
                        # what a pre-activation error of 2^-8 |pre| can do (|gelu'| <= 1.13, |gelu''| <= 0.8: ~2^-7 of the scale)
                        d = (r0[k].float() - r1[k].float())
                        rel = float(d.norm() / r0[k].float().norm())
                        same = rel < a.tol and bool(torch.isfinite(r1[k].float()).all())
                        print("   %s %s: relative L2 %.2e, max |d| %.3g, %.1f %% of the elements differ" % (
                            case, k, rel, float(d.abs().max()), 100.0 * float((d != 0).float().mean())))
                    else:
                        same = torch.equal(r0[k], r1[k])
                    if not same:
                        d = (r0[k].float() - r1[k].float()).abs()
                        print("MISMATCH %s %s: max |d| %.4g at %d of %d elements, finite %s" % (
                            case, k, float(d.max()), int((d > 0).sum()), d.numel(), bool(torch.isfinite(r1[k].float()).all())))
                        bad = (d > 0).nonzero()
                        print("   first bad:", bad[:4].tolist(), " last bad:", bad[-2:].tolist())
                        ok = False
                print("checked %s M=%d rows %d/%d: %s" % (case, Mc, rows0, rows1, "EQUAL" if ok else "DIFFERENT"), flush=True)
                rc |= 0 if ok else 1
                if a.skip_bench:
                    continue
                fl = 2.0 * Mc * kw["N"] * kw["K"]
                for rnd in range(2):
                    for v in [a.base, a.base | a.bit] + [a.base | a.bit | (int(k) << 8) for k in a.abl.split(",") if k] + [int(x) for x in a.variants.split(",") if x]:
                        bufs, _ = run(case, Mc, A, B, kw, v)
                        for _ in range(3):
                            run(case, Mc, A, B, kw, v, bufs)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(a.reps):
                            run(case, Mc, A, B, kw, v, bufs)
                        e1.record()
                        torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1) / a.reps
                        print("  %-16s variant %4d  %8.1f us  %7.1f TFLOP/s" % (case, v, ms * 1e3, fl / ms / 1e9), flush=True)
            for k in [int(x) for x in a.trace.split(",") if x]:
                import ctypes, numpy as np
                A, B, kw = make(case, M)
                bufs, _ = run(case, M, A, B, kw, a.base | a.bit | (k << 8))
                run(case, M, A, B, kw, a.base | a.bit | (k << 8), bufs)
                torch.cuda.synchronize()
                buf = (ctypes.c_uint32 * (256 * 8 * 128))()
                L.load().kbner_debug_read_xtrace(buf)
                t = np.frombuffer(buf, dtype=np.uint32).reshape(256, 8, 128)[:, :, :80].reshape(256, 8, 16, 5).astype(np.int64)
                # segments: P0->P1 (reads + role-0 slice), P1->P2 (MFMA phase), P2->P3 (role-1 slice + tail), P3->P4 (DMA wait), P4->next P0 (barrier)
                seg = np.zeros((256, 8, 16, 5))
                seg[..., 0:4] = (t[..., 1:5] - t[..., 0:4]) & 0xffffffff
                seg[:, :, :15, 4] = (t[:, :, 1:, 0] - t[:, :, :15, 4]) & 0xffffffff
                step = ((t[:, :, 1:, 0] - t[:, :, :15, 0]) & 0xffffffff)
                print("trace id %d: cycles per K step %.0f (p10 %.0f p90 %.0f)" % (k, step.mean(), np.percentile(step, 10), np.percentile(step, 90)))
                for role, ws in (("role0 (waves 0-3)", slice(0, 4)), ("role1 (waves 4-7)", slice(4, 8))):
                    m = seg[:, ws, 1:15, :].mean(axis=(0, 1, 2))
                    print("   %s: reads+slice0 %.0f | mfma phase %.0f | slice1+tail %.0f | dma wait %.0f | barrier %.0f" % (role, *m))
                    for sx in (3, 4, 5, 12):
                        m = seg[:, ws, sx, :].mean(axis=(0, 1))
                        print("        step %2d: %.0f | %.0f | %.0f | %.0f | %.0f" % (sx, *m))
            for k in [int(x) for x in a.strace.split(",") if x]:
                import ctypes, numpy as np
                A, B, kw = make(case, M)
                bufs, _ = run(case, M, A, B, kw, a.base | 32 | (k << 8))
                run(case, M, A, B, kw, a.base | 32 | (k << 8), bufs)
                torch.cuda.synchronize()
                buf = (ctypes.c_uint32 * (256 * 64))()
                L.load().kbner_debug_read_strace(buf)
                t = np.frombuffer(buf, dtype=np.uint32).reshape(256, 64)[:, :20].astype(np.int64)
                d = (t[:, 1:] - t[:, :-1]) & 0xffffffff
                m = d.mean(0)
                print("strace id %d (wave 0 of every workgroup, 6th tile): K steps 0-15 %s" % (k, " ".join("%.0f" % x for x in m[:16])))
                print("     step 15 -> dump start %.0f | dump issue %.0f | lgkm wait %.0f | barrier %.0f ; K-step mean %.0f" % (m[15], m[16], m[17], m[18], m[:15].mean()))
    finally:
        ops.gemm_variant(prev)
    return rc


if __name__ == "__main__":
    sys.exit(main())
