#!/usr/bin/env python
"""Round-4 A/B of the 256x256 GEMM main loops (kbner_gemm_set_variant): bit-identity of the ping-pong kernel against the two-stage
kernel on every engine epilogue, then the engine-shaped per-layer mix at M tokens, each variant timed back to back in one process.
  python tools/gemm_pp_lab.py [--M 65536] [--reps 10] [--variants 0,1,3,9,11]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import labenv; labenv.apply()   # KBNER_LIB / KBNER_GEMM_VARIANT (lab switches live in tools/, not in the product binding)
import torch
from kbner import ops
from kbner.lib import (EPI_ADD, EPI_BIAS, EPI_COLSUM, EPI_COLSUM_WS, EPI_DGELU, EPI_GELU, EPI_GELU_FWD, EPI_RMW32, EPI_STORE32,
                       GEMM_NN, GEMM_NT, GEMM_TN)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=65536)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--variants", default="0,1")
ap.add_argument("--skip-check", action="store_true")
ap.add_argument("--skip-bench", action="store_true")
ap.add_argument("--kstep", default="", help="variants for the long-K per-step timing (may include the timing-only ablation bits 16 / 32)")
a = ap.parse_args()
variants = [int(v) for v in a.variants.split(",")]
dev, BF = "cuda", torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)


def check():
    """every epilogue family on shapes that exercise 1 tile, several tiles per CU (persistent walk, ring wrap across tiles), K = 64
    (one stage per tile) and K = 128 / 192 / 320 (ring phases 2, 0, 2 mod 3 at the tile boundary)"""
    torch.manual_seed(1)
    bad = 0
    for (M, N, K) in ((256, 256, 64), (512, 768, 320), (4096, 5120, 128), (8192, 2560, 192), (16384, 4096, 64), (2048, 1024, 1024),
                      (65536, 1024, 64)):
        x, w, wt = r(M, K), r(N, K), r(K, N)
        xt = x.t().contiguous()
        bias = torch.randn(N, device=dev)
        add, aux = r(M, N), r(M, N)
        cases = [
            ("NT plain", GEMM_NT, x, w, dict()),
            ("NT bias", GEMM_NT, x, w, dict(bias=bias, epi=EPI_BIAS)),
            ("NT bias+add", GEMM_NT, x, w, dict(bias=bias, addend=add, epi=EPI_BIAS | EPI_ADD)),
            ("NT bias+add+drop", GEMM_NT, x, w, dict(bias=bias, addend=add, epi=EPI_BIAS | EPI_ADD, drop=(1234, ops.drop_thresh(0.1)))),
            ("NT bias+gelu", GEMM_NT, x, w, dict(bias=bias, epi=EPI_BIAS | EPI_GELU, out2=True)),
            ("NT bias+gelu_fwd", GEMM_NT, x, w, dict(bias=bias, epi=EPI_BIAS | EPI_GELU_FWD)),
            ("NT store32", GEMM_NT, x, w, dict(epi=EPI_STORE32, c32=True)),
            ("NN plain", GEMM_NN, x, wt, dict()),
            ("NN add", GEMM_NN, x, wt, dict(addend=add, epi=EPI_ADD)),
            ("NN dgelu", GEMM_NN, x, wt, dict(aux=aux, epi=EPI_DGELU)),
            ("NN dgelu+colsum_ws", GEMM_NN, x, wt, dict(aux=aux, epi=EPI_DGELU | EPI_COLSUM | EPI_COLSUM_WS, ws=True)),
            ("TN rmw32", GEMM_TN, xt, wt, dict(epi=EPI_RMW32, c32=True)),
        ]
        for name, layout, A, B, kw in cases:
            outs = []
            for v in variants:
                ops.gemm_variant(v)
                k = dict(kw)
                res = []
                if k.pop("c32", False):
                    C32 = torch.full((M, N), 0.5, device=dev)
                    k["C32"] = C32
                    res.append(C32)
                else:
                    C = torch.zeros(M, N, dtype=BF, device=dev)
                    k["C"] = C
                    res.append(C)
                if k.pop("out2", False):
                    o2 = torch.zeros(M, N, dtype=BF, device=dev)
                    k["out2"] = o2
                    res.append(o2)
                if k.pop("ws", False):
                    ws = torch.zeros((2 * (M // ops.gemm_tile_rows(layout, M, N)), N), device=dev)
                    k["colsum"] = ws
                    res.append(ws)
                ops.gemm(layout, A, B, M, N, K, **k)
                torch.cuda.synchronize()
                outs.append(res)
            for v, res in zip(variants[1:], outs[1:]):
                for t0, t1 in zip(outs[0], res):
                    if not torch.equal(t0, t1):
                        bad += 1
                        d = (t0.float() - t1.float()).abs()
                        print("MISMATCH %-20s M=%d N=%d K=%d variant %d: max abs %.4g, %d elements" % (name, M, N, K, v, float(d.max()), int((d > 0).sum())), flush=True)
        print("checked M=%d N=%d K=%d" % (M, N, K), flush=True)
    ops.gemm_variant(0)
    # vs fp64 on one shape per layout (the reference of tests/selftest.py:check_gemm)
    print("bit-identity vs variant %d: %s" % (variants[0], "OK" if bad == 0 else "%d MISMATCHES" % bad), flush=True)
    return bad


def bench():
    M, H, F = a.M, 1024, 4096
    x, ctx, act, dh, dpre, dqkv = r(M, H), r(M, H), r(M, F), r(M, H), r(M, F), r(M, 3 * H)
    Wqkv, Wo, W1, W2 = r(3 * H, H), r(H, H), r(F, H), r(H, F)
    bq, bo, b1, b2 = (torch.randn(n, device=dev) for n in (3 * H, H, F, H))
    o_qkv, o_h, o_pre, o_act, o_dx = r(M, 3 * H), r(M, H), r(M, F), r(M, F), r(M, H)
    g = {k: torch.zeros(s, device=dev) for k, s in (("qkv", (3 * H, H)), ("o", (H, H)), ("w1", (F, H)), ("w2", (H, F)))}
    ws = torch.zeros((2 * (M // 256), F), device=dev)
    cases = [
        ("NT qkv  +bias       ", lambda: ops.gemm(GEMM_NT, x, Wqkv, M, 3 * H, H, C=o_qkv, bias=bq, epi=EPI_BIAS), 2.0 * M * 3 * H * H),
        ("NT o    +bias+add   ", lambda: ops.gemm(GEMM_NT, ctx, Wo, M, H, H, C=o_h, bias=bo, addend=x, epi=EPI_BIAS | EPI_ADD), 2.0 * M * H * H),
        ("NT ffn1 +bias+gelu  ", lambda: ops.gemm(GEMM_NT, x, W1, M, F, H, C=o_act, out2=o_pre, bias=b1, epi=EPI_BIAS | EPI_GELU), 2.0 * M * F * H),
        ("NT ffn2 +bias+add   ", lambda: ops.gemm(GEMM_NT, act, W2, M, H, F, C=o_h, bias=b2, addend=x, epi=EPI_BIAS | EPI_ADD), 2.0 * M * H * F),
        ("NN dpre +dgelu+colws", lambda: ops.gemm(GEMM_NN, dh, W2, M, F, H, C=o_pre, aux=act, epi=EPI_DGELU | EPI_COLSUM | EPI_COLSUM_WS, colsum=ws), 2.0 * M * F * H),
        ("NN dx1  +add        ", lambda: ops.gemm(GEMM_NN, dpre, W1, M, H, F, C=o_dx, addend=dh, epi=EPI_ADD), 2.0 * M * H * F),
        ("NN dctx plain       ", lambda: ops.gemm(GEMM_NN, dh, Wo, M, H, H, C=o_dx), 2.0 * M * H * H),
        ("NN dx   +add        ", lambda: ops.gemm(GEMM_NN, dqkv, Wqkv, M, H, 3 * H, C=o_dx, addend=dh, epi=EPI_ADD), 2.0 * M * H * 3 * H),
        ("TN grouped 4 wgrads ", lambda: ops.gemm_grouped(GEMM_TN, [ops.make_problem(dh, act, H, F, M, C32=g["w2"], epi=EPI_RMW32),
              ops.make_problem(dpre, x, F, H, M, C32=g["w1"], epi=EPI_RMW32), ops.make_problem(dh, ctx, H, H, M, C32=g["o"], epi=EPI_RMW32),
              ops.make_problem(dqkv, x, 3 * H, H, M, C32=g["qkv"], epi=EPI_RMW32)]), 2.0 * M * 12 * H * H),
    ]
    table = {}
    for rnd in range(2):   # two alternating rounds: the second one is the one printed (clocks settled)
        for v in variants:
            ops.gemm_variant(v)
            tot_ms = tot_fl = 0
            for name, fn, fl in cases:
                for _ in range(2): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps): fn()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.reps
                table[(rnd, v, name)] = fl / ms / 1e9
                tot_ms += ms; tot_fl += fl
            table[(rnd, v, "layer total")] = tot_fl / tot_ms / 1e9
            table[(rnd, v, "layer ms")] = tot_ms
    ops.gemm_variant(0)
    for rnd in range(2):
        print("round %d  M=%d  TFLOP/s per variant %s" % (rnd, M, variants))
        for name in [c[0] for c in cases] + ["layer total", "layer ms"]:
            print("  %-22s" % name + "".join("%9.1f" % table[(rnd, v, name)] if name != "layer ms" else "%9.3f" % table[(rnd, v, name)] for v in variants), flush=True)


def kstep():
    """8192^3 per layout: 1024 tiles = 4 per CU, 128 K steps each -> us per K step (tile boundaries included: 4 epilogues in 512 steps).
    The chip's clock drifts by ~10 % inside a process (DVFS), so the variants are timed in 5 interleaved rounds; min and median."""
    vs = [int(v) for v in a.kstep.split(",")]
    M = N = K = 8192
    A, B = r(M, K), r(N, K)
    C = torch.zeros(M, N, dtype=BF, device=dev)
    C32 = torch.zeros(M, N, device=dev)
    for layout in (GEMM_NT, GEMM_NN, GEMM_TN):
        fn = (lambda: ops.gemm(layout, A, B, M, N, K, C32=C32, epi=EPI_RMW32)) if layout == GEMM_TN else (lambda: ops.gemm(layout, A, B, M, N, K, C=C))
        times = {v: [] for v in vs}
        for rnd in range(6):
            for v in (vs if rnd % 2 == 0 else vs[::-1]):
                ops.gemm_variant(v)
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
                if rnd:
                    times[v].append(e0.elapsed_time(e1) / 2)
        line = []
        for v in vs:
            t = sorted(times[v])
            line.append("v%-4d %.3f/%.3f" % (v, t[0] * 1e3 / 512, t[len(t) // 2] * 1e3 / 512))
        print("kstep layout %d (us per K step, min/median of 5): " % layout + "  ".join(line), flush=True)
    ops.gemm_variant(0)


if __name__ == "__main__":
    bad = 0 if a.skip_check else check()
    if a.kstep:
        kstep()
    if not a.skip_bench:
        bench()
    sys.exit(1 if bad else 0)
