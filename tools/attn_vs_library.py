#!/usr/bin/env python
"""Measurement only (never the product path): the attention kernels against torch's scaled_dot_product_attention (the ROCm flash /
memory-efficient backends) at the bench shape -- 256 sentences x 16 heads x 512 x 64, bf16, no mask, no dropout -- forward and
forward + backward, alternating rounds (see tools/gemm_vs_library_fair.py for why).   python tools/attn_vs_library.py [--B 256]"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
import torch.nn.functional as F
from kbner import ops
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev, BF = "cuda", torch.bfloat16
B, S, H, A = a.B, 512, 1024, 16
M = B * S
torch.manual_seed(0)
qkv = (torch.randn(M, 3 * H, device=dev) * 0.5).to(BF)
mb = torch.zeros(B, S, device=dev)
ctx = torch.empty(M, H, dtype=BF, device=dev)
lse = torch.empty(B * A * S, device=dev)
ctx_lo = torch.empty(M * H, dtype=torch.uint8, device=dev)
dctx = (torch.randn(M, H, device=dev) * 0.1).to(BF)
dqkv = torch.empty(M, 3 * H, dtype=BF, device=dev)
dws = torch.empty(B * A * S, device=dev)
dbias = torch.zeros(3 * H, device=dev)
q4 = qkv.view(B, S, 3, A, 64)
q, k, v = (q4[:, :, i].permute(0, 2, 1, 3).contiguous().requires_grad_(True) for i in range(3))
do = dctx.view(B, S, A, 64).permute(0, 2, 1, 3).contiguous()


def timed(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


def kb_fwd():
    ops.attn_fwd(qkv, mb, ctx, lse, B, S, H, A, ctx_lo=ctx_lo)


def kb_fb():
    ops.attn_fwd(qkv, mb, ctx, lse, B, S, H, A, ctx_lo=ctx_lo)
    ops.attn_bwd(qkv, ctx, dctx, mb, lse, dws, dqkv, B, S, H, A, dbias=dbias, ctx_lo=ctx_lo)


def lib_fwd():
    with torch.no_grad():
        F.scaled_dot_product_attention(q, k, v)


def lib_fb():
    o = F.scaled_dot_product_attention(q, k, v)
    o.backward(do)
    q.grad = k.grad = v.grad = None


fns = {"kbner fwd": kb_fwd, "library fwd": lib_fwd, "kbner fwd+bwd": kb_fb, "library fwd+bwd": lib_fb}
res = {nm: [] for nm in fns}
names = list(fns)
for rnd in range(5):
    for nm in (names if rnd % 2 else names[::-1]):
        t = timed(fns[nm])
        if rnd:
            res[nm].append(t)
fl = 4.0 * B * A * S * S * 64
for nm, ts in res.items():
    t = statistics.median(ts)
    print("%-16s %8.1f us (%.1f-%.1f)   %.0f TFLOP/s (forward flops x %s)" % (nm, t, min(ts), max(ts), fl * (3.5 if "bwd" in nm else 1.0) / t / 1e6,
                                                                         "3.5" if "bwd" in nm else "1"), flush=True)
