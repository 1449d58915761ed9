#!/usr/bin/env python
"""Attention micro-benchmark through the C ABI (B x A heads x S=512, d=64): TFLOP/s of fwd / bwd."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from kbner import ops
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=32); ap.add_argument("--S", type=int, default=512)
ap.add_argument("--A", type=int, default=16); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
B, S, A = a.B, a.S, a.A
H = A * 64
dev = "cuda"
qkv = torch.randn(B * S, 3 * H, device=dev).to(torch.bfloat16)
dctx = torch.randn(B * S, H, device=dev).to(torch.bfloat16)
ap2 = os.environ.get("ATTN_BENCH_REAL_LEN")   # e.g. 450: mask the tail like a real padded batch
mb = torch.zeros(B, S, device=dev)
if ap2:
    mb[:, int(ap2):] = -10000.0
ctx = torch.zeros(B * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(B, A, S, device=dev)
dws = torch.zeros(B, A, S, device=dev)
dqkv = torch.zeros(B * S, 3 * H, dtype=torch.bfloat16, device=dev)
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps
f = 4.0 * S * S * 64 * A * B
t = timeit(lambda: ops.attn_fwd(qkv, mb, ctx, lse, B, S, H, A))
print("fwd  %8.1f us  %7.1f TFLOP/s (2 matmuls)" % (t * 1e3, f / t / 1e9))
dbias = torch.zeros(3 * H, device=qkv.device)
t = timeit(lambda: ops.attn_bwd(qkv, ctx, dctx, mb, lse, dws, dqkv, B, S, H, A, dbias=dbias))
print("bwd  %8.1f us  %7.1f TFLOP/s (7 matmuls executed; %.1f counting the 5 algorithmic)" % (t * 1e3, 3.5 * f / t / 1e9, 2.5 * f / t / 1e9))
