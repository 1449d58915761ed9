#!/usr/bin/env python
"""LayerNorm forward / backward at the bench's size ([65536, 1024] bf16): us per call and algorithmic GB/s (fwd 4 B/elem, bwd 6)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from kbner import ops
M, H = 65536, 1024
dev = "cuda"
h = torch.randn(M, H, device=dev).bfloat16(); dy = torch.randn(M, H, device=dev).bfloat16()
y = torch.empty_like(h); dh = torch.empty_like(h)
g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); dbias = torch.zeros(H, device=dev)
big = [torch.randn(M, H, device=dev).bfloat16() for _ in range(8)]   # rotate inputs so that nothing stays cached
def t(fn, n=40):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
f = t(lambda i: ops.ln_fwd(big[i % 8], g, b, 1e-5, y, mean, rstd))
ops.ln_fwd(h, g, b, 1e-5, y, mean, rstd)
bw = t(lambda i: ops.ln_bwd(big[i % 8], h, mean, rstd, g, dh, dg, db, dbias))
print("ln_fwd %.1f us (%.2f TB/s)   ln_bwd %.1f us incl. its column-sum fold (%.2f TB/s)" % (f, M * H * 4 / f / 1e6, bw, M * H * 6 / bw / 1e6))
