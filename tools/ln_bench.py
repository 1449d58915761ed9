#!/usr/bin/env python
"""LayerNorm fwd/bwd micro-benchmark (HBM-bound kernels): us and effective TB/s at the engine's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from kbner import ops
dev, BF, F32 = "cuda", torch.bfloat16, torch.float32
H = 1024
for M in (16384, 65536):
    h = torch.randn(M, H, device=dev).to(BF); dy = torch.randn(M, H, device=dev).to(BF)
    g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev)
    y = torch.empty_like(h); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    dh = torch.empty_like(h); dhm = torch.empty_like(h)
    dg, db, dbias = (torch.zeros(H, device=dev) for _ in range(3))
    def t(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    tf = t(lambda: ops.ln_fwd(h, g, b, 1e-5, y, mean, rstd))
    tb = t(lambda: ops.ln_bwd(dy, h, mean, rstd, g, dh, dg, db, dbias))
    td = t(lambda: ops.ln_bwd(dy, h, mean, rstd, g, dh, dg, db, dbias, dhm=dhm, drop=(123, ops.drop_thresh(0.1))))
    by = M * H * 2
    print("M=%6d  ln_fwd %6.1f us (%.2f TB/s)   ln_bwd %6.1f us (%.2f TB/s)   ln_bwd+dropout %6.1f us (%.2f TB/s)"
          % (M, tf * 1e6, 2 * by / tf / 1e12, tb * 1e6, 3 * by / tb / 1e12, td * 1e6, 4 * by / td / 1e12))
