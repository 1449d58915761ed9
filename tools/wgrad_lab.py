#!/usr/bin/env python
"""The grouped weight-gradient launch (16 TN problems of 4 encoder layers, K = tokens) in isolation: time per launch per main-loop
variant and -- under `rocprofv3 --pmc ...` (tools/wgrad_pmc.sh) -- its L2-miss traffic.  WGRAD_TOKENS (default 65536 = 128
sentences), WGRAD_VARIANTS (default "0,1"), WGRAD_REPS."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import labenv; labenv.apply()   # KBNER_LIB / KBNER_GEMM_VARIANT (lab switches live in tools/, not in the product binding)
import torch
from kbner import ops
from kbner.lib import GEMM_TN, EPI_RMW32
dev, BF = "cuda", torch.bfloat16
Mp = int(os.environ.get("WGRAD_TOKENS", "65536"))
H, F = 1024, 4096
reps = int(os.environ.get("WGRAD_REPS", "3"))
variants = [int(v) for v in os.environ.get("WGRAD_VARIANTS", "0,1").split(",")]
g = torch.Generator(device=dev).manual_seed(1)
def rnd(r, c):
    return (torch.randn(r, c, device=dev, generator=g) * 0.5).to(BF)
probs, keep = [], []
for layer in range(4):
    dh, act, dpre, x1, dh1, ctx, dqkv, x = rnd(Mp, H), rnd(Mp, F), rnd(Mp, F), rnd(Mp, H), rnd(Mp, H), rnd(Mp, H), rnd(Mp, 3 * H), rnd(Mp, H)
    for dy, xx, n_, k_ in ((dh, act, H, F), (dpre, x1, F, H), (dh1, ctx, H, H), (dqkv, x, 3 * H, H)):
        c32 = torch.zeros(n_, k_, device=dev)
        keep.append((dy, xx, c32))
        probs.append(ops.make_problem(dy, xx, n_, k_, Mp, C32=c32, epi=EPI_RMW32))
flops = sum(2.0 * p.M * p.N * p.K for p in probs)
alg = sum(2.0 * (p.M + p.N) * p.K for p in probs)    # every operand panel once (bf16)
print("tokens %d: %.2f TFLOP, operands %.2f GB per launch (each dY / X matrix once)" % (Mp, flops / 1e12, alg / 1e9), flush=True)
for rnd_ in range(2):
    for v in variants:
        ops.gemm_variant(v)
        ops.gemm_grouped(GEMM_TN, probs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm_grouped(GEMM_TN, probs)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("variant %d: %.3f ms per launch, %.0f TFLOP/s" % (v, ms, flops / ms / 1e9), flush=True)
ops.gemm_variant(1)
