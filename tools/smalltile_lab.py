#!/usr/bin/env python
"""Lab (round 6): the N = 1024 GEMMs of the 4-sentence step (M = 2048) on the round-1 128 x 128-tile kernel WITHOUT split-K against
what the engine runs today (128 x 256 tiles on gemm128i; K >= 3072: split-K 4 / 3 + kbner_splitk_finish).  us per call, back to back
and with a 64-MB flush between calls (cold weights, as inside the step).
    python tools/smalltile_lab.py [--reps 50]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from kbner import ops
from kbner.lib import EPI_ADD, EPI_BIAS, GEMM_NN, GEMM_NT
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
dev, BF = "cuda", torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
M, H, F = 2048, 1024, 4096
flush = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, cold):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(a.reps):
        if cold: flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / a.reps * 1e3


for name, layout, K, splits in (("o-proj NT K=1024", GEMM_NT, H, 0), ("dctx NN K=1024", GEMM_NN, H, 0), ("ffn-down NT K=4096", GEMM_NT, F, 4),
                                ("dx1 NN K=4096", GEMM_NN, F, 4), ("dx NN K=3072", GEMM_NN, 3 * H, 3)):
    A = r(M, K)
    B = r(H, K) if layout == GEMM_NT else r(K, H)
    bias, add = torch.randn(H, device=dev), r(M, H)
    C0, C1 = torch.zeros(M, H, dtype=BF, device=dev), torch.zeros(M, H, dtype=BF, device=dev)
    ws = torch.empty((4, M, H), device=dev)
    kw = dict(bias=bias, addend=add, epi=EPI_BIAS | EPI_ADD) if layout == GEMM_NT else dict(addend=add, epi=EPI_ADD)

    def cur():
        if splits:
            ops.gemm_splitk(layout, A, B, M, H, K, splits, ws, C0, bias=kw.get("bias"), addend=add)
        else:
            ops.gemm(layout, A, B, M, H, K, C=C0, occupancy=True, **kw)

    def small():
        ops.FORCE_128 = True
        try:
            ops.gemm(layout, A, B, M, H, K, C=C1, **kw)
        finally:
            ops.FORCE_128 = False
    cur(); small(); torch.cuda.synchronize()
    d = float((C0.float() - C1.float()).abs().max())
    print("%-20s today %6.1f / %6.1f us   128x128 no split %6.1f / %6.1f us (warm / cold)   max |diff| %.3g" % (
        name, timeit(cur, False), timeit(cur, True), timeit(small, False), timeit(small, True), d), flush=True)
