#!/usr/bin/env python
"""Measurement only (never the product path): gemm256f_kernel against the vendor library (torch.mm -> hipBLASLt) on the step's forward /
dgrad shapes, plain epilogues, ALTERNATING rounds.  tools/gemm_vs_library.py times one implementation after the other, and on this
chip the clock follows the recent power history (the first kernel timed after an idle phase is slow, the one after a lighter kernel
fast): its round-2 table ("the library is 0-23 % faster") measured the ORDER as much as the kernels.  Here the two are timed in
rounds, order swapped every round, median (min - max) of 6 rounds of --reps launches.
    python tools/gemm_vs_library_fair.py [--M 131072] [--reps 5]"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from kbner import ops
from kbner.lib import GEMM_NN, GEMM_NT
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--M", type=int, default=131072)
a = ap.parse_args()
dev, BF = "cuda", torch.bfloat16


def timed(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e-3


torch.manual_seed(0)
M = a.M
for layout, name, shapes in ((GEMM_NT, "NT", ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096))),
                             (GEMM_NN, "NN", ((4096, 1024), (1024, 4096), (1024, 1024), (1024, 3072)))):
    for (N, K) in shapes:
        A = (torch.randn(M, K, device=dev) * 0.05).to(BF)
        B = (torch.randn(N, K, device=dev) * 0.05).to(BF) if layout == GEMM_NT else (torch.randn(K, N, device=dev) * 0.05).to(BF)
        C0, C1 = torch.zeros(M, N, dtype=BF, device=dev), torch.zeros(M, N, dtype=BF, device=dev)
        Bm = B.t() if layout == GEMM_NT else B
        fl = 2.0 * M * N * K
        fns = {"kbner": lambda: ops.gemm(layout, A, B, M, N, K, C=C0), "library": lambda: torch.mm(A, Bm, out=C1)}
        res = {k: [] for k in fns}
        for rnd in range(7):
            for k in (("kbner", "library") if rnd % 2 else ("library", "kbner")):
                t = timed(fns[k])
                if rnd:
                    res[k].append(fl / t / 1e12)
        md = {k: statistics.median(v) for k, v in res.items()}
        print("%s M=%d N=%4d K=%4d   kbner %.1f (%.1f-%.1f)   library %.1f (%.1f-%.1f) TFLOP/s   library / kbner %.3f" % (
            name, M, N, K, md["kbner"], min(res["kbner"]), max(res["kbner"]), md["library"], min(res["library"]), max(res["library"]),
            md["library"] / md["kbner"]), flush=True)
