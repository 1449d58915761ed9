"""Measurement only (never the product path): the step's nine GEMM shapes through torch.mm (hipBLASLt / rocBLAS on ROCm) next to
libkbner_hip's 256x256x64 kernel, plain epilogues on both sides (no bias / GELU / residual), so DESIGN.md can say how far the
hand-written main loop is from the vendor library's on the same box.   python tools/gemm_vs_library.py [--M 65536]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kb-ner_amd"))
from kbner import ops  # noqa: E402
from kbner.lib import EPI_RMW32, GEMM_NN, GEMM_NT, GEMM_TN  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    M = a.M
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*s):
        return (torch.randn(*s, generator=g) * 0.05).to(torch.bfloat16).to(dev)

    print("%-28s %10s %10s" % ("shape", "kbner TF/s", "torch TF/s"))
    for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        X, W = rnd(M, K), rnd(N, K)
        C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K
        t0 = timed(lambda: ops.gemm(GEMM_NT, X, W, M, N, K, C=C, occupancy=True), a.reps)
        t1 = timed(lambda: torch.mm(X, W.t(), out=C), a.reps)
        print("NT M=%d N=%4d K=%4d      %10.1f %10.1f" % (M, N, K, fl / t0 / 1e12, fl / t1 / 1e12))
    for N, K in ((4096, 1024), (1024, 4096), (1024, 1024), (1024, 3072)):
        X, W = rnd(M, K), rnd(K, N)
        C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K
        t0 = timed(lambda: ops.gemm(GEMM_NN, X, W, M, N, K, C=C, occupancy=True), a.reps)
        t1 = timed(lambda: torch.mm(X, W, out=C), a.reps)
        print("NN M=%d N=%4d K=%4d      %10.1f %10.1f" % (M, N, K, fl / t0 / 1e12, fl / t1 / 1e12))
    for N, K in ((1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)):
        dY, X = rnd(M, N), rnd(M, K)
        C32 = torch.zeros((N, K), dtype=torch.float32, device=dev)
        Cb = torch.empty((N, K), dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K
        t0 = timed(lambda: ops.gemm_grouped(GEMM_TN, [ops.make_problem(dY, X, N, K, M, C32=C32, epi=EPI_RMW32)]), a.reps)
        t1 = timed(lambda: torch.mm(dY.t(), X, out=Cb), a.reps)
        print("TN N=%4d K=%4d inner=%d  %10.1f %10.1f   (kbner: 1 problem alone; the step groups 16)" % (N, K, M, fl / t0 / 1e12, fl / t1 / 1e12))


if __name__ == "__main__":
    main()
