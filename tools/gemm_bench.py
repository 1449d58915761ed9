#!/usr/bin/env python
"""GEMM micro-benchmark through the C ABI: TFLOP/s per layout/shape (random bf16 data).
  python tools/gemm_bench.py [--reps 20] [--shapes M,N,K;M,N,K] [--layouts 0,1,2]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from kbner import ops
from kbner.lib import EPI_RMW32

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--shapes", default="16384,1024,1024;16384,4096,1024;16384,1024,4096;16384,3072,1024;4096,4096,4096;8192,8192,8192")
ap.add_argument("--layouts", default="0,1,2")
a = ap.parse_args()
dev = "cuda"
for sh in a.shapes.split(";"):
    M, N, K = (int(x) for x in sh.split(","))
    for layout in (int(x) for x in a.layouts.split(",")):
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
        if layout == 2:
            A = A.t().contiguous()
        if layout >= 1:
            B = B.t().contiguous()
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        C32 = torch.zeros(M, N, dtype=torch.float32, device=dev) if layout == 2 else None
        def run():
            if layout == 2:
                ops.gemm(layout, A, B, M, N, K, C32=C32, epi=EPI_RMW32)
            else:
                ops.gemm(layout, A, B, M, N, K, C=C)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print("layout %d  M=%6d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s" % (layout, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
