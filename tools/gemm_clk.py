#!/usr/bin/env python
"""Effective shader clock inside the GEMM kernels (experimental -DG2_TRACE build, tools/gemm_clk.sh): s_memtime (shader cycles)
against s_memrealtime (100 MHz) between workgroup start and end -> cycles per K step and MHz, per main-loop variant."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import labenv; labenv.apply()   # KBNER_LIB / KBNER_GEMM_VARIANT (lab switches live in tools/, not in the product binding)
import numpy as np, torch
from kbner import ops, lib as L
from kbner.lib import GEMM_NT, GEMM_NN, GEMM_TN, EPI_RMW32
LAYOUT = int(os.environ.get('CLK_LAYOUT', '0'))
dev, BF = "cuda", torch.bfloat16
lib = L.load()
lib.kbner_debug_read_clk.argtypes = [ctypes.c_void_p]
import os
VARIANTS = [int(v) for v in os.environ.get("CLK_VARIANTS", "0,1").split(",")]
SHAPES = os.environ.get("CLK_SHAPES", "8192,8192,8192,random;8192,8192,8192,zeros")
for (M, N, K, data) in [(int(x.split(",")[0]), int(x.split(",")[1]), int(x.split(",")[2]), x.split(",")[3]) for x in SHAPES.split(";")]:
    A = (torch.randn(M, K, device=dev) * 0.5).to(BF) if data == "random" else torch.zeros(M, K, device=dev, dtype=BF)
    B = (torch.randn(N, K, device=dev) * 0.5).to(BF) if data == "random" else torch.zeros(N, K, device=dev, dtype=BF)
    C = torch.empty(M, N, device=dev, dtype=BF)
    C32 = torch.zeros(M, N, device=dev) if LAYOUT == 2 else None
    if LAYOUT == 2:
        A = A.t().contiguous()
    if LAYOUT >= 1:
        B = B.t().contiguous()
    def run():
        if LAYOUT == 2:
            ops.gemm(GEMM_TN, A, B, M, N, K, C32=C32, epi=EPI_RMW32)
        else:
            ops.gemm(LAYOUT, A, B, M, N, K, C=C)
    steps = (M // 256) * (N // 256) / 256.0 * (K // 64)
    for rnd in range(2):
        for v in VARIANTS:
            ops.gemm_variant(v)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            buf = np.zeros(256 * 4, np.uint64)
            lib.kbner_debug_read_clk(buf.ctypes.data_as(ctypes.c_void_p))
            t = buf.reshape(256, 4).astype(np.int64)
            cyc = (t[:, 2] - t[:, 0]).astype(np.float64); rt = (t[:, 3] - t[:, 1]).astype(np.float64)
            mhz = cyc / (rt / 100.0)
            if rnd:
                print("layout %d " % LAYOUT + "%s M=%d N=%d K=%d variant %4d: %.3f ms  %.3f us/step  %.0f cycles/step  shader clock %.0f MHz (min %.0f max %.0f)  wg time %.3f ms" % (
                    data, M, N, K, v, ms, ms * 1e3 / steps, cyc.mean() / steps, mhz.mean(), mhz.min(), mhz.max(), rt.mean() / 100e3), flush=True)
ops.gemm_variant(0)
