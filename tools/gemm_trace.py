#!/usr/bin/env python
"""Per-tile timeline of the persistent GEMM (experimental -DG2_TRACE build made by tools/gemm_trace.sh): main-loop vs
epilogue time per workgroup.  The cycle counter ticks at ~2 GHz on this part (calibrated against event timings)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import labenv; labenv.apply()   # KBNER_LIB / KBNER_GEMM_VARIANT (lab switches live in tools/, not in the product binding)
import numpy as np, torch
from kbner import ops, lib as L
from kbner.lib import EPI_BIAS, EPI_GELU, GEMM_NT, GEMM_NN, EPI_DGELU, EPI_COLSUM, EPI_COLSUM_WS
dev, BF = "cuda", torch.bfloat16
M, N, K = 65536, 4096, 1024
x = (torch.randn(M, K, device=dev) * 0.5).to(BF); W = (torch.randn(N, K, device=dev) * 0.5).to(BF)
C = torch.empty(M, N, device=dev, dtype=BF); P = torch.empty(M, N, device=dev, dtype=BF); b = torch.randn(N, device=dev)
import os
VAR = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else os.environ.get("TRACE_VARIANTS", "0,1")).split(",")]
from kbner.lib import EPI_ADD
add = (torch.randn(M, N, device=dev)).to(BF)
for var in VAR:
  ops.gemm_variant(var)
  Wt = W.t().contiguous()
  ws = torch.zeros((2 * (M // 256), N), device=dev)
  for name, kw in (("plain", {}), ("bias", dict(bias=b, epi=EPI_BIAS)), ("bias+add", dict(bias=b, addend=add, epi=EPI_BIAS | EPI_ADD)), ("bias+gelu", dict(bias=b, out2=P, epi=EPI_BIAS | EPI_GELU)),
                   ("NN plain", dict(nn=1)), ("NN add", dict(nn=1, addend=add, epi=EPI_ADD)), ("NN dgelu", dict(nn=1, aux=add, epi=EPI_DGELU)), ("NN dgelu+colsum", dict(nn=1, aux=add, epi=EPI_DGELU | EPI_COLSUM | EPI_COLSUM_WS, colsum=ws))):
    kw = dict(kw)
    nn = kw.pop("nn", 0)
    for _ in range(2):
        if nn:
            ops.gemm(GEMM_NN, x, Wt, M, N, K, C=C, **kw)
        else:
            ops.gemm(GEMM_NT, x, W, M, N, K, C=C, **kw)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 32 * 4, np.uint64)
    lib = L.load()
    lib.kbner_debug_read_trace.argtypes = [ctypes.c_void_p]
    rc = lib.kbner_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(256, 32, 4).astype(np.int64)
    nt = 16
    main = (t[:, :nt, 1] - t[:, :nt, 0]); epi = (t[:, :nt, 2] - t[:, :nt, 1])
    gap = t[:, 1:nt, 0] - t[:, :nt - 1, 2]
    # s_memtime / readcyclecounter ticks at 100 MHz on this part (constant clock): report microseconds
    f = 1.0   # report shader cycles (s_memtime)
    print("variant %4d %-16s main-loop %.0f cyc (p10 %.0f p90 %.0f)   epilogue %.0f cyc (p10 %.0f p90 %.0f)   gap %.0f   tile %.0f cyc   rc=%d" % (
        var, name, main.mean() / f, np.percentile(main, 10) / f, np.percentile(main, 90) / f, epi.mean() / f,
        np.percentile(epi, 10) / f, np.percentile(epi, 90) / f, gap.mean() / f, (t[:, nt - 1, 2] - t[:, 0, 0]).mean() / f / nt, rc))
