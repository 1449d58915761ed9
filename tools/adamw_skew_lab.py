#!/usr/bin/env python
"""Lab (round 6): does the RELATIVE placement of AdamW's four fp32 streams (p, g, m, v: the kernel reads index i of all four at once)
matter?  The dense update runs at 1.48 ms on some boxes and 1.74 on others; if same-index elements of the four arrays fall into the
same HBM channel the streams collide.  One buffer, the four arrays carved out of it at offsets n + k * skew floats.
    python tools/adamw_skew_lab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from kbner import ops
dev = "cuda"
n = 24 * 12 * 1024 * 1024          # the GEMM weights of XLM-R-large: 302 M parameters
norm = torch.ones(1, device=dev)
shadow = torch.empty(n, dtype=torch.bfloat16, device=dev)


def run(skew_floats, separate=False, spacer=0, chunks=1):
    if separate and chunks > 1:      # every array as `chunks` allocations stitched by torch.cat? no: as views of ONE allocation per array pair
        bufs = [torch.zeros(n, device=dev) for _ in range(4)]
        p, g, m, v = bufs
    elif separate:
        arr = []
        keep = []
        for _ in range(4):
            arr.append(torch.zeros(n, device=dev))
            if spacer:
                keep.append(torch.zeros(spacer, device=dev))
        p, g, m, v = arr
    else:
        buf = torch.zeros(4 * n + 4 * skew_floats + 64, device=dev)
        p, g, m, v = (buf[k * (n + skew_floats):k * (n + skew_floats) + n] for k in range(4))
    g.normal_(0, 1e-3)
    f = lambda: ops.adamw(p, g, m, v, shadow, n, 1e-5, 0.0, 0.9, 0.999, 1e-6, norm, 5.0, 1.0, False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


print("separate allocations (what the arena does): median %.0f us, min %.0f" % run(0, True), flush=True)
for sp in (1 << 18, 1 << 22, (1 << 24) + 4096):
    print("separate + a %5.1f-MiB spacer allocation after each: median %.0f us, min %.0f" % ((sp * 4 / 2 ** 20,) + run(0, True, spacer=sp)), flush=True)
torch.cuda.empty_cache()
print("separate again (fresh blocks): median %.0f us, min %.0f" % run(0, True), flush=True)
for skew in (0, 1024, 1024 * 1024 + 4096):
    print("one buffer, skew %8d floats (%7.1f KiB): median %.0f us, min %.0f" % ((skew, skew * 4 / 1024) + run(skew)), flush=True)
