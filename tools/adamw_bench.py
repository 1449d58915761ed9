#!/usr/bin/env python
"""Time FusedAdamW.step (grad-norm pass + clip + AdamW + zero_grad + bf16 shadow) on the XLM-R-large arena (560 M parameters, every
word-embedding row live): ms per step and the HBM rate over the bytes the step moves (30 B / GEMM-weight parameter, 32 B / other
parameter, + 4 B / parameter for the norm pass).   python tools/adamw_bench.py [--reps 10]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch  # noqa: E402

from kbner import engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    cfg = engine.EncoderConfig.large()
    tg = engine.Tagger(cfg, 29, 27, 28, device="cuda")
    tg.init_random(seed=1)
    opt = engine.FusedAdamW(tg.arena, lr=5e-6, lr_rate=10000.0, t_total=1000)
    ar = tg.arena
    if ar.emb_flags is not None:
        ar.emb_flags.fill_(3)   # LIVE | TOUCHED: every row's gradient is read (the all-rows-touched case)

    def step():
        ar.g.normal_(0, 1e-3)
        if ar.emb_flags is not None:
            ar.emb_flags.fill_(3)
        ar.wgrad_stale = False      # as after a backward pass (which overwrites the GEMM-weight gradients AdamW no longer zeroes)
        opt.step()

    step()
    torch.cuda.synchronize()
    times = []
    for _ in range(a.reps):
        ar.g.normal_(0, 1e-3)
        if ar.emb_flags is not None:
            ar.emb_flags.fill_(3)
        ar.wgrad_stale = False
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = sorted(times)[len(times) // 2]
    n = ar.n
    V, H = ar.shapes["emb.word"]
    # bytes moved: GEMM weights 30 B (p, g, m, v read; p, m, v + 2-B shadow written; g left for the next backward to overwrite),
    # embedding rows 32 B, the other dense parameters 32 B (+ zeroed g), + 4 B / parameter for the norm pass
    ns = ar.n_shadow if ar.wgrad_overwrite_ok else 0
    nb = ns * 30 + (n - V * H - ns) * 32 + (0 if ar.wgrad_overwrite_ok else ar.n_shadow * 2) + V * H * 32 + n * 4
    print(json.dumps({"params": n, "ms_median": round(ms, 3), "ms_min": round(min(times), 3), "algorithmic_GB": round(nb / 1e9, 2),
                      "TB_per_s": round(nb / ms / 1e9, 3), "frac_of_8TBs": round(nb / ms / 1e9 / 8.0, 3)}))


if __name__ == "__main__":
    main()
