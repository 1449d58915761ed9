#!/usr/bin/env python
"""Secondary measurements of BASELINE.md's table (not the headline bench line):
  * evaluate path (cfg 1/2 'sentences/s for evaluate'): encoder forward + emissions for every word token + CRF loss + Viterbi
  * CRF only (cfg 5): Viterbi and NLL fwd+bwd sentences/s and tokens/s at (B, n') in {(32,16),(32,64),(256,32),(32,512)}
usage: python tools/bench_extra.py [--model large|base] [--batch 128] [--reps 5]   -> one JSON line per measurement"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch  # noqa: E402

from kbner import batch as kb  # noqa: E402
from kbner import engine, ops  # noqa: E402


def timed(fn, reps):
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
    ap.add_argument("--model", default="large", choices=["large", "base"])
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda"
    T, start, stop, x_idx = 29, 27, 28, 9
    cfg = engine.EncoderConfig.large() if a.model == "large" else engine.EncoderConfig.base()
    tg = engine.Tagger(cfg, T, start, stop, device=dev)
    tg.init_random(seed=kb.SEED)
    hb = kb.synthetic_batch(a.batch, 512, vocab=cfg.vocab_size, T=T, x_idx=x_idx, start=start, stop=stop, seed=kb.SEED)
    b = kb.to_device(hb, dev)

    def evaluate():
        em = tg.forward_features(b)                                      # FastSequenceTagger.forward: all word tokens
        comp = ops.gather_rows_f32(em.view(-1, T), b["crow_idx_all"])     # remove_x compaction (sequence_tagger_model.py:2474-2488)
        comp = comp.view(a.batch, -1, T)
        ops.crf_nll_fwd(comp, tg.arena.param("transitions"), b["ctags"], b["clens"], start, stop)   # evaluate()'s loss
        return tg.viterbi(comp, b["clens"])                              # _obtain_labels

    # compaction index over the ALL-token emission rows
    import numpy as np
    keep = hb["keep"]
    n = keep.shape[1]
    nc = hb["ctags"].shape[1]
    idx = np.full((a.batch, nc), -1, np.int32)
    for r in range(a.batch):
        k = np.nonzero(keep[r])[0]
        idx[r, :len(k)] = r * n + k
    b["crow_idx_all"] = torch.from_numpy(idx.reshape(-1)).to(dev)
    dt = timed(evaluate, a.reps)
    print(json.dumps({"metric": "evaluate sentences/sec XLM-R-%s+CRF seq512 (fwd + loss + Viterbi)" % a.model,
                      "value": round(a.batch / dt, 1), "unit": "sentences/sec", "batch": a.batch, "ms": round(dt * 1e3, 3)}))
    trans = tg.arena.param("transitions")
    g = torch.Generator(device=dev).manual_seed(1)
    for B, n_ in ((32, 16), (32, 64), (256, 32), (32, 512), (4096, 32)):
        em = torch.randn(B, n_, T, device=dev, generator=g)
        lens = torch.full((B,), n_, dtype=torch.int32, device=dev)
        tags = torch.randint(1, 9, (B, n_), device=dev, dtype=torch.int32, generator=g)
        dtr = torch.zeros(T, T, device=dev)
        dl = torch.full((B,), 1.0 / B, device=dev)
        tv = timed(lambda: ops.crf_viterbi(em, trans, lens, start, stop), 20)

        def nll():
            logz, gold, alpha = ops.crf_nll_fwd(em, trans, tags, lens, start, stop)
            ops.crf_nll_bwd(em, trans, tags, lens, alpha, logz, dl, start, stop, dtr)
        tn = timed(nll, 20)
        print(json.dumps({"metric": "CRF only (T=29)", "B": B, "n": n_, "viterbi_us": round(tv * 1e6, 1),
                          "viterbi_sentences_per_s": round(B / tv), "viterbi_tokens_per_s": round(B * n_ / tv),
                          "nll_fwd_bwd_us": round(tn * 1e6, 1), "nll_sentences_per_s": round(B / tn)}))


if __name__ == "__main__":
    main()
