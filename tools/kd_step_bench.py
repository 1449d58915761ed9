#!/usr/bin/env python
"""What the teacher-student KD terms cost on top of a plain fine-tuning micro-batch at full size (XLM-R-large, 512 sub-tokens, B
sentences of ~130 word tokens, T = 29): Tagger.forward_loss vs Tagger.kd_loss with (a) posterior + 10-best CRF + path weights,
(b) exact pairwise -- teacher targets produced on the device from random teacher emissions.   python tools/kd_step_bench.py [--batch 32]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch  # noqa: E402

from kbner import batch as kb  # noqa: E402
from kbner import engine, ops  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    T, start, stop, x_idx = 29, 27, 28, 9
    cfg = engine.EncoderConfig.large(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    tg = engine.Tagger(cfg, T, start, stop)
    tg.init_random(seed=1)
    B = a.batch
    hb = kb.synthetic_batch(B, 512, vocab=cfg.vocab_size, T=T, x_idx=x_idx, start=start, stop=stop, n_real=16, seed=kb.SEED)
    b = kb.to_device(hb)
    n = hb["row_idx"].size // B
    g = torch.Generator(device="cuda").manual_seed(3)
    t_logits = torch.randn((B, n, T), device="cuda", generator=g) * 2
    t_logits[:, :, [start, stop]] -= 50.0
    t_trans = torch.randn((T, T), device="cuda", generator=g)
    lens = b["lengths"]
    sup = (stop, start, 0)
    t_teacher = timed(lambda: (ops.crf_fb_score(t_logits, t_trans, lens, start, stop, sup),
                               ops.crf_viterbi_nbest(t_logits, t_trans, lens, start, stop, 10),
                               ops.crf_pair_posterior(t_logits, t_trans, lens, 2.0, start, stop, sup)))
    score = ops.crf_fb_score(t_logits, t_trans, lens, start, stop, sup)
    ps, dec = ops.crf_viterbi_nbest(t_logits, t_trans, lens, start, stop, 10)
    valid = torch.arange(n, device="cuda")[None, :] < lens[:, None]
    dec = (dec * valid[:, :, None]).to(torch.int32).contiguous()
    pair, s_sc, e_sc = ops.crf_pair_posterior(t_logits, t_trans, lens, 2.0, start, stop, sup)
    kd_a = {"scores": [score], "targets": dec, "weights": ps, "att_nums": B}
    kd_b = {"exact": (pair, s_sc, e_sc)}
    plain = timed(lambda: tg.forward_loss(b, backward=True))
    ta = timed(lambda: tg.kd_loss(b, kd_a, 0.5, 2.0, backward=True))
    tb = timed(lambda: tg.kd_loss(b, kd_b, 0.5, 2.0, backward=True))
    la, lb = float(tg.kd_loss(b, kd_a, 0.5, 2.0, backward=False)), float(tg.kd_loss(b, kd_b, 0.5, 2.0, backward=False))
    print(json.dumps({"B": B, "word_tokens_per_sentence": n, "plain_fwd_bwd_ms": round(plain, 3),
                      "kd_posterior_crf10_attention_ms": round(ta, 3), "kd_exact_ms": round(tb, 3),
                      "teacher_targets_all_three_ms_per_batch": round(t_teacher, 3), "loss_a": la, "loss_b": lb,
                      "pair_posterior_MB_per_batch": round(pair.numel() * 4 / 1e6, 1)}))


if __name__ == "__main__":
    main()
