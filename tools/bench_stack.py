#!/usr/bin/env python
"""BASELINE config 5 (ACE-style stacked embeddings, inference only) at engine level, random weights of the real sizes:
  * encoder forward + Viterbi at the four (B, n') points of SURVEY.md §8d cfg 5 (n' = decoded word tokens per sentence; the encoder
    always reads 512 sub-tokens: sentence + retrieved context)
  * the whole stack at B=32: `--encoders` XLM-R-large-sized frozen encoders + `--lms` character LMs (hidden 2048, ~n'*6 characters)
    -> concat -> BiLSTM(hidden 1000) -> linear -> Viterbi, with the time of each stage
usage: python tools/bench_stack.py [--encoders 3] [--lms 4] [--reps 3]      -> one JSON line per measurement"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kbner import batch as kb  # noqa: E402
from kbner import engine, ops, stack  # noqa: E402


def timed(fn, reps):
    for _ in range(3):   # a forward-only encoder pass is captured into a HIP graph on its third run over a shape (Tagger.INFER_GRAPH)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def measure(encoders=3, lms=4, reps=3, lm_hidden=2048):
    """-> list of result dicts (the four encoder+Viterbi points, then the whole stack)"""
    import types
    a = types.SimpleNamespace(encoders=encoders, lms=lms, reps=reps, lm_hidden=lm_hidden)
    results = []
    dev = "cuda"
    T, start, stop = 29, 27, 28
    cfg = engine.EncoderConfig.large(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    encs = []
    for i in range(a.encoders):
        e = engine.Tagger(cfg, T, start, stop, device=dev, inference=True)
        e.init_random(seed=kb.SEED + i)
        encs.append(e)
    trans = encs[0].arena.param("transitions")
    g = torch.Generator().manual_seed(3)
    rng = np.random.default_rng(5)
    # ---- encoder forward + Viterbi at the cfg-5 points
    for B, n_ in ((32, 16), (32, 64), (256, 32), (32, 512)):
        hb = kb.synthetic_batch(B, 512, vocab=cfg.vocab_size, T=T, x_idx=9, start=start, stop=stop, n_real=n_, seed=kb.SEED)
        b = kb.to_device(hb, dev)
        n_tok = hb["first_idx"].shape[1]
        nn = min(n_, n_tok)
        idx = torch.from_numpy(np.ascontiguousarray(hb["row_idx"].reshape(B, n_tok)[:, :nn]).reshape(-1).astype(np.int32)).to(dev)
        lens = torch.full((B,), nn, dtype=torch.int32, device=dev)

        def enc_vit():
            hid = encs[0].encoder_forward(b["ids"], b["pos_ids"], b["maskbias"], B, 512, need_grad=False)
            em, _ = encs[0].emissions(hid, idx, B, nn)
            return encs[0].viterbi(em, lens)
        dt = timed(enc_vit, a.reps)
        em = torch.randn(B, nn, T, device=dev)
        tv = timed(lambda: ops.crf_viterbi(em, trans, lens, start, stop), 20)
        results.append({"metric": "cfg5 encoder fwd (XLM-R-large, 512 sub-tokens) + linear + Viterbi", "B": B, "n": nn,
                        "n_requested": n_,
                        "note": (None if nn == n_ else "BASELINE.md's point is (B=%d, n'=%d); the synthetic 512-sub-token generator yields at "
                                 "most %d word tokens per sentence (<s>, </s> and multi-piece words take the rest), so n' = %d is timed" %
                                 (B, n_, n_tok, nn)),
                        "sentences_per_s": round(B / dt, 1), "ms": round(dt * 1e3, 3), "viterbi_alone_us": round(tv * 1e6, 1),
                        "viterbi_alone_sentences_per_s": round(B / tv)})
    # ---- the whole stack at B = 32, n' = 20 real tokens (sentences chunked at <EOS>), ~6 characters per token
    B, n_ = 32, 20
    H_lm, H_rnn = a.lm_hidden, 1000
    blocks = [cfg.hidden_size] * a.encoders + [H_lm] * a.lms
    D = sum(blocks)
    k = 1.0 / H_rnn ** 0.5
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * k   # noqa: E731
    rnn = {}
    for sfx in ("", "_reverse"):
        rnn["weight_ih_l0" + sfx], rnn["weight_hh_l0" + sfx] = u(4 * H_rnn, D), u(4 * H_rnn, H_rnn)
        rnn["bias_ih_l0" + sfx], rnn["bias_hh_l0" + sfx] = u(4 * H_rnn), u(4 * H_rnn)
    head = stack.BiLSTMHead(rnn, u(T, 2 * H_rnn), u(T), blocks, H_rnn, dev)
    lms = []
    for i in range(a.lms):
        kk = 1.0 / H_lm ** 0.5
        sd = {"encoder.weight": torch.rand(300, 100, generator=g) * 0.2 - 0.1,
              "rnn.weight_ih_l0": (torch.rand(4 * H_lm, 100, generator=g) * 2 - 1) * kk,
              "rnn.weight_hh_l0": (torch.rand(4 * H_lm, H_lm, generator=g) * 2 - 1) * kk,
              "rnn.bias_ih_l0": torch.zeros(4 * H_lm), "rnn.bias_hh_l0": torch.zeros(4 * H_lm)}
        lms.append(stack.CharLM(sd, H_lm, dev))
    hb = kb.synthetic_batch(B, 512, vocab=cfg.vocab_size, T=T, x_idx=9, start=start, stop=stop, n_real=n_, seed=kb.SEED + 9)
    b = kb.to_device(hb, dev)
    n_tok = hb["first_idx"].shape[1]
    idx = torch.from_numpy(np.ascontiguousarray(hb["row_idx"].reshape(B, n_tok)[:, :n_]).reshape(-1).astype(np.int32)).to(dev)
    lengths = np.full(B, n_, np.int64)
    steps = 2 + n_ * 7
    char_ids = rng.integers(1, 290, size=(steps, B)).astype(np.int32)
    out_rows = np.full((steps, B), -1, np.int32)
    for bb in range(B):
        for t in range(n_):
            out_rows[1 + (t + 1) * 7 - 1, bb] = bb * n_ + t
    lens_d = torch.full((B,), n_, dtype=torch.int32, device=dev)
    times = {}

    def stage(name, fn):
        times[name] = timed(fn, a.reps)

    X = head.new_input(B, n_)

    def run_encoders():
        for i, e in enumerate(encs):
            hid = e.encoder_forward(b["ids"], b["pos_ids"], b["maskbias"], B, 512, need_grad=False)
            ops.gather_rows_into(hid, idx, X, head.cols[i], cfg.hidden_size)

    lm_group = stack.CharLMGroup(lms) if lms else None

    def run_lms():
        if lm_group is not None:
            lm_group.run([char_ids] * len(lms), [out_rows] * len(lms), X, [head.cols[a.encoders + i] for i in range(len(lms))])

    def run_lms_one_by_one():
        for i, lm in enumerate(lms):
            lm.run(char_ids, out_rows, X, head.cols[a.encoders + i])

    def run_head():
        em = head.emissions(X, lengths, B, n_)
        return ops.crf_viterbi(em.contiguous(), trans, lens_d, start, stop)

    def whole():
        run_encoders()
        run_lms()
        return run_head()

    stage("encoders", run_encoders)
    stage("char_lms", run_lms)
    stage("char_lms_one_launch_per_model_and_step", run_lms_one_by_one)
    stage("bilstm_linear_viterbi", run_head)
    stage("whole", whole)
    results.append({"metric": "cfg5 whole stack: %d XLM-R-large encoders + %d char LMs (hidden %d, %d chars) + BiLSTM 1000 + CRF"
                              % (a.encoders, a.lms, H_lm, steps), "B": B, "n": n_, "sentences_per_s": round(B / times["whole"], 1),
                    "ms": {k_: round(v * 1e3, 3) for k_, v in times.items()},
                    "lstm_step_us": {"char_lm": round(times["char_lms"] / max(a.lms, 1) / steps * 1e6, 2),
                                     "bilstm": round(times["bilstm_linear_viterbi"] / n_ * 1e6, 2)}})
    del encs, head, lms, lm_group
    torch.cuda.empty_cache()
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoders", type=int, default=3)
    ap.add_argument("--lms", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lm-hidden", type=int, default=2048)
    a = ap.parse_args()
    for r in measure(a.encoders, a.lms, a.reps, a.lm_hidden):
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
