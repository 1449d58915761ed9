#!/usr/bin/env python
"""Throughput of the DROP-IN path (YAML-equivalent objects -> ModelFinetuner.train) on a synthetic KB-NER-style corpus with
~500-sub-token sentences and an XLM-R-large-sized random encoder: shows what the flair mirror's host side (tokenizer cache,
batch assembly, logging, evaluation) costs on top of the kernels that bench.py times.
usage: python tools/train_throughput.py [--sentences 512] [--batch 32] [--accum 4] [--model large|base]"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def setup(sentences=512, model="large"):
    """synthetic KB-NER-style corpus (12 labelled tokens + <EOS> + ~100 context tokens, ~500 sub-tokens) + a random XLM-R-sized
    encoder directory + the drop-in objects a YAML would build -> (tagger, trainer, ColumnCorpus, tag dictionary, work dir)"""
    import tiny_assets
    import torch  # noqa: F401
    from flair.datasets import ColumnCorpus
    from flair.embeddings import TransformerWordEmbeddings
    from flair.models import FastSequenceTagger
    from flair.trainers import ModelFinetuner
    d = tempfile.mkdtemp(prefix="kbner_tp_")
    dims = dict(hidden=1024, layers=24, heads=16, inter=4096) if model == "large" else dict(hidden=768, layers=12, heads=12, inter=3072)
    tiny_assets.build_model_dir(os.path.join(d, "enc"), **dims)
    rng = np.random.default_rng(0)
    folder = os.path.join(d, "data")
    os.makedirs(folder)

    def sentence(i):
        words = [str(w) for w in rng.choice(tiny_assets.WORDS, size=12)]
        lines = ["# id s%d" % i] + ["%s _ _ %s" % (w, "B-LOC" if k == 3 else "O") for k, w in enumerate(words)]
        lines.append("<EOS> B-X B-X B-X")
        lines += ["%s B-X B-X B-X" % w for w in rng.choice(tiny_assets.WORDS, size=int(rng.integers(96, 104)))]
        return "\n".join(lines) + "\n\n"

    for name, k in (("train.txt", sentences), ("dev.txt", 32), ("test.txt", 32)):
        with open(os.path.join(folder, name), "w") as f:
            for i in range(k):
                f.write(sentence(i))
    cc = ColumnCorpus(folder, {0: "text", 1: "pos", 2: "upos", 3: "ner"}, tag_to_bioes="ner", comment_symbol="# id")
    td = cc.make_tag_dictionary("ner")
    from flair.list_data import ListCorpus
    corpus = ListCorpus(train=[cc.train], dev=[cc.dev], test=[cc.test], targets=["ColumnCorpus-SYN"])
    emb = TransformerWordEmbeddings(model=os.path.join(d, "enc"), layers="-1", pooling_operation="first", fine_tune=True)
    toks = [emb.tokenize_sentence(s) for s in list(cc.train)[:64]]
    assert max(len(t[0]) for t in toks) == 1, "synthetic sentences must fit one window for this measurement"
    from flair.embeddings import StackedEmbeddings
    tagger = FastSequenceTagger(hidden_size=256, embeddings=StackedEmbeddings([emb]), tag_dictionary=td, tag_type="ner", use_crf=True, use_rnn=False,
                                remove_x=True, sentence_loss=True, word_dropout=0.1, dropout=0.0, locked_dropout=0.0)
    trainer = ModelFinetuner(tagger, None, corpus, config={}, distill_mode=False, sentence_level_batch=True)
    sub = [len(t[0][0]) for t in toks]
    return tagger, trainer, cc, td, d, sub


def evaluate_rate(sentences=512, batch=32, model="large"):
    """FastSequenceTagger.evaluate (tokenise -> batch -> encoder forward -> emissions of every word token -> CRF loss -> Viterbi ->
    labels -> span metric) in sentences/s on the synthetic corpus, second pass (the first warms the tokenizer cache and buffers)"""
    import shutil
    import torch
    from flair.custom_data_loader import ColumnDataLoader
    tagger, _, cc, td, d, sub = setup(sentences, model)
    dl = ColumnDataLoader(list(cc.train), batch, sentence_level_batch=True)
    dl.assign_tags("ner", td)
    tagger.eval()
    # the host half of evaluate is small-tensor / numpy work: a wide intra-op pool (the CPU leg of bench.py leaves 32 threads set,
    # a GPU box may report 256 logical CPUs under a 16-CPU quota) only adds wake-up latency to it
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(4, prev_threads)))
    tagger.evaluate(dl)
    torch.cuda.synchronize()
    # host / device split per batch: the device half of batch k (enqueue) runs while the host half of batch k - 1 (finish: wait,
    # labels, spans, metric) is computed; the wall time per batch is ~max(device, host)
    enq_host, fin_host, gpu_ms, evs = [], [], [], []
    orig_enq, orig_fin = tagger._eval_enqueue, tagger._eval_finish

    def enq(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter()
        e0.record()
        r = orig_enq(*a, **k)
        e1.record()
        enq_host.append(time.perf_counter() - t)
        evs.append((e0, e1))
        return r

    def fin(*a, **k):
        t = time.perf_counter()
        r = orig_fin(*a, **k)
        fin_host.append(time.perf_counter() - t)
        return r

    tagger._eval_enqueue, tagger._eval_finish = enq, fin
    t0 = time.perf_counter()
    tagger.evaluate(dl)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tagger._eval_enqueue, tagger._eval_finish = orig_enq, orig_fin
    gpu_ms = [a.elapsed_time(b) for a, b in evs]
    torch.set_num_threads(prev_threads)
    shutil.rmtree(d, ignore_errors=True)
    nb = max(1, len(evs))
    return {"value": round(sentences / dt, 1), "unit": "sentences/sec", "sentences": sentences, "batch": batch,
            "per_batch_ms": {"wall": round(1e3 * dt / nb, 2), "device": round(float(np.mean(gpu_ms)), 2),
                             "host_enqueue": round(1e3 * float(np.mean(enq_host)), 2),
                             "host_finish_incl_wait": round(1e3 * float(np.mean(fin_host)), 2),
                             "host_other (tokenise, batch)": round(1e3 * (dt - sum(enq_host) - sum(fin_host)) / nb, 2)},
            "host_threads": max(1, min(4, prev_threads)),
            "sub_tokens_per_sentence": round(float(np.mean(sub)), 1),
            "what": "FastSequenceTagger.evaluate end to end (host tokenisation + batching + XLM-R-%s-sized encoder forward + emissions "
                    "+ CRF NLL + Viterbi + labels + span metric), second pass over the corpus" % model}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sentences", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--accum", type=int, default=4)
    ap.add_argument("--model", default="large", choices=["large", "base"])
    a = ap.parse_args()
    import torch
    from flair.embeddings import StackedEmbeddings, TransformerWordEmbeddings
    from flair.models import FastSequenceTagger
    tagger, trainer, cc, td, d, lens = setup(a.sentences, a.model)
    out = {}
    if os.environ.get("KBNER_PROFILE"):
        import cProfile
        import pstats
        from flair.custom_data_loader import ColumnDataLoader
        dl = ColumnDataLoader(list(cc.train), a.batch, sentence_level_batch=True)
        dl.assign_tags("ner", td)
        tagger.train()
        for bi in range(4):
            tagger.forward_backward(dl[bi], loss_scale=0.25)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        t0 = time.perf_counter()
        for bi in range(4, 12):
            tagger.forward_backward(dl[bi], loss_scale=0.25)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pr.disable()
        print("forward_backward: %.1f ms per batch of %d" % (dt / 8 * 1e3, a.batch))
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
        return
    if os.environ.get("KBNER_STAGE"):
        # BASELINE config 4 (multi-stage fine-tuning): what lies between two stages on the host -- the HF directory written by
        # save_finetuned_embedding (finetune_trainer.py:1289-1312), the next stage's YAML naming it as `model:`, best-model.pt
        import json
        base = os.path.join(d, "stage1")
        os.makedirs(base, exist_ok=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.save_finetuned_embedding(base)
        t_save = time.perf_counter() - t0
        hf_dir = os.path.join(base, os.path.basename(os.path.join(d, "enc")))
        size = sum(os.path.getsize(os.path.join(hf_dir, f)) for f in os.listdir(hf_dir))
        t0 = time.perf_counter()
        emb2 = TransformerWordEmbeddings(model=hf_dir, layers="-1", pooling_operation="first", fine_tune=True)
        tagger2 = FastSequenceTagger(hidden_size=256, embeddings=StackedEmbeddings([emb2]), tag_dictionary=td, tag_type="ner", use_crf=True, use_rnn=False,
                                     remove_x=True, sentence_loss=True, word_dropout=0.1, dropout=0.0, locked_dropout=0.0)
        torch.cuda.synchronize()
        t_load = time.perf_counter() - t0
        same = bool(torch.equal(tagger2.engine.arena.param("l0.qkv.weight"), tagger.engine.arena.param("l0.qkv.weight")))
        t0 = time.perf_counter()
        tagger.save(os.path.join(base, "best-model.pt"))
        t_pt = time.perf_counter() - t0
        t0 = time.perf_counter()
        FastSequenceTagger.load(os.path.join(base, "best-model.pt"))
        torch.cuda.synchronize()
        t_ptl = time.perf_counter() - t0
        print(json.dumps({"what": "stage hand-over, XLM-R-%s-sized encoder" % a.model, "hf_dir_bytes": size,
                          "save_finetuned_embedding_s": round(t_save, 2), "next_stage_load_to_gpu_s": round(t_load, 2),
                          "weights_identical_after_reload": same, "best_model_pt_save_s": round(t_pt, 2),
                          "best_model_pt_load_s": round(t_ptl, 2)}))
        return
    if os.environ.get("KBNER_EVAL"):
        from flair.custom_data_loader import ColumnDataLoader
        dl = ColumnDataLoader(list(cc.train), a.batch, sentence_level_batch=True)
        dl.assign_tags("ner", td)
        tagger.eval()
        tagger.evaluate(dl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tagger.evaluate(dl)
        torch.cuda.synchronize()
        print("evaluate (no profiler): %.1f sentences/s" % (a.sentences / (time.perf_counter() - t0)))
        import cProfile
        import pstats
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        res, loss = tagger.evaluate(dl)
        pr.disable()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("evaluate: %.1f sentences/s (%d sentences, batch %d)" % (a.sentences / dt, a.sentences, a.batch))
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
        return
    for epochs in (1, 2):  # epoch 1 warms up (tokenizer cache, buffers); report the second epoch's rate
        t0 = time.perf_counter()
        trainer.train(os.path.join(d, "out%d" % epochs), learning_rate=5e-6, mini_batch_size=a.batch, max_epochs=1, lr_rate=10000,
                      gradient_accumulation_steps=a.accum, embeddings_storage_mode="none", fine_tune_mode=True,
                      save_final_model=False, train_with_dev=False, monitor_test=False)
        torch.cuda.synchronize()
        out[epochs] = time.perf_counter() - t0
    print({"sub_tokens_per_sentence_mean": float(np.mean(lens)), "sentences": a.sentences, "micro_batch": a.batch, "accumulate": a.accum,
           "epoch1_s": round(out[1], 2), "epoch2_s": round(out[2], 2),
           "epoch2_sentences_per_s_incl_dev_eval": round(a.sentences / out[2], 1)})


if __name__ == "__main__":
    main()
