#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  `ref_cpu` of SURVEY.md §8d(i) / BASELINE.md §2: the TRUE reference -- the classes
under /root/reference, imported read-only through oracle/ref_import.py -- timed on this container's host cores on BASELINE
config 1 (xlm-roberta-base dims L12/H768/A12/F3072, V=250 002, random init, dropout off, fp32; micro-batch 2 x accumulate 2;
512-sub-token 'sentence <EOS> context' inputs with 16 real tokens): one optimiser step =
    FastSequenceTagger.forward_loss (sequence_tagger_model.py:1899) -> loss / accum .backward() (finetune_trainer.py:939-957)
    -> clip_grad_norm_(5.0) (:1010) -> transformers-3.0.0-semantics AdamW.step with the trainer's two lr groups (:552-571,:1018)
and FastSequenceTagger.evaluate per batch (:2593).  The reference cannot travel to the GPU box, so this number is recorded in
BASELINE.md; bench.py's `cpu_baseline` leg times the oracle restatement on the GPU box instead.

    python oracle/time_reference_cpu.py [--layers 12] [--steps 2]      (about 2-4 minutes, ~6 GB of RAM)
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_import  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--inter", type=int, default=3072)
    ap.add_argument("--vocab", type=int, default=250002)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--no-port", action="store_true", help="skip the oracle-port leg")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    flair = ref_import.load_reference()
    ref_import.wrap_auto_tokenizer()
    import tiny_assets
    import transformers
    from transformers import XLMRobertaConfig, XLMRobertaModel
    _am = transformers.AutoModel.from_pretrained
    transformers.AutoModel.from_pretrained = staticmethod(lambda *a, **k: _am(*a, attn_implementation="eager", **k))
    from flair.custom_data_loader import BatchedData, ColumnDataLoader
    from flair.data import Dictionary, Sentence
    from flair.embeddings import StackedEmbeddings, TransformerWordEmbeddings
    from flair.models import FastSequenceTagger
    from flair.training_utils import store_embeddings

    work = tempfile.mkdtemp(prefix="refcpu_")
    mdir = os.path.join(work, "xlmr")
    tok = tiny_assets.build_tokenizer_dir(mdir)
    cfg = XLMRobertaConfig(vocab_size=args.vocab, hidden_size=args.hidden, num_hidden_layers=args.layers,
                           num_attention_heads=args.heads, intermediate_size=args.inter, max_position_embeddings=514,
                           type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5, hidden_act="gelu", hidden_dropout_prob=0.0,
                           attention_probs_dropout_prob=0.0, return_dict=False, output_hidden_states=True)
    torch.manual_seed(20220711)
    XLMRobertaModel(cfg, add_pooling_layer=True).save_pretrained(mdir)
    cfg.save_pretrained(mdir)
    # tag dictionary with the EN-English_x layout (T = 29)
    td = Dictionary.load_from_file(os.path.join(ref_import.REFERENCE_ROOT, "resources", "taggers", "EN-English_x.pkl"))
    real_tags = [t for t in td.get_items() if t not in ("<unk>", "<START>", "<STOP>", "S-X")]
    rng = np.random.default_rng(20220711)

    def sentence():
        """16 real tokens, <EOS>, then context words until the tokenizer yields 510 sub-tokens"""
        words = [str(w) for w in rng.choice(tiny_assets.WORDS, size=16)]
        tags = [str(t) for t in rng.choice(real_tags, size=16)]
        words.append("<EOS>")
        tags.append("S-X")
        n_sub = len(tok.tokenize(" ".join(words).replace("<EOS>", "</s>")))
        while True:
            w = str(rng.choice(tiny_assets.WORDS))
            k = len(tok.tokenize(w))
            if n_sub + k > 510:
                break
            words.append(w)
            tags.append("S-X")
            n_sub += k
        s = Sentence(" ".join(words))
        for t, tg in zip(s, tags):
            t.add_tag("ner", tg)
        return s, n_sub

    sents = [sentence() for _ in range(4)]
    print("sub-tokens per sentence:", [n for _, n in sents], "word tokens:", [len(s) for s, _ in sents])
    emb = TransformerWordEmbeddings(model=mdir, layers="-1", pooling_operation="first", fine_tune=True)
    tagger = FastSequenceTagger(hidden_size=256, embeddings=StackedEmbeddings([emb]), tag_dictionary=td, tag_type="ner", use_crf=True,
                                use_rnn=False, use_cnn=False, dropout=0.0, word_dropout=0.0, locked_dropout=0.0, sentence_loss=True,
                                remove_x=True, config=None)
    n_params = sum(p.numel() for p in tagger.parameters())
    loader = ColumnDataLoader([s for s, _ in sents], 2, False, use_bert=False, sort_data=False, sentence_level_batch=True, model=tagger)
    loader.assign_tags("ner", td)
    # optimiser exactly as finetune_trainer.py:552-571 groups it
    lr, lr_rate, accum = 5e-6, 10000, 2
    finetune = {n: p for n, p in tagger.named_parameters() if "embedding" in n or n in ("linear.weight", "linear.bias")}
    other = {n: p for n, p in tagger.named_parameters() if n not in finetune}
    opt = transformers.AdamW([{"params": list(other.values()), "lr": lr * lr_rate}, {"params": list(finetune.values())}], lr=lr)
    tagger.train()

    def step():
        for batch in loader:
            loss = tagger.forward_loss(batch)
            (loss / accum).backward()
            store_embeddings(batch, "none")     # finetune_trainer.py:1046,1059-1061
            batch.features = {}
        torch.nn.utils.clip_grad_norm_(tagger.parameters(), 5.0)
        opt.step()
        tagger.zero_grad()
        return float(loss)

    t0 = time.time()
    l0 = step()     # warm-up: allocator, Adam state
    warm = time.time() - t0
    times = []
    for _ in range(args.steps):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    tagger.eval()
    t0 = time.time()
    with torch.no_grad():
        res, eloss = tagger.evaluate(loader, embeddings_storage_mode="none")
    t_eval = time.time() - t0
    out = {"config": "BASELINE configs[0] shape: L%d/H%d/A%d/F%d V=%d, fp32, micro-batch 2 x accumulate 2, 4 sentences of 510 "
                     "sub-tokens" % (args.layers, args.hidden, args.heads, args.inter, args.vocab),
           "parameters": n_params, "threads": args.threads, "cpu_count": os.cpu_count(), "torch": torch.__version__,
           "transformers": transformers.__version__,
           "train_step_s": [round(t, 2) for t in times], "warmup_step_s": round(warm, 2),
           "train_sentences_per_s": round(4 / (sum(times) / len(times)), 4),
           "evaluate_s_for_4_sentences": round(t_eval, 2), "evaluate_sentences_per_s": round(4 / t_eval, 4), "first_loss": l0}
    print(json.dumps(out), flush=True)
    shutil.rmtree(work, ignore_errors=True)
    if args.no_port:
        return
    # ---- the oracle PORT (oracle/train_step.py, what bench.py's cpu_baseline leg times on the GPU box) on the same workload, same
    # process, same thread count: validates the port's timing against the true reference
    del tagger, opt, emb, loader
    import gc
    gc.collect()
    sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
    from kbner import batch as kb
    from oracle import encoder as oenc
    from oracle import train_step as ots
    T, start, stop, x_idx = 29, 27, 28, 9
    ocfg = oenc.EncoderConfig(vocab_size=args.vocab, hidden_size=args.hidden, num_hidden_layers=args.layers,
                              num_attention_heads=args.heads, intermediate_size=args.inter, max_position_embeddings=514)
    params = oenc.init_params(ocfg, seed=kb.SEED)
    g = torch.Generator().manual_seed(1)
    params["linear.weight"] = torch.empty(T, ocfg.hidden_size).uniform_(-0.03, 0.03, generator=g)
    params["linear.bias"] = torch.zeros(T)
    tr = torch.randn(T, T, generator=g)
    tr[start, :] = -1e12
    tr[:, stop] = -1e12
    params["transitions"] = tr
    trainer = ots.OracleTrainer(params, ocfg, start, stop, x_idx, accum=2, t_total=1000)
    del params
    mbs = []
    for i in range(2):
        b = kb.synthetic_batch(2, 512, vocab=ocfg.vocab_size, T=T, x_idx=x_idx, start=start, stop=stop, n_real=16, seed=kb.SEED + i)
        mbs.append(dict(input_ids=torch.from_numpy(b["input_ids"]), attention_mask=torch.from_numpy(b["attention_mask"]),
                        first_idx=torch.from_numpy(b["first_idx"]), tags=torch.from_numpy(b["tags"].astype(np.int64)),
                        lengths=torch.from_numpy(b["lengths"].astype(np.int64))))

    def pstep():
        for mb in mbs:
            trainer.micro_batch(mb)
        trainer.optimizer_step()

    t0 = time.time()
    pstep()
    pwarm = time.time() - t0
    ptimes = []
    for _ in range(args.steps):
        t0 = time.time()
        pstep()
        ptimes.append(time.time() - t0)
    pout = {"what": "oracle PORT (oracle/train_step.py: fp32 torch restatement, no tokeniser / flair objects) on the same shape, same "
                    "process and threads", "train_step_s": [round(t, 2) for t in ptimes], "warmup_step_s": round(pwarm, 2),
            "train_sentences_per_s": round(4 / (sum(ptimes) / len(ptimes)), 4),
            "port_over_reference": round((sum(times) / len(times)) / (sum(ptimes) / len(ptimes)), 3)}
    print(json.dumps(pout), flush=True)


if __name__ == "__main__":
    main()
