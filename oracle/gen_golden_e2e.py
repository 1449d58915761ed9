#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  Golden vectors that need the reference's WHOLE stack -- reader, loader, embeddings
class, trainer, evaluate -- captured by RUNNING THE REFERENCE (imported read-only from /root/reference through
oracle/ref_import.py) on the tiny KB-NER-shaped corpus of tests/tiny_assets.py.   python oracle/gen_golden_e2e.py

  loader_reader.json   a19/a20: ColumnCorpus (CoNLL reader, `# id` comments, IOB->IOBES so B-X -> S-X) + make_tag_dictionary
                       + ColumnDataLoader at batch 1 and 4: sentences, tag-dictionary order, batch membership, ner_tags rows
                       (flair/datasets.py:852-956, flair/data.py:1083, flair/custom_data_loader.py:84-149,199-378)
  pooling.npz          G7: TransformerWordEmbeddings pooling half on crafted sentences -- a word token the tokenizer drops
                       (0 sub-tokens -> zero vector, embeddings.py:3306-3308), sub-token counts clamped by
                       maximum_subtoken_length (:3182-3195), <EOS> substitution -- ids, mask, last hidden state, features[B,n,H]
  e2e_train.json/.npz  G12: ModelFinetuner.train (finetune_trainer.py:876-1023) for 10 epochs without dropout / shuffling:
                       per-micro-batch losses, train_loss_history, dev_score_history, dev_loss_history, initial + final
                       head / transitions; then FastSequenceTagger.evaluate (sequence_tagger_model.py:2593-2729) on the dev
                       and test loaders with the trained model: every "token gold pred score" line, Result.log_line,
                       main / macro score, detailed_results, eval loss
  encoder_d64.npz      G6 addendum: transformers 5.15 XLMRobertaModel (eager fp32) with head_dim 64 and 3 layers (the HIP
                       attention kernels are d=64 only, so the 'tiny' d=16 case of encoder_tiny.npz cannot run on them)
Fixtures are data only (inputs + the reference's outputs)."""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import ref_import  # noqa: E402


def patch_model_dir(path):
    """what the reference needs from a transformers-5.x config: tuple outputs + all hidden states, eager attention"""
    p = os.path.join(path, "config.json")
    c = json.load(open(p))
    c.update(return_dict=False, output_hidden_states=True)
    json.dump(c, open(p, "w"))


def sent_record(s, tag="ner"):
    return {"tokens": [t.text for t in s], "tags": [t.get_tag(tag).value for t in s]}


def main():
    flair = ref_import.load_reference()
    ref_import.wrap_auto_tokenizer()
    import tiny_assets
    import yaml
    import transformers
    # eager attention for AutoModel.from_pretrained (embeddings.py:2953)
    _orig_am = transformers.AutoModel.from_pretrained
    transformers.AutoModel.from_pretrained = staticmethod(lambda *a, **k: _orig_am(*a, attn_implementation="eager", **k))
    from flair.config_parser import ConfigParser
    from flair.custom_data_loader import ColumnDataLoader
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    import flair.nn

    work = tempfile.mkdtemp(prefix="g12_")
    cfg = tiny_assets.e2e_config(work, word_dropout=0.0, max_epochs=10, shuffle=False, n_train=30, n_dev=9, n_test=7,
                                 save_finetuned_embedding=False)
    patch_model_dir(os.path.join(work, "xlmr-tiny"))
    with open(os.path.join(work, "cfg.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    torch.manual_seed(1)
    cp = ConfigParser(Params.from_file(os.path.join(work, "cfg.yaml")))
    td = cp.tag_dictionary

    # ------------------------------------------------------------------ G12: train
    student = cp.create_student()
    emb = student.embeddings.embeddings[0]
    init = {"linear.weight": student.linear.weight.detach().clone().numpy(), "linear.bias": student.linear.bias.detach().clone().numpy(),
            "transitions": student.transitions.detach().clone().numpy()}
    # ------------------------------------------------------------------ a19 / a20
    rec = {"tag_dictionary": td.get_items(), "corpus": {}, "loaders": {}}
    for part, lst in (("train", cp.corpus.train_list), ("dev", cp.corpus.dev_list), ("test", cp.corpus.test_list)):
        rec["corpus"][part] = [sent_record(s) for s in lst[0]]
    train = list(cp.corpus.train_list[0])
    pos = {id(s): i for i, s in enumerate(train)}
    for bs in (1, 4):
        dl = ColumnDataLoader(train, bs, False, use_bert=False, sort_data=True, sentence_level_batch=True, model=student)
        dl.assign_tags("ner", td)
        rec["loaders"][str(bs)] = {"batches": [[pos[id(s)] for s in b] for b in dl],
                                   "ner_tags": [b.ner_tags.tolist() for b in dl], "num_examples": dl.num_examples}
    dl = ColumnDataLoader(train, 4, False, use_bert=False, sort_data=False, sentence_level_batch=True)
    rec["loaders"]["4_unsorted"] = {"batches": [[pos[id(s)] for s in b] for b in dl]}
    dl = ColumnDataLoader(train, 40, False, use_bert=False, sort_data=True, sentence_level_batch=False)   # token-budget batching
    rec["loaders"]["40_tokens"] = {"batches": [[pos[id(s)] for s in b] for b in dl]}
    with open(os.path.join(GOLD, "loader_reader.json"), "w") as f:
        json.dump(rec, f, indent=1)

    step_losses = []
    _fl = student.forward_loss

    def forward_loss(*a, **k):
        out = _fl(*a, **k)
        step_losses.append(float(out))
        return out

    student.forward_loss = forward_loss
    flair.nn.Model.save = lambda self, f: None                      # the .pt pickles the embeddings OBJECT: not needed here
    ModelFinetuner.final_test = lambda self, *a, **k: 0.0           # (it reloads best-model.pt)
    trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
    out = trainer.train(cp.get_target_path, **cp.config["train"])
    student.forward_loss = _fl
    student.eval()
    final = {"linear.weight": student.linear.weight.detach().numpy(), "linear.bias": student.linear.bias.detach().numpy(),
             "transitions": student.transitions.detach().numpy()}
    enc_sd = {k: v.detach().numpy() for k, v in emb.model.state_dict().items() if "position_ids" not in k and "token_type_ids" not in k}

    # ------------------------------------------------------------------ G12: evaluate with the trained model
    evals = {}
    for part, lst in (("dev", cp.corpus.dev_list), ("test", cp.corpus.test_list)):
        for order in ("evaluate",):
            dl = ColumnDataLoader(list(lst[0]), 4, False, use_bert=False, sort_data=True, sentence_level_batch=True, model=student)
            dl.assign_tags("ner", td)
            path = os.path.join(work, "%s.tsv" % part)
            res, loss = student.evaluate(dl, out_path=path, embeddings_storage_mode="none")
            evals[part] = {"lines": open(path, encoding="utf-8").read().split("\n"), "log_line": res.log_line,
                           "log_header": res.log_header, "main_score": res.main_score, "macro_score": res.macro_score,
                           "detailed_results": res.detailed_results, "eval_loss": float(loss),
                           "batches": [[sent_record(s) for s in b] for b in dl]}
    e2e = {"config_kwargs": dict(word_dropout=0.0, max_epochs=10, shuffle=False, n_train=30, n_dev=9, n_test=7,
                                 save_finetuned_embedding=False),
           "step_losses": step_losses, "train_loss_history": [float(x) for x in out["train_loss_history"]],
           "dev_score_history": [float(x) for x in out["dev_score_history"]],
           "dev_loss_history": [float(x) for x in out["dev_loss_history"]], "evaluate": evals}
    with open(os.path.join(GOLD, "e2e_train.json"), "w") as f:
        json.dump(e2e, f, indent=1, ensure_ascii=False)
    arrs = {"init/" + k: v for k, v in init.items()}
    arrs.update({"final/" + k: v for k, v in final.items()})
    arrs.update({"final_enc/" + k: v for k, v in enc_sd.items()})
    np.savez_compressed(os.path.join(GOLD, "e2e_train.npz"), **arrs)

    # ------------------------------------------------------------------ G7: pooling
    from flair.data import Sentence as RS
    from flair.custom_data_loader import BatchedData
    from flair.embeddings import TransformerWordEmbeddings as RefTWE

    def capture_pooling(emb_obj, texts, max_sub):
        emb_obj.eval()
        old_max = emb_obj.maximum_subtoken_length
        emb_obj.maximum_subtoken_length = max_sub
        sents = [RS(t) for t in texts]
        batch = BatchedData(sents)
        cap = {}
        fwd = emb_obj.model.forward

        def spy(input_ids, attention_mask=None, **k):
            o = fwd(input_ids, attention_mask=attention_mask, **k)
            cap["ids"], cap["mask"] = input_ids.clone(), attention_mask.clone()
            cap["hidden"] = o[2][-1].detach().clone()
            return o

        emb_obj.model.forward = spy
        with torch.no_grad():
            emb_obj.embed(batch)
        emb_obj.model.forward = fwd
        emb_obj.maximum_subtoken_length = old_max
        return {"ids": cap["ids"].numpy(), "mask": cap["mask"].numpy(), "hidden": cap["hidden"].numpy(),
                "features": batch.features[emb_obj.name].detach().numpy(), "texts": np.asarray(texts),
                "n_tokens": np.asarray([len(s) for s in sents]), "maximum_subtoken_length": np.int64(max_sub)}

    # (a) sentencepiece-style tokenizer (the XLM-R case): a soft-hyphen word token is deleted by the normaliser but leaves a lone
    # metaspace piece; a long word is clamped to 3 pieces (:3182-3195); <EOS> -> </s>
    pool = {"sp/" + k: v for k, v in capture_pooling(emb, ["alice visited \u00ad berlin <EOS> the museum of art",
                                                          "zalandoresearchuniversitycambridge is located in berlin",
                                                          "bob <EOS> wikipedia"], 3).items()}
    # (b) BERT-style tokenizer (config 5's mBERT): a control-character word token receives NO sub-token -> zero vector (:3306-3308);
    # <EOS> -> [SEP] (no eos token: :3149-3153)
    wp_dir = os.path.join(work, "bert-tiny")
    tiny_assets.build_model_dir(wp_dir, tokenizer="wordpiece", seed=3)
    patch_model_dir(wp_dir)
    emb_wp = RefTWE(model=wp_dir, layers="-1", pooling_operation="first", fine_tune=False)
    pool.update({"wp/" + k: v for k, v in capture_pooling(emb_wp, ["alice visited \x01 berlin <EOS> the museum of art",
                                                                  "\x01 zalandoresearchuniversity works \x02",
                                                                  "bob <EOS> wikipedia"], 4).items()})
    np.savez_compressed(os.path.join(GOLD, "pooling.npz"), **pool)

    # ------------------------------------------------------------------ G6 addendum: d = 64, 3 layers
    from transformers import XLMRobertaConfig, XLMRobertaModel
    rng = np.random.default_rng(20220711)
    V, H, L, A, F_, S, B = 200, 128, 3, 2, 256, 40, 3
    hcfg = XLMRobertaConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=A, intermediate_size=F_,
                            max_position_embeddings=S + 10, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5,
                            hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hcfg._attn_implementation = "eager"
    torch.manual_seed(11)
    model = XLMRobertaModel(hcfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
            else:
                p.mul_(3.0)
    ids = torch.from_numpy(rng.integers(3, V, size=(B, S))).long()
    ids[:, 0] = 0
    am = torch.ones(B, S, dtype=torch.long)
    for b in range(1, B):
        cut = S - 5 * b
        ids[b, cut - 1] = 2
        ids[b, cut:] = 0
        am[b, cut:] = 0
    ids[0, -1] = 2
    with torch.no_grad():
        o = model(input_ids=ids, attention_mask=am, output_hidden_states=True, return_dict=True)
    cases = {"cfg": np.asarray([V, H, L, A, F_, S + 10], np.int64), "ids": ids.numpy(), "mask": am.numpy(),
             "last": o.last_hidden_state.numpy()}
    for i, h in enumerate(o.hidden_states):
        cases["hs%d" % i] = h.numpy()
    for k, v in model.state_dict().items():
        if "position_ids" not in k and "token_type_ids" not in k:
            cases["w/" + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "encoder_d64.npz"), **cases)

    shutil.rmtree(work, ignore_errors=True)
    for f in ("loader_reader.json", "pooling.npz", "e2e_train.json", "e2e_train.npz", "encoder_d64.npz"):
        print("  %-24s %8d bytes" % (f, os.path.getsize(os.path.join(GOLD, f))))


if __name__ == "__main__":
    main()
