#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  G13: BASELINE config 5's inference stack captured by RUNNING THE REFERENCE (imported
read-only from /root/reference): FastSequenceTagger(use_rnn=True) over StackedEmbeddings of two frozen TransformerWordEmbeddings
(one with `use_internal_doc`) and a forward + a backward FlairEmbeddings character LM, embedding selection mask, sentences
chunked at <EOS> by assign_ext_context_doc -- on tiny random-init models (no pretrained weights exist offline).
    python oracle/gen_golden_stack.py      -> tests/golden/stack.npz + stack.json
Captured: every weight the mirror needs (BiLSTM / linear / transitions / both LMs; the transformer dirs are rebuilt from
tests/tiny_assets.py seeds), each embedding's features [B, n, D_i] (flair/embeddings.py:108-124), the emissions of forward()
(sequence_tagger_model.py:844-1052) for two selection masks, and the Viterbi labels + scores of _obtain_labels (:1157).
Fixtures are data only."""
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
from oracle.gen_golden_e2e import patch_model_dir  # noqa: E402

LM_CHARS = list(" \nabcdefghijklmnopqrstuvwxyz0123456789<>EOS.,")


def main():
    flair = ref_import.load_reference()
    ref_import.wrap_auto_tokenizer()
    import tiny_assets
    import transformers
    _am = transformers.AutoModel.from_pretrained
    transformers.AutoModel.from_pretrained = staticmethod(lambda *a, **k: _am(*a, attn_implementation="eager", **k))
    from flair.custom_data_loader import BatchedData, ColumnDataLoader
    from flair.data import Dictionary
    from flair.datasets import ColumnCorpus
    from flair.embeddings import FlairEmbeddings, StackedEmbeddings, TransformerWordEmbeddings
    from flair.list_data import ListCorpus
    from flair.models import FastSequenceTagger, LanguageModel
    from flair.trainers.distillation_trainer import ModelDistiller

    work = tempfile.mkdtemp(prefix="g13_")
    tiny_assets.build_model_dir(os.path.join(work, "enc_a"), seed=0)
    tiny_assets.build_model_dir(os.path.join(work, "enc_b"), seed=5)
    for d in ("enc_a", "enc_b"):
        patch_model_dir(os.path.join(work, d))
    tiny_assets.write_conll_corpus(os.path.join(work, "data"), n_train=6, n_dev=2, n_test=2, seed=3)
    cc = ColumnCorpus(os.path.join(work, "data"), {0: "text", 1: "pos", 2: "upos", 3: "ner"}, tag_to_bioes="ner", comment_symbol="# id")
    corpus = ListCorpus(train=[cc.train], dev=[cc.dev], test=[cc.test], targets=["TINY"])
    td = corpus.make_tag_dictionary(tag_type="ner")

    # character LMs (random init; hidden 48 is deliberately not a multiple of 32)
    cd = Dictionary()
    for ch in LM_CHARS:
        cd.add_item(ch)
    torch.manual_seed(77)
    lms = {}
    for tag, fwd in (("lm_f", True), ("lm_b", False)):
        lm = LanguageModel(cd, fwd, hidden_size=48, nlayers=1, embedding_size=20, nout=None, dropout=0.0)
        with torch.no_grad():
            for p in lm.parameters():
                p.mul_(3.0)     # livelier gates than the default init
        path = os.path.join(work, tag + ".pt")
        lm.save(path)
        lms[tag] = (lm, path)
    embs = [TransformerWordEmbeddings(model=os.path.join(work, "enc_a"), layers="-1", pooling_operation="first", use_internal_doc=True),
            TransformerWordEmbeddings(model=os.path.join(work, "enc_b"), layers="-1", pooling_operation="first"),
            FlairEmbeddings(lms["lm_f"][1]), FlairEmbeddings(lms["lm_b"][1])]
    names = {"enc_a": embs[0].name, "enc_b": embs[1].name, "lm_f": embs[2].name, "lm_b": embs[3].name}
    torch.manual_seed(5)
    tagger = FastSequenceTagger(hidden_size=40, embeddings=StackedEmbeddings(embs), tag_dictionary=td, tag_type="ner", use_crf=True,
                                use_rnn=True, dropout=0.0, word_dropout=0.05, locked_dropout=0.5, sentence_loss=True, remove_x=True,
                                embedding_selector=True, use_rl=True, config=None)
    tagger.eval()

    # assign_ext_context_doc (distillation_trainer.py:675-686) without constructing a trainer
    class _T:
        pass

    t = _T()
    t.corpus = corpus
    ModelDistiller.assign_ext_context_doc(t, corpus)
    sents = list(corpus.train_list[0])
    loader = ColumnDataLoader(sents, 4, False, use_bert=False, sort_data=False, sentence_level_batch=True, model=tagger)
    loader.assign_tags("ner", td)
    order = sorted(names.values())
    out = {"names": names, "sorted_names": order, "tag_dictionary": td.get_items(), "lm_chars": LM_CHARS,
           "corpus_seed": 3, "batches": []}
    arrs = {}
    sd = tagger.state_dict()
    for k, v in sd.items():
        if k.startswith("rnn.") or k in ("linear.weight", "linear.bias", "transitions"):
            arrs["w/" + k] = v.detach().numpy()
    for tag, (lm, _) in lms.items():
        for k, v in lm.state_dict().items():
            arrs["%s/%s" % (tag, k)] = v.detach().numpy()
    with torch.no_grad():
        for bi, batch in enumerate(loader):
            rec = {"sentences": [[tok.text for tok in s] for s in batch], "doc_sentences": [[tok.text for tok in s.doc_sent] for s in batch],
                   "tags": [[tok.get_tag("ner").value for tok in s] for s in batch]}
            for si, sel in enumerate(([1, 1, 1, 1], [1, 0, 1, 1], [0, 1, 0, 1])):
                tagger.selection = torch.tensor(sel)
                for s in batch:
                    s.clear_embeddings()
                batch.features = {}
                feats = tagger.forward(batch)
                if si == 0:
                    for key, nm in names.items():
                        arrs["b%d/feat/%s" % (bi, key)] = batch.features[nm].detach().numpy()
                arrs["b%d/emissions/%d" % (bi, si)] = feats.detach().numpy()
                labels, _ = tagger._obtain_labels(feats, batch)
                rec.setdefault("labels", []).append([[l.value for l in row] for row in labels])
                rec.setdefault("scores", []).append([[float(l.score) for l in row] for row in labels])
                rec.setdefault("selections", []).append(sel)
            out["batches"].append(rec)
    np.savez_compressed(os.path.join(GOLD, "stack.npz"), **arrs)
    with open(os.path.join(GOLD, "stack.json"), "w") as f:
        json.dump(out, f, indent=1)
    shutil.rmtree(work, ignore_errors=True)
    for f in ("stack.npz", "stack.json"):
        print("  %-16s %8d bytes" % (f, os.path.getsize(os.path.join(GOLD, f))))


if __name__ == "__main__":
    main()
