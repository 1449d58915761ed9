#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  G14: flair's BertEmbeddings (the mBERT slot of BASELINE config 5, flair/embeddings.py
:2667-2905) captured by RUNNING THE REFERENCE on a tiny BERT-shaped model (tests/tiny_assets.build_bert_dir): per-token
word-piece tokenisation, [CLS] .. [SEP] framing, zero padding, absolute positions, last-four-layer concatenation of each
token's first piece -- including a control-character token that receives no piece.   python oracle/gen_golden_bert.py"""
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


def main():
    ref_import.load_reference()
    import tiny_assets
    import transformers
    _bm = transformers.BertModel.from_pretrained
    transformers.BertModel.from_pretrained = classmethod(lambda cls, *a, **k: _bm(*a, attn_implementation="eager", **k))
    from flair.custom_data_loader import BatchedData
    from flair.data import Sentence
    from flair.embeddings import BertEmbeddings
    work = tempfile.mkdtemp(prefix="g14_")
    mdir = tiny_assets.build_bert_dir(os.path.join(work, "bert-tiny"), seed=9)
    patch_model_dir(mdir)
    emb = BertEmbeddings(bert_model_or_path=mdir, layers="-1,-2,-3,-4", pooling_operation="first")
    texts = ["alice visited berlin and the museum of art", "zalandoresearchuniversity works \x01 near london", "bob"]
    sents = [Sentence(t) for t in texts]
    batch = BatchedData(sents)
    cap = {}
    fwd = emb.model.forward

    def spy(input_ids, token_type_ids=None, attention_mask=None, **k):
        o = fwd(input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask, **k)
        cap["ids"], cap["mask"] = input_ids.clone(), attention_mask.clone()
        cap["hs"] = [h.detach().clone() for h in o[2]]
        return o

    emb.model.forward = spy
    with torch.no_grad():
        emb.embed(batch)
    out = {"texts": np.asarray(texts), "ids": cap["ids"].numpy(), "mask": cap["mask"].numpy(),
           "features": batch.features[emb.name].numpy(), "n_tokens": np.asarray([len(s) for s in sents]),
           "embedding_length": np.int64(emb.embedding_length)}
    for i, h in enumerate(cap["hs"]):
        out["hs%d" % i] = h.numpy()
    np.savez_compressed(os.path.join(GOLD, "bert_embeddings.npz"), **out)
    shutil.rmtree(work, ignore_errors=True)
    print("  bert_embeddings.npz %8d bytes" % os.path.getsize(os.path.join(GOLD, "bert_embeddings.npz")), out["features"].shape)


if __name__ == "__main__":
    main()
