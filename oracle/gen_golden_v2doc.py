#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  f-3: TransformerWordEmbeddings(v2_doc) -- add_document_embeddings_v2
(flair/embeddings.py:3657-3878) -- captured by RUNNING THE REFERENCE on a tiny model: a 7-sentence document, windows of 60
sub-tokens (model_max_length 62), so sentences near the ends get asymmetric context; ids, mask, last hidden state and the
features [B, n, H] the reference assigns.    python oracle/gen_golden_v2doc.py -> tests/golden/v2doc.npz"""
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


def main():
    ref_import.load_reference()
    ref_import.wrap_auto_tokenizer()
    import tiny_assets
    import transformers
    _am = transformers.AutoModel.from_pretrained
    transformers.AutoModel.from_pretrained = staticmethod(lambda *a, **k: _am(*a, attn_implementation="eager", **k))
    from flair.custom_data_loader import BatchedData
    from flair.data import Sentence
    from flair.embeddings import TransformerWordEmbeddings
    work = tempfile.mkdtemp(prefix="v2doc_")
    mdir = tiny_assets.build_model_dir(os.path.join(work, "enc"), seed=0)
    patch_model_dir(mdir)
    tj = os.path.join(mdir, "tokenizer_config.json")
    cfg = json.load(open(tj))
    cfg["model_max_length"] = 62
    json.dump(cfg, open(tj, "w"))
    emb = TransformerWordEmbeddings(model=mdir, layers="-1", pooling_operation="first", v2_doc=True)
    emb.eval()
    rng = np.random.default_rng(4)
    texts = [" ".join(str(w) for w in rng.choice(tiny_assets.WORDS, size=int(k))) for k in (5, 9, 4, 7, 11, 3, 6)]
    doc = [Sentence(t) for t in texts]
    for i, s in enumerate(doc):
        s.doc, s.doc_pos = doc, i
    batch = BatchedData(doc)
    cap = {}
    fwd = emb.model.forward

    def spy(input_ids, attention_mask=None, **k):
        o = fwd(input_ids, attention_mask=attention_mask, **k)
        cap["ids"], cap["mask"], cap["hidden"] = input_ids.clone(), attention_mask.clone(), o[2][-1].detach().clone()
        return o

    emb.model.forward = spy
    with torch.no_grad():
        emb.embed(batch)
    np.savez_compressed(os.path.join(GOLD, "v2doc.npz"), texts=np.asarray(texts), ids=cap["ids"].numpy(), mask=cap["mask"].numpy(),
                        hidden=cap["hidden"].numpy(), features=batch.features[emb.name].numpy(),
                        batch_pos=np.asarray([s.batch_pos[emb.name] for s in doc]), model_max_length=np.int64(62))
    shutil.rmtree(work, ignore_errors=True)
    print("wrote v2doc.npz", cap["ids"].shape, [tuple(s.batch_pos[emb.name]) for s in doc])


if __name__ == "__main__":
    main()
