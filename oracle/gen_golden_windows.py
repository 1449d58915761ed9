#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  f-3: the sliding-window path of TransformerWordEmbeddings -- the encode_plus overflow
loop (flair/embeddings.py:3203-3227) and the seam stitching of window states (:3292-3299) -- captured by RUNNING THE REFERENCE on a
tiny model with max_subtokens_sequence_length = 64 and stride = 32, on one batch of sentences spanning 1, 5, 9 and 18 windows:
the input-id rows and mask the reference feeds its encoder, the encoder's last hidden state and the features [B, n, H] it assigns.

The reference calls `tokenizer.encode_plus(list_of_ids, max_length, stride, return_overflowing_tokens, truncation=True)`, a
transformers-3.0.0 API the installed 5.x no longer has; oracle/ref_import.TokenizerAdapter restates it.  3.0.0's `longest_first`
loop returns the overflow in a scrambled order (the window's last stride+1 ids, then the earlier removed ids in REVERSE): the
capture is therefore made twice --
  * `*_intended`: overflow = the tail starting `stride` ids before the cut (what the loop was meant to do, and what every later
    release returns for a single sequence).  This is what the product implements and what tests pin it to;
  * `ids_quirk` / `mask_quirk`: the rows under the literal 3.0.0 loop, kept as the record of the ONE deliberate deviation
    (DESIGN.md section 4).  KB-NER's own data never reaches this path in training (kb/context_process.py:974 budgets every
    sentence + context to one window).
    python oracle/gen_golden_windows.py -> tests/golden/windows.npz"""
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

MAXLEN, STRIDE = 64, 32
N_WORDS = (9, 40, 75, 140)   # 1, 5, 9 and 18 windows of 62 content sub-tokens with the tiny test vocabulary


def capture(emb, texts, mode):
    from flair.custom_data_loader import BatchedData
    from flair.data import Sentence
    ref_import.TokenizerAdapter.OVERFLOW = mode
    sents = [Sentence(t) for t in texts]
    batch = BatchedData(sents)
    cap = {}
    fwd = emb.model.forward

    def spy(input_ids, attention_mask=None, **k):
        o = fwd(input_ids, attention_mask=attention_mask, **k)
        cap["ids"], cap["mask"], cap["hidden"] = input_ids.clone(), attention_mask.clone(), o[2][-1].detach().clone()
        return o

    emb.model.forward = spy
    try:
        with torch.no_grad():
            emb.embed(batch)
    finally:
        emb.model.forward = fwd
        ref_import.TokenizerAdapter.OVERFLOW = "3.0.0"
    return cap, batch.features[emb.name].numpy(), [len(s) for s in sents]


def main():
    ref_import.load_reference()
    ref_import.wrap_auto_tokenizer()
    import tiny_assets
    import transformers
    _am = transformers.AutoModel.from_pretrained
    transformers.AutoModel.from_pretrained = staticmethod(lambda *a, **k: _am(*a, attn_implementation="eager", **k))
    from flair.embeddings import TransformerWordEmbeddings
    work = tempfile.mkdtemp(prefix="windows_")
    mdir = tiny_assets.build_model_dir(os.path.join(work, "enc"), seed=0)
    patch_model_dir(mdir)
    emb = TransformerWordEmbeddings(model=mdir, layers="-1", pooling_operation="first")
    emb.eval()
    emb.max_subtokens_sequence_length, emb.stride, emb.allow_long_sentences = MAXLEN, STRIDE, True
    rng = np.random.default_rng(33)
    texts = [" ".join(str(w) for w in rng.choice(tiny_assets.WORDS, size=int(k))) for k in N_WORDS]
    good, feats, lens = capture(emb, texts, "intended")
    quirk, _, _ = capture(emb, texts, "3.0.0")
    rows_per_sentence = []
    for t in texts:   # windows per sentence under the intended semantics: 62 content ids per row, restart 32 before the cut
        n = len(emb.tokenizer.tokenize(t))
        k, lo = 1, 0
        while lo + (MAXLEN - 2) < n:
            lo += MAXLEN - 2 - STRIDE
            k += 1
        rows_per_sentence.append(k)
    assert sum(rows_per_sentence) == good["ids"].shape[0], (rows_per_sentence, good["ids"].shape)
    np.savez_compressed(os.path.join(GOLD, "windows.npz"), texts=np.asarray(texts), max_len=np.int64(MAXLEN), stride=np.int64(STRIDE),
                        ids_intended=good["ids"].numpy(), mask_intended=good["mask"].numpy(), hidden_intended=good["hidden"].numpy(),
                        features_intended=feats, lengths=np.asarray(lens), rows_per_sentence=np.asarray(rows_per_sentence),
                        ids_quirk=quirk["ids"].numpy(), mask_quirk=quirk["mask"].numpy())
    shutil.rmtree(work, ignore_errors=True)
    print("wrote windows.npz", good["ids"].shape, "rows per sentence", rows_per_sentence, "quirk rows", quirk["ids"].shape)


if __name__ == "__main__":
    main()
