#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  n-best Viterbi golden vectors captured by RUNNING THE REFERENCE's
SequenceTagger._viterbi_decode_nbest (flair/models/sequence_tagger_model.py:1660) on tie-free random inputs (emissions and a
transition matrix WITHOUT the -1e12 sentinels: with them the first recursion step is a 29-way tie at -1e12 whose order torch.topk
does not define).   python oracle/gen_golden_nbest.py  -> tests/golden/viterbi_nbest.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import ref_import  # noqa: E402
from oracle.gen_golden import build_reference_tagger  # noqa: E402


def main():
    flair = ref_import.load_reference()
    tagger, td = build_reference_tagger(flair, os.path.join(ref_import.REFERENCE_ROOT, "resources", "taggers", "EN-English_x.pkl"))
    T = len(td)
    start, stop = td.get_idx_for_item("<START>"), td.get_idx_for_item("<STOP>")
    rng = np.random.default_rng(20220712)
    cases = {"start": np.int64(start), "stop": np.int64(stop)}
    ci = 0
    for B, n, nbest in ((3, 7, 4), (4, 12, 10), (2, 5, 2), (5, 9, 3)):
        feats = rng.standard_normal((B, n, T)).astype(np.float32) * 2.0
        trans = rng.standard_normal((T, T)).astype(np.float32)
        lengths = rng.integers(2, n + 1, size=B)
        lengths[0] = n
        mask = (np.arange(n)[None, :] < lengths[:, None]).astype(np.float32)
        with torch.no_grad():
            tagger.transitions.copy_(torch.from_numpy(trans))
            ps, dec = tagger._viterbi_decode_nbest(torch.from_numpy(feats), torch.from_numpy(mask), nbest)
        cases["c%d_feats" % ci], cases["c%d_trans" % ci], cases["c%d_lengths" % ci] = feats, trans, lengths.astype(np.int64)
        cases["c%d_nbest" % ci] = np.int64(nbest)
        cases["c%d_path_score" % ci], cases["c%d_decode" % ci] = ps.numpy(), dec.numpy().astype(np.int64)
        ci += 1
    cases["n_cases"] = np.int64(ci)
    np.savez_compressed(os.path.join(GOLD, "viterbi_nbest.npz"), **cases)
    print("wrote viterbi_nbest.npz", os.path.getsize(os.path.join(GOLD, "viterbi_nbest.npz")), "bytes")


if __name__ == "__main__":
    main()
