"""Golden vectors for the posterior-decoding row, produced by RUNNING THE REFERENCE in this container:
SequenceTagger._forward_alg(distill_mode=True) (:1329) + _backward_alg (:1396) + the softmax / argmax of the
predict_posterior branch of _obtain_labels (:1182-1192).  Writes tests/golden/posterior.npz.  (Separate from gen_golden.py so
the earlier fixtures stay byte-identical.)   usage: python oracle/gen_golden_posterior.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402
from oracle.gen_golden import GOLD, build_reference_tagger  # noqa: E402


def main():
    flair = ref_import.load_reference()
    from flair.models.sequence_tagger_model import START_TAG, STOP_TAG
    dict_path = os.path.join(ref_import.REFERENCE_ROOT, "resources/taggers/EN-English_x.pkl")
    tagger, td = build_reference_tagger(flair, dict_path)
    T = len(td)
    start, stop = td.get_idx_for_item(START_TAG), td.get_idx_for_item(STOP_TAG)
    rng = np.random.default_rng(20220712)
    trans = tagger.transitions.detach().clone().numpy()
    keep = trans > -1e11
    trans[keep] += (rng.standard_normal((T, T)).astype(np.float32) * 0.5)[keep]
    cases, ci = {}, 0
    for (B, n) in ((1, 1), (2, 3), (3, 9), (4, 33)):
        feats = (rng.standard_normal((B, n, T)) * 2.0).astype(np.float32)
        lens = rng.integers(1, n + 1, size=B)
        lens[0] = n
        with torch.no_grad():
            tagger.transitions.copy_(torch.from_numpy(trans))
            lt = torch.from_numpy(lens.astype(np.int64))
            f = torch.from_numpy(feats)
            fw = tagger._forward_alg(f, lt, distill_mode=True)
            bw = tagger._backward_alg(f, lt)
            mask = (torch.arange(n)[None, :] < lt[:, None]).float()
            score = (fw + bw) * mask.unsqueeze(-1)
            dist = torch.nn.functional.softmax(score, dim=-1)
            idx = torch.max(score, -1)[1]
        for k, v in (("feats", feats), ("lens", lens.astype(np.int64)), ("fw", fw.numpy()), ("bw", bw.numpy()),
                     ("dist", dist.numpy()), ("idx", idx.numpy())):
            cases["c%d_%s" % (ci, k)] = v
        ci += 1
    cases.update(n_cases=np.int64(ci), trans=trans, start=np.int64(start), stop=np.int64(stop))
    np.savez_compressed(os.path.join(GOLD, "posterior.npz"), **cases)
    print("wrote", os.path.join(GOLD, "posterior.npz"), os.path.getsize(os.path.join(GOLD, "posterior.npz")), "bytes")


if __name__ == "__main__":
    main()
