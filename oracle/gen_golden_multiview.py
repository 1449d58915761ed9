"""Golden vectors for the multi-view (cooperative-learning) posterior distillation loss ("KD losses"), produced
by RUNNING THE REFERENCE's own methods in this container under autograd: FastSequenceTagger._forward_alg(distill_mode=True),
._backward_alg and ._calculate_distillation_loss exactly as _calculate_multi_view_loss's `distill_posterior` branch chains them
(flair/models/sequence_tagger_model.py:2080-2093).  Writes tests/golden/multiview_kl.npz.
usage: python oracle/gen_golden_multiview.py"""
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
    rng = np.random.default_rng(20220713)
    trans = tagger.transitions.detach().clone().numpy()
    keep = trans > -1e11
    trans[keep] += (rng.standard_normal((T, T)).astype(np.float32) * 0.5)[keep]
    cases, ci = {}, 0
    for (B, n, tau) in ((1, 1, 1.0), (2, 3, 4.0), (3, 9, 2.0), (4, 33, 4.0)):
        es = (rng.standard_normal((B, n, T)) * 2.0).astype(np.float32)
        et = (es + rng.standard_normal((B, n, T)) * 1.0).astype(np.float32)      # the other view: correlated, not equal
        lens = rng.integers(1, n + 1, size=B)
        lens[0] = n
        lt = torch.from_numpy(lens.astype(np.int64))
        with torch.no_grad():
            tagger.transitions.copy_(torch.from_numpy(trans))
        tagger.transitions.grad = None
        tagger.temperature = tau
        fs = torch.from_numpy(es).requires_grad_(True)
        ft = torch.from_numpy(et)
        mask = (torch.arange(n)[None, :] < lt[:, None]).float()
        # the chain of :2080-2093
        fb = (tagger._forward_alg(fs, lt, distill_mode=True) + tagger._backward_alg(fs, lt)) * mask.unsqueeze(-1)
        tfb = ((tagger._forward_alg(ft, lt, distill_mode=True) + tagger._backward_alg(ft, lt)) * mask.unsqueeze(-1)).detach()
        loss = tagger._calculate_distillation_loss(fb, tfb, mask, T=tau)
        loss.backward()
        for k, v in (("es", es), ("et", et), ("lens", lens.astype(np.int64)), ("tau", np.float32(tau)),
                     ("loss", np.float32(loss.item())), ("des", fs.grad.numpy().copy()),
                     ("dtrans", tagger.transitions.grad.numpy().copy())):
            cases["c%d_%s" % (ci, k)] = v
        ci += 1
    cases.update(n_cases=np.int64(ci), trans=trans, start=np.int64(start), stop=np.int64(stop))
    path = os.path.join(GOLD, "multiview_kl.npz")
    np.savez_compressed(path, **cases)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
