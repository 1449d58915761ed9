#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  Golden vectors for the OTHER branches of the multi-view loss,
produced by running the reference's FastSequenceTagger._calculate_multi_view_loss ITSELF (flair/models/
sequence_tagger_model.py:1958-2107, imported read-only through oracle/ref_import.py) under autograd, with `forward` replaced by a
stand-in that returns prepared student-view emissions / token representations:
  distill_exact       :2049-2087  -- pairwise posteriors of the context view at temperature T as the teacher
                                     (_calculate_xstruct_distillation_loss, :2400-2425)
  calculate_l2_loss   :1988-1996,2026-2037 -- mean squared distance of the two views' token representations
  l2_loss_only        :2038-2039
Writes tests/golden/multiview_branches.npz.      usage: python oracle/gen_golden_multiview_branches.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402
from oracle.gen_golden import GOLD, build_reference_tagger  # noqa: E402


class _Sent:
    def __init__(self, n, orig=None):
        self.tokens = [None] * n
        if orig is not None:
            self.orig_sent = orig

    def __len__(self):
        return len(self.tokens)


class _Batch(list):
    pass


def main():
    flair = ref_import.load_reference()
    from flair.models.sequence_tagger_model import START_TAG, STOP_TAG
    dict_path = os.path.join(ref_import.REFERENCE_ROOT, "resources/taggers/EN-English_x.pkl")
    tagger, td = build_reference_tagger(flair, dict_path)
    T = len(td)
    start, stop, x_idx = td.get_idx_for_item(START_TAG), td.get_idx_for_item(STOP_TAG), td.get_idx_for_item("S-X")
    rng = np.random.default_rng(20220715)
    trans = tagger.transitions.detach().clone().numpy()
    keep = trans > -1e11
    trans[keep] += (rng.standard_normal((T, T)).astype(np.float32) * 0.5)[keep]
    valid = [i for i in range(T) if i not in (start, stop, x_idx, 0)]
    H = 16
    cases, ci = {}, 0
    #        B  n_ctx tau  exact  posterior l2     l2_only
    SPECS = [(3, 11, 1.0, True, False, False, False), (4, 19, 3.0, True, True, False, False), (2, 5, 2.0, True, False, True, False),
             (3, 12, 4.0, False, True, True, False), (3, 9, 1.0, False, True, True, True), (2, 3, 2.0, True, False, False, False)]
    for (B, n, tau, exact, posterior, l2, l2_only) in SPECS:
        # context view: n tokens per sentence, the first real[b] are the sentence, the rest S-X context
        real = rng.integers(1, max(2, n // 2 + 1), size=B)
        real[0] = max(1, n // 2)
        if ci == 5:
            real[:] = 1          # one-token sentences: no tag pair, only the start / end terms
        lens = np.minimum(n, real + rng.integers(1, n, size=B))
        lens[0] = n
        tags = np.zeros((B, n), np.int64)
        for b in range(B):
            tags[b, :real[b]] = rng.choice(valid, size=real[b])
            tags[b, real[b]:lens[b]] = x_idx
        nr = int(real.max())
        feats_ctx = (rng.standard_normal((B, n, T)) * 2.0).astype(np.float32)
        rep_ctx = rng.standard_normal((B, n, H)).astype(np.float32)
        feats_orig = (feats_ctx[:, :nr] + rng.standard_normal((B, nr, T))).astype(np.float32)
        rep_orig = (rep_ctx[:, :nr] + 0.3 * rng.standard_normal((B, nr, H))).astype(np.float32)
        lt = torch.from_numpy(lens.astype(np.int64))
        mask = (torch.arange(n)[None, :] < lt[:, None]).float()
        origs = [_Sent(int(real[b])) for b in range(B)]
        sents = _Batch(_Sent(int(lens[b]), origs[b]) for b in range(B))
        with torch.no_grad():
            tagger.transitions.copy_(torch.from_numpy(trans))
        tagger.transitions.grad = None
        tagger.temperature = tau
        tagger.distill_exact, tagger.distill_posterior = exact, posterior
        tagger.calculate_l2_loss, tagger.l2_loss_only = l2, l2_only
        fs = torch.from_numpy(feats_orig).requires_grad_(True)
        rs = torch.from_numpy(rep_orig).requires_grad_(True)
        rl = torch.from_numpy(real.astype(np.int64))

        def forward(orig_sentences, _fs=fs, _rs=rs, _rl=rl, _nr=nr):
            tagger.sentence_tensor = _rs
            tagger.mask = (torch.arange(_nr)[None, :] < _rl[:, None]).float()
            return _fs

        tagger.forward = forward
        tagger.sentence_tensor = torch.from_numpy(rep_ctx)
        loss = tagger._calculate_multi_view_loss(torch.from_numpy(feats_ctx), sents, mask, torch.from_numpy(tags))
        loss.backward()
        zero = np.zeros((T, T), np.float32)
        for k, v in (("feats_ctx", feats_ctx), ("rep_ctx", rep_ctx), ("feats_orig", feats_orig), ("rep_orig", rep_orig),
                     ("lens", lens.astype(np.int64)), ("real", real.astype(np.int64)), ("tags", tags), ("tau", np.float32(tau)),
                     ("flags", np.asarray([exact, posterior, l2, l2_only], np.int64)), ("loss", np.float32(loss.item())),
                     ("dfeats", fs.grad.numpy().copy() if fs.grad is not None else np.zeros_like(feats_orig)),
                     ("drep", rs.grad.numpy().copy() if rs.grad is not None else np.zeros_like(rep_orig)),
                     ("dtrans", tagger.transitions.grad.numpy().copy() if tagger.transitions.grad is not None else zero)):
            cases["c%d_%s" % (ci, k)] = v
        print("case %d: B=%d n=%d tau=%g flags=%s loss=%.6f" % (ci, B, n, tau, (exact, posterior, l2, l2_only), loss.item()))
        ci += 1
    cases.update(n_cases=np.int64(ci), trans=trans, start=np.int64(start), stop=np.int64(stop), x_idx=np.int64(x_idx))
    path = os.path.join(GOLD, "multiview_branches.npz")
    np.savez_compressed(path, **cases)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
