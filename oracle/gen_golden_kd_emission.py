#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  Golden vectors for the EMISSION-level knowledge-distillation term of a CRF student
(SURVEY.md section 8f-4 residue; model switches distill_emission / distill_prob), produced by RUNNING THE REFERENCE's own method
in this container (imported read-only through oracle/ref_import.py):

  FastSequenceTagger.simple_forward_distillation_loss (flair/models/sequence_tagger_model.py:2110-2372) with
  `distill_emission=True`: the branch `if not self.use_crf or self.distill_emission` (:2311-2365) takes the teachers' emission
  scores from the sentences (`get_teacher_prediction()`: mean over the teachers, flair/data.py:786-807 -- what
  ModelFinetuner.assign_pretrained_teacher_predictions stored, finetune_trainer.py:1417-1492, zero-padded to the batch by
  `resort`, :1937-2047), or -- with distill_posterior also on -- the first teacher's forward-backward scores, and adds
  `_calculate_distillation_loss` (:2384-2398): T^2 * sum_tokens KL(softmax(teacher / T) || softmax(student / T)) / batch size
  (teacher used as given when distill_prob: the trainer stored softmax(logits) then).

Called with stand-in sentences that carry those targets, under autograd: loss, d features, d transitions.
Writes tests/golden/kd_emission.npz.    usage: python oracle/gen_golden_kd_emission.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402
from oracle.gen_golden import GOLD, build_reference_tagger  # noqa: E402


class _Sent:
    """what simple_forward_distillation_loss / _calculate_loss touch on a sentence (flair/data.py:762-806)"""

    def __init__(self, n, tags):
        self.tokens = [None] * n
        self.ner_tags = torch.as_tensor(tags, dtype=torch.int64)
        self._teacher_prediction, self._teacher_posteriors = [], []

    def __len__(self):
        return len(self.tokens)

    def get_teacher_prediction(self, pooling="mean", weight=None):
        assert pooling == "mean"
        return torch.stack(self._teacher_prediction).mean(0)

    def get_teacher_posteriors(self):
        return torch.stack(self._teacher_posteriors, -2)


class _Batch(list):
    pass


def main():
    flair = ref_import.load_reference()
    from flair.models.sequence_tagger_model import START_TAG, STOP_TAG
    dict_path = os.path.join(ref_import.REFERENCE_ROOT, "resources/taggers/EN-English_x.pkl")
    student, td = build_reference_tagger(flair, dict_path)
    teacher, _ = build_reference_tagger(flair, dict_path)
    T = len(td)
    start, stop, unk, x_idx = (td.get_idx_for_item(START_TAG), td.get_idx_for_item(STOP_TAG), td.get_idx_for_item("<unk>"),
                               td.get_idx_for_item("S-X"))
    rng = np.random.default_rng(20220927)
    base = student.transitions.detach().clone().numpy()
    keep = base > -1e11

    def perturbed(scale):
        t = base.copy()
        t[keep] += (rng.standard_normal((T, T)).astype(np.float32) * scale)[keep]
        return t

    trans_s, trans_t = perturbed(0.5), perturbed(0.7)
    valid = [i for i in range(T) if i not in (start, stop, unk, x_idx)]
    # (B, n, tau, interpolation, n_teachers, distill_prob, distill_posterior as well, context tokens)
    SPECS = [
        (3, 9, 1.0, 0.5, 1, False, False, False),
        (4, 17, 3.0, 0.6, 2, False, False, True),
        (3, 11, 1.0, 0.5, 1, True, False, False),
        (4, 13, 2.0, 0.3, 2, True, False, True),
        (3, 10, 2.0, 0.5, 1, False, True, False),
        (2, 1, 1.0, 0.5, 1, False, False, False),
    ]
    cases, ci = {}, 0
    for (B, n, tau, interp, nt, prob, posterior, ctx) in SPECS:
        es = (rng.standard_normal((B, n, T)) * 2.0).astype(np.float32)
        lens = rng.integers(1, n + 1, size=B)
        lens[0] = n
        tags = np.zeros((B, n), np.int64)
        for b in range(B):
            nreal = int(lens[b]) if not ctx else max(1, int(lens[b]) // 2)
            tags[b, :nreal] = rng.choice(valid, size=nreal)
            tags[b, nreal:lens[b]] = x_idx
        lt = torch.from_numpy(lens.astype(np.int64))
        mask = (torch.arange(n)[None, :] < lt[:, None]).float()
        sents = _Batch(_Sent(int(lens[b]), tags[b]) for b in range(B))
        for ti in range(nt):
            et = (es + rng.standard_normal((B, n, T)) * 1.5).astype(np.float32)
            cases["c%d_t%d_logits" % (ci, ti)] = et
            logits = torch.from_numpy(et.copy())
            pred = torch.softmax(logits, -1) if prob else logits           # finetune_trainer.py:1474-1475
            for b, s in enumerate(sents):                                  # :1492 slices to the sentence, resort (:2008-2012) zero-pads
                p = torch.zeros((n, T))
                p[:len(s)] = pred[b, :len(s)]
                s._teacher_prediction.append(p)
            if posterior and ti == 0:
                with torch.no_grad():
                    teacher.transitions.copy_(torch.from_numpy(trans_t))
                    lg = logits.clone()
                    for idx in (stop, start, unk):
                        lg[:, :, idx] -= 1e12
                    fv = teacher._forward_alg(lg, lt, distill_mode=True)
                    bv = teacher._backward_alg(lg, lt)
                    fbs = (fv + bv) * mask.unsqueeze(-1)
                for b, s in enumerate(sents):
                    s._teacher_posteriors.append(fbs[b])
                cases["c%d_t0_fb_score" % ci] = fbs.numpy().copy()
        with torch.no_grad():
            student.transitions.copy_(torch.from_numpy(trans_s))
        student.transitions.grad = None
        student.temperature = tau
        student.distill_emission, student.distill_prob = True, prob
        student.distill_posterior, student.distill_crf, student.crf_attention, student.distill_exact = posterior, False, False, False
        fs = torch.from_numpy(es).requires_grad_(True)

        def forward(data_points, _fs=fs, _mask=mask):
            student.mask = _mask
            return _fs

        student.forward = forward
        loss = student.simple_forward_distillation_loss(sents, interpolation=interp)
        loss.backward()
        with torch.no_grad():
            nll = student._calculate_loss(torch.from_numpy(es), sents, mask)
        for key, v in (("es", es), ("lens", lens.astype(np.int64)), ("tags", tags), ("tau", np.float32(tau)),
                       ("interpolation", np.float32(interp)), ("n_teachers", np.int64(nt)),
                       ("flags", np.asarray([prob, posterior], np.int64)), ("loss", np.float32(loss.item())),
                       ("nll", np.float32(nll.item())), ("des", fs.grad.numpy().copy()),
                       ("dtrans", student.transitions.grad.numpy().copy())):
            cases["c%d_%s" % (ci, key)] = v
        print("case %d: B=%d n=%d tau=%g teachers=%d prob=%s posterior=%s loss=%.6f nll=%.6f" % (ci, B, n, tau, nt, prob, posterior,
                                                                                               loss.item(), nll.item()))
        ci += 1
    cases.update(n_cases=np.int64(ci), trans_s=trans_s, trans_t=trans_t, start=np.int64(start), stop=np.int64(stop),
                 unk=np.int64(unk), x_idx=np.int64(x_idx))
    path = os.path.join(GOLD, "kd_emission.npz")
    np.savez_compressed(path, **cases)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
