#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  Golden vectors for the teacher-student knowledge-distillation losses (SURVEY.md section
8f-4), produced by RUNNING THE REFERENCE's own methods in this container (imported read-only through oracle/ref_import.py):

  teacher side   the chain ModelFinetuner.assign_pretrained_teacher_targets runs on a teacher's logits
                 (flair/trainers/finetune_trainer.py:1600 n-best decode; :1627-1634 forward-backward scores after the
                 START / STOP / <unk> logits were lowered by 1e12; :1705-1722 pairwise posteriors + start / end scores of
                 distill_exact), calling teacher._viterbi_decode_nbest / ._forward_alg(distill_mode=True) / ._backward_alg;
  student side   FastSequenceTagger.simple_forward_distillation_loss itself (flair/models/sequence_tagger_model.py:2110-2372),
                 called with stand-in sentences that carry those targets, under autograd: loss, d features, d transitions.

Writes tests/golden/kd_loss.npz.    usage: python oracle/gen_golden_kd.py"""
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
        self._teacher_target, self._teacher_weights, self._teacher_posteriors = [], [], []

    def __len__(self):
        return len(self.tokens)

    def get_teacher_target(self):
        return torch.cat(self._teacher_target, -1)

    def get_teacher_weights(self):
        return torch.cat(self._teacher_weights, -1)

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
    rng = np.random.default_rng(20220714)
    base = student.transitions.detach().clone().numpy()
    keep = base > -1e11

    def perturbed(scale):
        t = base.copy()
        t[keep] += (rng.standard_normal((T, T)).astype(np.float32) * scale)[keep]
        return t

    trans_s = perturbed(0.5)
    trans_ts = [perturbed(0.7), perturbed(0.4)]
    # With the -1e12 sentinels every real model carries (row START, column STOP of transitions[to, from]) the reference's n-best
    # decoder, which reads the matrix as [from, to] (:1688-1693), starts every path at -1e12: in fp32 that absorbs the emissions,
    # EVERY candidate ties, the path weights come out uniform and the tags are whatever torch.topk's unspecified tie order gives.
    # The n-best cases therefore use teacher matrices WITHOUT sentinels (tie-free: the only inputs on which that decoder is a
    # function); one sentinel case is kept to record the degenerate behaviour (uniform weights) itself.
    free_ts = [rng.standard_normal((T, T)).astype(np.float32), rng.standard_normal((T, T)).astype(np.float32)]
    valid = [i for i in range(T) if i not in (start, stop, unk, x_idx)]

    # (B, n, tau, interpolation, n_teachers, best_k, posterior, crf, attention, exact, context tokens)
    SPECS = [
        (3, 9, 1.0, 0.5, 1, 0, True, False, False, False, False),
        (4, 17, 4.0, 0.5, 2, 0, True, False, False, False, False),
        (3, 9, 1.0, 0.5, 1, 3, False, True, False, False, False),
        (3, 9, 1.0, 0.5, 1, 3, False, True, True, False, "sentinel"),
        (4, 12, 1.0, 0.7, 2, 4, False, True, True, False, False),
        (3, 14, 2.0, 0.3, 1, 5, True, True, False, False, True),
        (2, 1, 2.0, 0.5, 1, 1, True, True, True, False, False),
        (3, 9, 1.0, 0.5, 1, 0, False, False, False, True, False),
        (4, 13, 3.0, 0.6, 1, 0, False, False, False, True, False),
        (2, 1, 2.0, 0.5, 1, 0, False, False, False, True, False),
        # distill_with_gold (:2289-2304): path weights scaled by gold_const / (errors + gold_const), or exp(-errors / gold_const)
        (4, 11, 1.0, 0.5, 2, 3, False, True, True, False, ("gold", 1.0)),
        (3, 10, 1.0, 0.4, 1, 4, True, True, True, False, ("gold_exp", 2.0)),
    ]
    cases, ci = {}, 0
    for (B, n, tau, interp, nt, k, posterior, crf, att, exact, ctx) in SPECS:
        es = (rng.standard_normal((B, n, T)) * 2.0).astype(np.float32)
        lens = rng.integers(1, n + 1, size=B)
        lens[0] = n
        tags = np.zeros((B, n), np.int64)
        sentinel = ctx == "sentinel" or not crf
        with_gold, exp_score, gold_const = False, False, 1.0
        if isinstance(ctx, tuple):
            with_gold, exp_score, gold_const = True, ctx[0] == "gold_exp", ctx[1]
        ctx = ctx is True
        for b in range(B):
            nreal = int(lens[b]) if not ctx else max(1, int(lens[b]) // 2)
            tags[b, :nreal] = rng.choice(valid, size=nreal)
            tags[b, nreal:lens[b]] = x_idx
        lt = torch.from_numpy(lens.astype(np.int64))
        mask = (torch.arange(n)[None, :] < lt[:, None]).float()
        sents = _Batch(_Sent(int(lens[b]), tags[b]) for b in range(B))
        tf = {}
        t_logits = []
        for ti in range(nt):
            et = (es + rng.standard_normal((B, n, T)) * 1.5).astype(np.float32)
            if crf and not sentinel:
                et[:, :, [start, stop]] -= 50.0      # a trained teacher never proposes START / STOP as a token's tag
            t_logits.append(et)
            with torch.no_grad():
                tt = trans_ts[ti] if sentinel else free_ts[ti]
                cases["c%d_t%d_trans" % (ci, ti)] = tt
                teacher.transitions.copy_(torch.from_numpy(tt))
                logits = torch.from_numpy(et.copy())
                m3 = mask.unsqueeze(-1).long()
                if crf:
                    path_score, decode_idx = teacher._viterbi_decode_nbest(logits, m3, k)
                    for b, s in enumerate(sents):
                        if att:
                            s._teacher_weights.append(path_score[b])
                        s._teacher_target.append(decode_idx[b] * m3[b])
                    cases["c%d_t%d_path_score" % (ci, ti)] = path_score.numpy().copy()
                    cases["c%d_t%d_decode" % (ci, ti)] = (decode_idx * m3).numpy().copy()
                if posterior or exact:
                    for idx in (stop, start, unk):
                        logits[:, :, idx] -= 1e12
                    fv = teacher._forward_alg(logits, lt, distill_mode=True)
                    bv = teacher._backward_alg(logits, lt)
                if posterior:
                    fbs = (fv + bv) * m3.float()
                    for b, s in enumerate(sents):
                        s._teacher_posteriors.append(fbs[b])
                    cases["c%d_t%d_fb_score" % (ci, ti)] = fbs.numpy().copy()
                if exact:
                    br = torch.arange(B)
                    bm = student.sequence_mask(lt - 1, n - 1).long() if n > 1 else torch.zeros((B, 0), dtype=torch.long)
                    ss = logits[:, :, :, None] + teacher.transitions[None, None, :, :]
                    pair = (fv[:, :-1, None, :] + bv[:, 1:, :, None] + ss[:, 1:]) * bm.unsqueeze(-1).unsqueeze(-1).float() / tau
                    s_sc = (ss[:, 0, :, start] + bv[:, 0, :]) / tau
                    e_sc = (teacher.transitions[None, stop, :] + fv[br, lt - 1, :]) / tau
                    pair = pair.view(B, pair.shape[1], T * T).softmax(-1)
                    tf = {"posteriors": pair.unsqueeze(2), "start_scores": s_sc.unsqueeze(1), "end_scores": e_sc.unsqueeze(1)}
                    cases["c%d_t%d_pair" % (ci, ti)] = pair.numpy().copy()
                    cases["c%d_t%d_start_score" % (ci, ti)] = s_sc.numpy().copy()
                    cases["c%d_t%d_end_score" % (ci, ti)] = e_sc.numpy().copy()
        if tf:
            sents.teacher_features = tf
        with torch.no_grad():
            student.transitions.copy_(torch.from_numpy(trans_s))
        student.transitions.grad = None
        student.temperature = tau
        student.distill_posterior, student.distill_crf, student.crf_attention, student.distill_exact = posterior, crf, att, exact
        student.distill_with_gold, student.exp_score, student.gold_const = with_gold, exp_score, gold_const
        fs = torch.from_numpy(es).requires_grad_(True)

        def forward(data_points, _fs=fs, _mask=mask):
            student.mask = _mask
            return _fs

        student.forward = forward
        if exact:
            # the method builds its length mask with a hard .cuda() on a LongTensor of (lengths - 1) (:2170): shimmed to identity
            pass
        loss = student.simple_forward_distillation_loss(sents, interpolation=interp)
        loss.backward()
        # the plain NLL part on its own (what (1 - interpolation) multiplies)
        with torch.no_grad():
            nll = student._calculate_loss(torch.from_numpy(es), sents, mask)
        for key, v in (("es", es), ("lens", lens.astype(np.int64)), ("tags", tags), ("tau", np.float32(tau)),
                       ("interpolation", np.float32(interp)), ("n_teachers", np.int64(nt)), ("best_k", np.int64(k)),
                       ("flags", np.asarray([posterior, crf, att, exact], np.int64)), ("sentinel", np.int64(sentinel)),
                       ("with_gold", np.asarray([with_gold, exp_score], np.int64)), ("gold_const", np.float32(gold_const)), ("loss", np.float32(loss.item())),
                       ("nll", np.float32(nll.item())), ("des", fs.grad.numpy().copy()),
                       ("dtrans", student.transitions.grad.numpy().copy())):
            cases["c%d_%s" % (ci, key)] = v
        for ti in range(nt):
            cases["c%d_t%d_logits" % (ci, ti)] = t_logits[ti]
        print("case %d: B=%d n=%d tau=%g flags=%s loss=%.6f nll=%.6f" % (ci, B, n, tau, (posterior, crf, att, exact), loss.item(),
                                                                          nll.item()))
        ci += 1
    cases.update(n_cases=np.int64(ci), trans_s=trans_s, start=np.int64(start),
                 stop=np.int64(stop), unk=np.int64(unk), x_idx=np.int64(x_idx))
    path = os.path.join(GOLD, "kd_loss.npz")
    np.savez_compressed(path, **cases)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
