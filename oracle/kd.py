"""TEST INFRASTRUCTURE (never imported by the product path): torch restatement of the teacher-student knowledge-distillation
losses of the CRF tagger.  Only tests/ use it, as the checker of the HIP path at sizes / shapes the golden file does not hold.

Reference (restated, never copied):
  teacher side  ModelFinetuner.assign_pretrained_teacher_targets (flair/trainers/finetune_trainer.py:1515-1910): per training
                sentence and teacher, (a) the n-best tag sequences + softmax of their scores (:1600, oracle/crf.py:viterbi_nbest),
                (b) `(forward_var + backward_var) * mask` of the teacher's CRF after the START / STOP / <unk> logits were lowered by
                1e12 (:1627-1634), (c) for distill_exact the softmax over tag PAIRS of (alpha_{i-1}[f] + beta_i[t] + e_i[t] +
                trans[t,f]) / T plus start / end scores (:1705-1722,1885).
  student side  FastSequenceTagger.simple_forward_distillation_loss (flair/models/sequence_tagger_model.py:2110-2372):
                interpolation * (posterior + crf + exact) + (1 - interpolation) * NLL, each term as restated below.

Pinned by tests/golden/kd_loss.npz (+ kd_emission.npz for the emission-level term), captured by running those reference methods
under autograd (oracle/gen_golden_kd.py, oracle/gen_golden_kd_emission.py);
tests/test_oracle_golden.py checks every function here against it.  Differentiable through torch autograd, which is what the
reference itself uses."""
import torch

from . import multiview as mv
from .train_step import crf_nll_torch


def lengths_mask(lens, n, dtype):
    return (torch.arange(n)[None, :] < torch.as_tensor(lens)[:, None]).to(dtype)


def suppressed(logits, idxs):
    out = logits.clone()
    for i in idxs:          # applied one index at a time like the reference: an index listed twice is lowered twice
        out[:, :, i] -= 1e12
    return out


def teacher_fb_score(logits, trans_t, lens, start, stop, suppress):
    """finetune_trainer.py:1627-1634 -> [B, n, T]"""
    lg = suppressed(logits, suppress)
    mask = lengths_mask(lens, lg.shape[1], lg.dtype)
    return (mv.forward_vars(lg, trans_t, start) + mv.backward_vars(lg, lens, trans_t, stop)) * mask[:, :, None]


def teacher_pair_posterior(logits, trans_t, lens, start, stop, suppress, tau):
    """finetune_trainer.py:1705-1722,1885 -> (pair [B, n-1, T*T] softmax over (to, from), start_score [B,T], end_score [B,T])"""
    lg = suppressed(logits, suppress)
    B, n, T = lg.shape
    lens = torch.as_tensor(lens)
    fv, bv = mv.forward_vars(lg, trans_t, start), mv.backward_vars(lg, lens, trans_t, stop)
    ss = lg[:, :, :, None] + trans_t[None, None, :, :]
    bm = lengths_mask(lens - 1, n - 1, lg.dtype) if n > 1 else torch.zeros((B, 0), dtype=lg.dtype)
    pair = (fv[:, :-1, None, :] + bv[:, 1:, :, None] + ss[:, 1:]) * bm[:, :, None, None] / tau
    s_sc = (ss[:, 0, :, start] + bv[:, 0, :]) / tau
    e_sc = (trans_t[None, stop, :] + fv[torch.arange(B), lens - 1, :]) / tau
    return pair.reshape(B, max(n - 1, 0), T * T).softmax(-1), s_sc, e_sc


def posterior_term(es, trans, lens, scores_t, tau, start, stop):
    """:2120-2136: mean over teachers of sum_b sum_i T^2 KL(softmax(teacher score / T) || softmax(student alpha+beta / T)) / B.
    scores_t: list of [B, n, T] (one per teacher)"""
    B, n, T = es.shape
    mask = lengths_mask(lens, n, es.dtype)
    gs = (mv.forward_vars(es, trans, start) + mv.backward_vars(es, lens, trans, stop)) * mask[:, :, None]
    tot = 0.0
    for st in scores_t:
        kd = torch.nn.functional.kl_div(torch.log_softmax(gs / tau, dim=-1), torch.softmax(st.to(es.dtype) / tau, dim=-1),
                                        reduction="none")
        tot = tot + (kd * mask[:, :, None]).sum() * tau * tau / B
    return tot / len(scores_t)


def crf_term(es, trans, lens, targets, start, stop, weights=None, att_nums=None):
    """:2249-2309: NLL of each of the K teacher paths (targets int [B, n, K]); mean over B*K, or with crf_attention
    sum(nll * weights [B, K]) / att_nums (att_nums = number of (sentence, teacher) pairs)"""
    B, n, K = targets.shape
    tot = 0.0
    for k in range(K):
        nll = crf_nll_torch(es, targets[:, :, k], lens, trans, start, stop)
        tot = tot + ((nll * weights[:, k].to(es.dtype)).sum() / att_nums if weights is not None else nll.sum() / (B * K))
    return tot


def gold_reweighted(att, targets, tags, lens, gold_const, exp_score, att_nums):
    """distill_with_gold (:2289-2304): att [B,K] * gold_const / (errors + gold_const) (or exp(-errors / gold_const)), errors =
    tokens where the path disagrees with the gold tags; renormalised per sentence to att_nums / B"""
    B, n, K = targets.shape
    mask = lengths_mask(lens, n, torch.float32)
    num_error = ((targets != torch.as_tensor(tags)[:, :, None]).float() * mask[:, :, None]).sum(1)
    sw = torch.exp(-num_error / gold_const) if exp_score else gold_const / (num_error + gold_const)
    att = att * sw.to(att.dtype)
    return att / att.sum(-1, keepdim=True) * (att_nums / B)


def exact_term(es, trans, lens, pair, s_sc, e_sc, tau, start, stop):
    """:2139-2244 + _calculate_xstruct_distillation_loss (:2400-2425): sum over sentences / B"""
    return exact_per_sentence(es, trans, lens, pair, s_sc, e_sc, tau, start, stop).sum() / es.shape[0]


def exact_per_sentence(es, trans, lens, pair, s_sc, e_sc, tau, start, stop):
    """-(E_teacher[score / T] - logZ_T) * T^2 per sentence, negative values replaced by 0 (a constant: no gradient)"""
    B, n, T = es.shape
    lens = torch.as_tensor(lens)
    # tempered partition: the plain recursion on es / T, trans / T (:1348-1350)
    fv = mv.forward_vars(es / tau, trans / tau, start)
    logz = torch.logsumexp(fv[torch.arange(B), lens - 1, :] + trans[None, stop, :] / tau, dim=1)
    struct = (es[:, :, :, None] + trans[None, None, :, :]).reshape(B, n, T * T)[:, 1:]
    bm = lengths_mask(lens - 1, n - 1, es.dtype) if n > 1 else torch.zeros((B, 0), dtype=es.dtype)
    start_score = es[:, 0] + trans[None, :, start]
    end_score = trans[None, stop, :]
    sp, ep = s_sc.to(es.dtype).softmax(-1), e_sc.to(es.dtype).softmax(-1)
    ends = (sp * start_score / tau + ep * end_score / tau).sum(-1)
    if n > 1:
        expect = (pair.to(es.dtype) * struct / tau * bm[:, :, None]).sum((-1, -2)) + ends
    else:
        expect = ends
    kd = -((expect - logz) * tau * tau)
    return torch.where(kd < 0, torch.zeros_like(kd), kd)


def target_term(es, trans, lens, tags, start, stop, x_idx=None):
    """_calculate_loss (:2426-2506): NLL over the tokens whose gold tag is not S-X (remove_x), mean over sentences"""
    B, n, T = es.shape
    lens = torch.as_tensor(lens)
    tags = torch.as_tensor(tags)
    keep = lengths_mask(lens, n, torch.float32).bool()
    if x_idx is not None:
        keep = keep & (tags != x_idx)
    kl = keep.sum(1)
    nc = max(1, int(kl.max()))
    ce = torch.zeros((B, nc, T), dtype=es.dtype)
    ct = torch.zeros((B, nc), dtype=torch.int64)
    for b in range(B):
        idx = torch.nonzero(keep[b])[:, 0]
        ce[b, :len(idx)] = es[b, idx]
        ct[b, :len(idx)] = tags[b, idx]
    return crf_nll_torch(ce, ct, kl, trans, start, stop).mean()


def emission_term(es, lens, teacher, tau, teacher_is_prob=False):
    """distill_emission (:2311-2365 -> _calculate_distillation_loss :2384-2398, use_crf: sum / batch size):
    T^2 sum_b sum_{i < len_b} KL(p_teacher || softmax(es / T)) / B with p_teacher = softmax(teacher / T), or `teacher` itself when
    the trainer stored probabilities (distill_prob).  teacher [B, n, T]: the mean over the teachers of what
    assign_pretrained_teacher_predictions stored (zero rows behind a sentence's end), or -- distill_posterior also on -- the
    first teacher's forward-backward scores.  Pinned by tests/golden/kd_emission.npz (oracle/gen_golden_kd_emission.py)."""
    B, n, T = es.shape
    mask = lengths_mask(lens, n, es.dtype)
    tp = teacher.to(es.dtype) if teacher_is_prob else torch.softmax(teacher.to(es.dtype) / tau, dim=-1)
    kd = torch.nn.functional.kl_div(torch.log_softmax(es / tau, dim=-1), tp, reduction="none")
    return (kd * mask[:, :, None]).sum() * tau * tau / B


def kd_loss(es, trans, lens, tags, start, stop, x_idx, tau, interpolation, scores_t=None, targets=None, weights=None,
            att_nums=None, exact=None, emission=None, emission_is_prob=False):
    """simple_forward_distillation_loss for a CRF student (:2372)"""
    kd = 0.0
    if emission is not None:
        kd = kd + emission_term(es, lens, emission, tau, emission_is_prob)
    if scores_t:
        kd = kd + posterior_term(es, trans, lens, scores_t, tau, start, stop)
    if exact is not None:
        kd = kd + exact_term(es, trans, lens, exact[0], exact[1], exact[2], tau, start, stop)
    if targets is not None:
        kd = kd + crf_term(es, trans, lens, targets, start, stop, weights, att_nums)
    return interpolation * kd + (1.0 - interpolation) * target_term(es, trans, lens, tags, start, stop, x_idx)
