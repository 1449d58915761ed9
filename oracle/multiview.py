"""TEST INFRASTRUCTURE (never imported by the product path): torch restatement of the multi-view posterior distillation loss.

Reference (restated): FastSequenceTagger._calculate_multi_view_loss, `distill_posterior` branch
(flair/models/sequence_tagger_model.py:2080-2093) = _forward_alg(distill_mode=True) (:1329-1380) + _backward_alg (:1396-1470) of
both views, masked, fed to _calculate_distillation_loss (:2384-2398): kl_div(log_softmax(student / T), softmax(teacher / T)) *
mask * T^2, summed and divided by the number of sentences (use_crf).  Pinned by tests/golden/multiview_kl.npz, captured by
running those reference methods under autograd (oracle/gen_golden_multiview.py).  Differentiable: gradients come from torch
autograd, which is what the reference itself uses."""
import torch


def forward_vars(feats, trans, start):
    """alpha INCLUDING token i's emission, all n positions (the reference scans past the sentence length; masked later)"""
    B, n, T = feats.shape
    fv = torch.full((B, T), -1e12, dtype=feats.dtype)
    fv[:, start] = 0.0
    out = []
    for i in range(n):
        tv = feats[:, i, :, None] + trans[None, :, :] + fv[:, None, :]        # [b, to, from]
        fv = torch.logsumexp(tv, dim=2)
        out.append(fv)
    return torch.stack(out, 1)


def backward_vars(feats, lens, trans, stop):
    """beta EXCLUDING token i's emission; beta_{L-1} = trans[STOP, :]; rows at or past lens[b] are zero"""
    B, n, T = feats.shape
    rows = []
    for b in range(B):
        L = int(lens[b])
        fv = torch.full((T,), -1e12, dtype=feats.dtype)
        fv[stop] = 0.0
        col = [None] * n
        for i in range(L):
            em = torch.zeros(T, dtype=feats.dtype) if i == 0 else feats[b, L - i]
            tv = em[None, :] + trans.t() + fv[None, :]                         # [t, f] = emit[f] + trans[f, t] + fv[f]
            fv = torch.logsumexp(tv, dim=1)
            col[L - 1 - i] = fv
        for i in range(L, n):
            col[i] = torch.zeros(T, dtype=feats.dtype)
        rows.append(torch.stack(col, 0))
    return torch.stack(rows, 0)


def posterior_kl(emit_s, emit_t, trans, lens, tau, start, stop):
    """-> per-sentence loss [B] (the reference returns their sum / B); differentiable w.r.t. emit_s and trans (student side
    only: the teacher's scores are detached, :2090)"""
    B, n, T = emit_s.shape
    mask = (torch.arange(n)[None, :] < torch.as_tensor(lens)[:, None]).to(emit_s.dtype)
    gs = (forward_vars(emit_s, trans, start) + backward_vars(emit_s, lens, trans, stop)) * mask[:, :, None]
    with torch.no_grad():
        gt = (forward_vars(emit_t, trans, start) + backward_vars(emit_t, lens, trans, stop)) * mask[:, :, None]
    kd = torch.nn.functional.kl_div(torch.log_softmax(gs / tau, dim=-1), torch.softmax(gt / tau, dim=-1), reduction="none")
    return (kd * mask[:, :, None]).sum((1, 2)) * tau * tau


def exact_teacher(emit_t, trans, lens, tau, start, stop):
    """the distill_exact branch's teacher (:2065-2080): softmax over tag pairs of (alpha_{i-1}[from] + beta_i[to] + e_i[to] +
    trans[to, from]) / T of the CONTEXT view under the shared transitions; the start score is (e_0 + trans[:, START]) / T and the end
    score trans[STOP, :] / T -- without the backward / forward variables the KD trainer's teacher adds (finetune_trainer.py:1719-1722)"""
    B, n, T = emit_t.shape
    lens = torch.as_tensor(lens)
    fv, bv = forward_vars(emit_t, trans, start), backward_vars(emit_t, lens, trans, stop)
    ss = emit_t[:, :, :, None] + trans[None, None, :, :]
    bm = (torch.arange(max(n - 1, 0))[None, :] < (lens - 1)[:, None]).to(emit_t.dtype)
    pair = ((fv[:, :-1, None, :] + bv[:, 1:, :, None] + ss[:, 1:]) * bm[:, :, None, None] / tau).reshape(B, max(n - 1, 0), T * T).softmax(-1)
    return pair, (emit_t[:, 0] + trans[None, :, start]) / tau, (trans[None, stop, :] / tau).expand(B, T)


def exact_kd(emit_s, emit_t, trans, lens, tau, start, stop):
    """-> per-sentence loss [B] of the distill_exact branch (:2049-2087); the teacher side is detached (:2081-2083)"""
    from . import kd
    with torch.no_grad():
        pair, s_sc, e_sc = exact_teacher(emit_t, trans, lens, tau, start, stop)
    return kd.exact_per_sentence(emit_s, trans, lens, pair, s_sc, e_sc, tau, start, stop)


def l2_term(rep_s, rep_t, lens):
    """calculate_l2_loss (:2026-2035): squared distance of the two views' token representations at the real tokens, summed,
    / B / H -> per-sentence values [B] (the reference returns their sum / B); rep_t is detached (:2027)"""
    B, n, H = rep_s.shape
    mask = (torch.arange(n)[None, :] < torch.as_tensor(lens)[:, None]).to(rep_s.dtype)
    return (((rep_s - rep_t.detach()) ** 2) * mask[:, :, None]).sum((1, 2)) / H
