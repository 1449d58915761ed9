"""Typed Python wrappers over the C ABI: argument checking + pointer marshalling, nothing else.
Every function enqueues on torch's current HIP stream and returns immediately."""
import ctypes

import torch

from . import lib as L
from .lib import c_void_p, ptr, stream_ptr

BF16 = torch.bfloat16
F32 = torch.float32
I32 = torch.int32


NO_DROP = (0, 0)  # (seed, thresh): thresh = 0 disables a dropout site


def drop_thresh(p):
    """dropout probability -> the 32-bit threshold the kernels compare against (include/kbner.h, dropout section)"""
    if not 0.0 <= p < 1.0:
        raise L.KbnerError("dropout probability must be in [0, 1)")
    return min(int(round(p * 4294967296.0)), 0xFFFFFFFF)


def dropout_mask(Z, M, N, seed, thresh, device="cuda"):
    """tests: the multiplier (0 or 1/(1-p)) a site applies, f32[Z,M,N]"""
    out = torch.empty((Z, M, N), dtype=F32, device=device)
    L.call("kbner_dropout_mask", ptr(out), Z, M, N, seed, thresh, stream_ptr())
    return out


def _chk(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise L.KbnerError("%s must be a contiguous cuda %s tensor (got %s, cuda=%s, contiguous=%s)"
                           % (name, dtype, t.dtype, t.is_cuda, t.is_contiguous()))


# ---------------------------------------------------------------- CRF
def crf_viterbi(emit, trans, lens, start, stop, want_popped=False):
    """emit f32[B,n,T], trans f32[T,T], lens i32[B] -> tags i32[B,n], conf f32[B,n] (, popped i32[B])"""
    _chk(emit, F32, "emit"); _chk(trans, F32, "trans"); _chk(lens, I32, "lens")
    B, n, T = emit.shape
    tags = torch.empty((B, n), dtype=I32, device=emit.device)
    conf = torch.empty((B, n), dtype=F32, device=emit.device)
    popped = torch.empty((B,), dtype=I32, device=emit.device) if want_popped else None
    L.call("kbner_crf_viterbi", ptr(emit), ptr(trans), ptr(lens), B, n, T, start, stop, ptr(tags), ptr(conf), ptr(popped),
           stream_ptr())
    return (tags, conf, popped) if want_popped else (tags, conf)


def crf_nll_fwd(emit, trans, tags, lens, start, stop):
    _chk(emit, F32, "emit"); _chk(trans, F32, "trans"); _chk(tags, I32, "tags"); _chk(lens, I32, "lens")
    B, n, T = emit.shape
    logz = torch.empty((B,), dtype=F32, device=emit.device)
    gold = torch.empty((B,), dtype=F32, device=emit.device)
    alpha = torch.empty((B, n + 1, T), dtype=F32, device=emit.device)
    L.call("kbner_crf_nll_fwd", ptr(emit), ptr(trans), ptr(tags), ptr(lens), B, n, T, start, stop, ptr(logz), ptr(gold),
           ptr(alpha), stream_ptr())
    return logz, gold, alpha


def crf_nll_bwd(emit, trans, tags, lens, alpha, logz, dloss, start, stop, dtrans):
    """-> demit f32[B,n,T]; dtrans f32[T,T] is accumulated into (+=)."""
    _chk(dloss, F32, "dloss"); _chk(dtrans, F32, "dtrans")
    B, n, T = emit.shape
    demit = torch.empty_like(emit)
    L.call("kbner_crf_nll_bwd", ptr(emit), ptr(trans), ptr(tags), ptr(lens), ptr(alpha), ptr(logz), ptr(dloss), B, n, T, start,
           stop, ptr(demit), ptr(dtrans), stream_ptr())
    return demit


def crf_viterbi_nbest(emit, trans, lens, start, stop, nbest):
    """emit f32[B,n,T] -> (path_score f32[B,nbest], decode i32[B,n,nbest]) -- _viterbi_decode_nbest"""
    _chk(emit, F32, "emit"); _chk(trans, F32, "trans"); _chk(lens, I32, "lens")
    B, n, T = emit.shape
    ws = torch.empty(max(1, L.load().kbner_crf_viterbi_nbest_ws_bytes(B, n, T, nbest) // 2), dtype=torch.int16, device=emit.device)
    decode = torch.zeros((B, n, nbest), dtype=I32, device=emit.device)
    score = torch.zeros((B, nbest), dtype=F32, device=emit.device)
    L.call("kbner_crf_viterbi_nbest", ptr(emit), ptr(trans), ptr(lens), B, n, T, start, stop, nbest, ptr(ws), ptr(decode), ptr(score),
           stream_ptr())
    return score, decode


def crf_posterior(emit, trans, lens, start, stop):
    """token marginals f32[B,n,T] (zero rows past lens) -- _obtain_labels' predict_posterior branch"""
    _chk(emit, F32, "emit"); _chk(trans, F32, "trans"); _chk(lens, I32, "lens")
    B, n, T = emit.shape
    zeros = torch.zeros((B, n), dtype=I32, device=emit.device)
    logz, _, alpha = crf_nll_fwd(emit, trans, zeros, lens, start, stop)
    marg = torch.empty_like(emit)
    L.call("kbner_crf_posterior", ptr(emit), ptr(trans), ptr(lens), ptr(alpha), ptr(logz), B, n, T, start, stop, ptr(marg),
           stream_ptr())
    return marg


def crf_posterior_kl(emit_s, emit_t, trans, lens, weights, tau, start, stop, dtrans):
    """multi-view posterior distillation (FastSequenceTagger._calculate_multi_view_loss, distill_posterior branch):
    -> (loss f32[B] = tau^2 * sum_i KL(teacher || student tempered marginals), demit f32[B,n,T] = d(sum_b weights[b] loss[b]) /
    d emit_s); the transition gradient is ADDED to dtrans f32[T,T].  emit_t is a constant (the detached context view)."""
    _chk(emit_s, F32, "emit_s"); _chk(emit_t, F32, "emit_t"); _chk(trans, F32, "trans"); _chk(lens, I32, "lens")
    _chk(weights, F32, "weights"); _chk(dtrans, F32, "dtrans")
    if emit_s.shape != emit_t.shape:
        raise L.KbnerError("the two views must have the same [B, n, T] emissions shape")
    B, n, T = emit_s.shape
    loss = torch.empty((B,), dtype=F32, device=emit_s.device)
    demit = torch.empty_like(emit_s)
    ws = torch.empty((int(L.load().kbner_crf_posterior_kl_ws_floats(B, n, T)),), dtype=F32, device=emit_s.device)
    L.call("kbner_crf_posterior_kl", ptr(emit_s), ptr(emit_t), ptr(trans), ptr(lens), ptr(weights), float(tau), B, n, T, start, stop,
           ptr(loss), ptr(demit), ptr(dtrans), ptr(ws), stream_ptr())
    return loss, demit


def crf_fb_score(emit, trans, lens, start, stop, suppress=()):
    """teacher forward-backward scores f32[B,n,T] (alpha + beta below lens, 0 past it); the emissions of the tags in `suppress`
    are lowered by 1e12 first -- the `distill_posterior` target (finetune_trainer.py:1627-1634)"""
    _chk(emit, F32, "emit"); _chk(trans, F32, "trans"); _chk(lens, I32, "lens")
    B, n, T = emit.shape
    score = torch.empty_like(emit)
    bits = 0
    for t in suppress:
        bits |= 1 << int(t)
    L.call("kbner_crf_fb_score", ptr(emit), ptr(trans), ptr(lens), bits, B, n, T, start, stop, ptr(score), stream_ptr())
    return score


def crf_posterior_kl_scores(emit_s, score_t, trans, lens, weights, tau, start, stop, dtrans):
    """teacher-student posterior distillation (simple_forward_distillation_loss, `distill_posterior`): as crf_posterior_kl with the
    teacher given as its forward-backward scores (crf_fb_score under the teacher's own transitions)"""
    _chk(emit_s, F32, "emit_s"); _chk(score_t, F32, "score_t"); _chk(trans, F32, "trans"); _chk(lens, I32, "lens")
    _chk(weights, F32, "weights"); _chk(dtrans, F32, "dtrans")
    if emit_s.shape != score_t.shape:
        raise L.KbnerError("student emissions and teacher scores must have the same [B, n, T] shape")
    B, n, T = emit_s.shape
    loss = torch.empty((B,), dtype=F32, device=emit_s.device)
    demit = torch.empty_like(emit_s)
    ws = torch.empty((int(L.load().kbner_crf_posterior_kl_ws_floats(B, n, T)),), dtype=F32, device=emit_s.device)
    L.call("kbner_crf_posterior_kl_scores", ptr(emit_s), ptr(score_t), ptr(trans), ptr(lens), ptr(weights), float(tau), B, n, T,
           start, stop, ptr(loss), ptr(demit), ptr(dtrans), ptr(ws), stream_ptr())
    return loss, demit


def emission_kl(emit, teacher, lens, weights, tau, teacher_is_prob=False):
    """`distill_emission` term of a CRF student: (loss f32[B] = tau^2 sum_tokens KL(p_teacher || softmax(emit / tau)),
    demit f32[B,n,T] = d(sum_b weights[b] loss[b]) / d emit)"""
    _chk(emit, F32, "emit"); _chk(teacher, F32, "teacher"); _chk(lens, I32, "lens"); _chk(weights, F32, "weights")
    if emit.shape != teacher.shape:
        raise L.KbnerError("student emissions and teacher predictions must have the same [B, n, T] shape")
    B, n, T = emit.shape
    loss = torch.empty((B,), dtype=F32, device=emit.device)
    demit = torch.empty_like(emit)
    L.call("kbner_emission_kl", ptr(emit), ptr(teacher), ptr(lens), ptr(weights), float(tau), int(bool(teacher_is_prob)), B, n, T,
           ptr(loss), ptr(demit), stream_ptr())
    return loss, demit


def softmax_ce(emit, tags, lens, weights):
    """softmax head (use_crf=False): (loss f32[B] = sum over the sentence's tokens of the cross entropy, unweighted;
    demit f32[B,n,T] = d(sum_b weights[b] loss[b]) / d emit)"""
    _chk(emit, F32, "emit"); _chk(tags, I32, "tags"); _chk(lens, I32, "lens"); _chk(weights, F32, "weights")
    B, n, T = emit.shape
    loss = torch.empty((B,), dtype=F32, device=emit.device)
    demit = torch.empty_like(emit)
    L.call("kbner_softmax_ce", ptr(emit), ptr(tags), ptr(lens), ptr(weights), B, n, T, ptr(loss), ptr(demit), stream_ptr())
    return loss, demit


def softmax_decode(emit, lens, want_dist=False):
    """softmax head decode: (tags i32[B,n], conf f32[B,n][, dist f32[B,n,T]])"""
    _chk(emit, F32, "emit"); _chk(lens, I32, "lens")
    B, n, T = emit.shape
    tags = torch.empty((B, n), dtype=I32, device=emit.device)
    conf = torch.empty((B, n), dtype=F32, device=emit.device)
    dist = torch.empty((B, n, T), dtype=F32, device=emit.device) if want_dist else None
    L.call("kbner_softmax_decode", ptr(emit), ptr(lens), B, n, T, ptr(tags), ptr(conf), ptr(dist), stream_ptr())
    return (tags, conf, dist) if want_dist else (tags, conf)


def crf_pair_posterior(emit, trans, lens, tau, start, stop, suppress=()):
    """teacher targets of `distill_exact` (finetune_trainer.py:1705-1722): (pair f32[B,n-1,T*T], start_score f32[B,T],
    end_score f32[B,T])"""
    _chk(emit, F32, "emit"); _chk(trans, F32, "trans"); _chk(lens, I32, "lens")
    B, n, T = emit.shape
    pair = torch.empty((B, max(n - 1, 0), T * T), dtype=F32, device=emit.device)
    s_sc = torch.empty((B, T), dtype=F32, device=emit.device)
    e_sc = torch.empty((B, T), dtype=F32, device=emit.device)
    ws = torch.empty((max(1, int(L.load().kbner_crf_pair_ws_floats(B, n, T))),), dtype=F32, device=emit.device)
    bits = 0
    for t in suppress:
        bits |= 1 << int(t)
    L.call("kbner_crf_pair_posterior", ptr(emit), ptr(trans), ptr(lens), bits, float(tau), B, n, T, start, stop,
           ptr(pair) if n > 1 else None, ptr(s_sc), ptr(e_sc), ptr(ws), stream_ptr())
    return pair, s_sc, e_sc


def crf_exact_kd(emit, trans, lens, pair, start_score, end_score, weights, tau, start, stop, dtrans):
    """student loss of `distill_exact` (sequence_tagger_model.py:2139-2244,2400-2425) -> (loss f32[B], demit f32[B,n,T]);
    dtrans f32[T,T] is accumulated into"""
    _chk(emit, F32, "emit"); _chk(trans, F32, "trans"); _chk(lens, I32, "lens"); _chk(start_score, F32, "start_score")
    _chk(end_score, F32, "end_score"); _chk(weights, F32, "weights"); _chk(dtrans, F32, "dtrans")
    B, n, T = emit.shape
    if n > 1:
        _chk(pair, F32, "pair")
        if tuple(pair.shape) != (B, n - 1, T * T):
            raise L.KbnerError("pair must be [B, n-1, T*T] = %s, got %s" % ((B, n - 1, T * T), tuple(pair.shape)))
    loss = torch.empty((B,), dtype=F32, device=emit.device)
    demit = torch.empty_like(emit)
    ws = torch.empty((max(1, int(L.load().kbner_crf_pair_ws_floats(B, n, T))),), dtype=F32, device=emit.device)
    L.call("kbner_crf_exact_kd", ptr(emit), ptr(trans), ptr(lens), ptr(pair) if n > 1 else None, ptr(start_score), ptr(end_score),
           ptr(weights), float(tau), B, n, T, start, stop, ptr(loss), ptr(demit), ptr(dtrans), ptr(ws), stream_ptr())
    return loss, demit


# ---------------------------------------------------------------- rows / head
def gather_rows(src, idx, out=None):
    _chk(src, BF16, "src"); _chk(idx, I32, "idx")
    R, H = idx.numel(), src.shape[-1]
    if out is None:
        out = torch.empty((R, H), dtype=BF16, device=src.device)
    L.call("kbner_gather_rows", ptr(src), ptr(idx), ptr(out), R, H, stream_ptr())
    return out


def gather_rows_into(src, idx, out2d, col, H):
    """out2d[:, col:col+H] = rows of src picked by idx (-1 -> zeros); out2d bf16 [R, ld] (col % 8 == 0)"""
    _chk(src, BF16, "src"); _chk(idx, I32, "idx"); _chk(out2d, BF16, "out2d")
    if col % 8 or idx.numel() > out2d.shape[0]:
        raise L.KbnerError("gather_rows_into: bad column offset / row count")
    L.call("kbner_gather_rows_ld", ptr(src), src.shape[-1], ptr(idx), c_void_p(out2d.data_ptr() + 2 * col), out2d.shape[-1],
           idx.numel(), H, stream_ptr())


def gather_rows_f32(src, idx):
    """src f32 [Rsrc, W], idx i32 [R] (-1 -> zero row) -> f32 [R, W]"""
    _chk(src, F32, "src"); _chk(idx, I32, "idx")
    R, W = idx.numel(), src.shape[-1]
    out = torch.empty((R, W), dtype=F32, device=src.device)
    L.call("kbner_gather_rows_f32", ptr(src), ptr(idx), ptr(out), R, W, stream_ptr())
    return out


def scatter_rows_f32(rows, idx, dst):
    """dst f32[V, W][idx[r], :] = rows f32[R, W][r, :] (unique indices; idx < 0 skipped)"""
    _chk(rows, F32, "rows"); _chk(idx, I32, "idx"); _chk(dst, F32, "dst")
    L.call("kbner_scatter_rows_f32", ptr(rows), ptr(idx), ptr(dst), idx.numel(), rows.shape[-1], stream_ptr())


def scatter_add_rows_f32(rows, idx, dst):
    """dst[idx[r],:] += rows[r,:] (idx[r] < 0 skipped; indices unique), fp32, any width"""
    _chk(rows, F32, "rows"); _chk(idx, I32, "idx"); _chk(dst, F32, "dst")
    R, W = rows.shape
    if idx.numel() != R or dst.shape[-1] != W:
        raise L.KbnerError("scatter_add_rows_f32: rows [R,W], idx [R], dst [*,W]")
    L.call("kbner_scatter_add_rows_f32", ptr(rows), ptr(idx), ptr(dst), R, W, stream_ptr())
    return dst


def l2_rows(a, b, w, loss, da=None, gscale=1.0):
    """loss[0] += sum_r w[r] |a[r] - b[r]|^2; da[r] += 2 gscale w[r] (a[r] - b[r]) -- multi-view calculate_l2_loss (a, b, da bf16 [R,H])"""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(w, F32, "w"); _chk(loss, F32, "loss")
    if da is not None:
        _chk(da, BF16, "da")
    R, H = a.shape
    if tuple(b.shape) != (R, H) or w.numel() != R or (da is not None and tuple(da.shape) != (R, H)):
        raise L.KbnerError("l2_rows: a, b, da [R,H], w [R]")
    L.call("kbner_l2_rows", ptr(a), ptr(b), ptr(w), float(gscale), ptr(da), ptr(loss), R, H, stream_ptr())
    return loss


def scatter_rows(dout, idx, dsrc):
    _chk(dout, BF16, "dout"); _chk(idx, I32, "idx"); _chk(dsrc, BF16, "dsrc")
    L.call("kbner_scatter_rows", ptr(dout), ptr(idx), ptr(dsrc), idx.numel(), dout.shape[-1], stream_ptr())


def head_fwd(x, w, bias):
    _chk(x, BF16, "x"); _chk(w, F32, "w"); _chk(bias, F32, "bias")
    R, H = x.shape
    T = w.shape[0]
    out = torch.empty((R, T), dtype=F32, device=x.device)
    L.call("kbner_head_fwd", ptr(x), ptr(w), ptr(bias), ptr(out), R, H, T, stream_ptr())
    return out


def head_bwd(de, x, w, dw, db):
    """de f32[R,T] -> dx bf16[R,H]; dw f32[T,H], db f32[T] accumulated into."""
    _chk(de, F32, "de"); _chk(x, BF16, "x"); _chk(w, F32, "w"); _chk(dw, F32, "dw"); _chk(db, F32, "db")
    R, H = x.shape
    T = w.shape[0]
    dx = torch.empty((R, H), dtype=BF16, device=x.device)
    L.call("kbner_head_bwd_dx", ptr(de), ptr(w), ptr(dx), R, H, T, stream_ptr())
    L.call("kbner_head_bwd_dw", ptr(de), ptr(x), ptr(dw), ptr(db), R, H, T, stream_ptr())
    return dx


def gemm_tile_rows(layout, M, N):
    """output-tile height (256 or 128) the grouped kernel picks for one (layout, M, N) problem with the static tile walk"""
    return int(L.load().kbner_gemm_tile_rows(layout, M, N))


def colsum_rows_f32(ws, rows, out):
    """out f32[N] += sum over the first `rows` rows of ws f32[>= rows, N] (the EPI_COLSUM_WS workspace)"""
    _chk(ws, F32, "ws"); _chk(out, F32, "out")
    L.call("kbner_colsum_rows_f32", ptr(ws), rows, out.numel(), ptr(out), stream_ptr())


def colsum(x, out, M=None):
    _chk(x, BF16, "x"); _chk(out, F32, "out")
    rows, N = x.shape
    L.call("kbner_colsum", ptr(x), ptr(out), rows if M is None else M, N, N, stream_ptr())


# ---------------------------------------------------------------- LayerNorm / embeddings
HBM_HOOK = None  # bench.py sets this to a list: (kernel name, event, event, algorithmic bytes) per launch of an HBM-bound kernel


class _hbm_timed:
    """HIP events on the launch stream around one launch of an HBM-bound kernel, when bench.py asked for them (HBM_HOOK)"""

    def __init__(self, name, nbytes):
        self.name, self.nbytes, self.hook = name, nbytes, HBM_HOOK

    def __enter__(self):
        if self.hook is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.hook is not None:
            self.e1.record()
            self.hook.append((self.name, self.e0, self.e1, float(self.nbytes)))
        return False


def ln_fwd(h, gamma, beta, eps, y, mean, rstd):
    M, H = h.shape
    with _hbm_timed("ln_fwd", 4 * M * H):   # read h, write y (bf16): DESIGN.md section 3
        L.call("kbner_ln_fwd", ptr(h), ptr(gamma), ptr(beta), eps, ptr(y), ptr(mean), ptr(rstd), M, H, stream_ptr())


_LN_WS = {}


def ln_ws(H, device):
    """scratch for the LayerNorm-backward column sums (one buffer per (H, device); launches on one stream serialise)"""
    key = (H, str(device))
    if key not in _LN_WS:
        _LN_WS[key] = torch.empty(L.load().kbner_ln_bwd_ws_floats(H), dtype=F32, device=device)
    return _LN_WS[key]


def ln_fwd_slabs(ws, splits, bias, addend, drop, h, gamma, beta, eps, y, mean, rstd):
    """ln_fwd of h = bf16(dropout(sum_s ws[s] + bias) + addend) -- the fold kbner_splitk_finish does, same bits -- which is also stored"""
    M, H = h.shape
    _chk(ws, F32, "ws")
    L.call("kbner_ln_fwd_slabs", ptr(ws), splits, ptr(bias), ptr(addend), addend.shape[1] if addend is not None else 0, drop[0], drop[1],
           ptr(h), ptr(gamma), ptr(beta), eps, ptr(y), ptr(mean), ptr(rstd), M, H, stream_ptr())


def ln_bwd(dy, h, mean, rstd, gamma, dh, dgamma, dbeta, dbias=None, dhm=None, drop=NO_DROP, defer_ws=None, dy_slabs=None):
    """drop=(seed, thresh) with thresh != 0: also writes dhm = mask * dh / (1-p) (the dY of the GEMM behind the dropout).
    defer_ws (f32, >= ln_bwd_blocks(M) * 3 H): the per-block partial column sums stay there instead of being reduced into dgamma /
    dbeta / dbias by a launch of their own -- the caller reduces a whole backward pass's worth with ln_colreduce_batched."""
    M, H = h.shape
    if dy_slabs is not None:   # (ws f32[splits, M, H], splits, addend): dy = bf16(sum_s ws[s] + addend), folded on the way in
        sws, splits, sadd = dy_slabs
        _chk(sws, F32, "dy_slabs")
        deferred = defer_ws is not None
        L.call("kbner_ln_bwd_slabs", ptr(sws), splits, ptr(sadd), sadd.shape[1] if sadd is not None else 0, ptr(h), ptr(mean), ptr(rstd),
               ptr(gamma), ptr(dh), None if deferred else ptr(dgamma), None if deferred else ptr(dbeta), None if deferred else ptr(dbias),
               ptr(defer_ws if deferred else ln_ws(H, h.device)), M, H, ptr(dhm), drop[0], drop[1], stream_ptr())
        return
    with _hbm_timed("ln_bwd", (8 if drop[1] else 6) * M * H):   # read dy, h; write dh (+ dhm with dropout)
        if defer_ws is not None:
            L.call("kbner_ln_bwd", ptr(dy), ptr(h), ptr(mean), ptr(rstd), ptr(gamma), ptr(dh), None, None, None,
                   ptr(defer_ws), M, H, ptr(dhm), drop[0], drop[1], stream_ptr())
        else:
            L.call("kbner_ln_bwd", ptr(dy), ptr(h), ptr(mean), ptr(rstd), ptr(gamma), ptr(dh), ptr(dgamma), ptr(dbeta), ptr(dbias),
                   ptr(ln_ws(H, h.device)), M, H, ptr(dhm), drop[0], drop[1], stream_ptr())


def ln_bwd_blocks(M):
    return int(L.load().kbner_ln_bwd_blocks(int(M)))


def _host_items(records):
    import ctypes
    flat = [int(x) for r in records for x in r]
    return (ctypes.c_longlong * len(flat))(*flat)


def ln_colreduce_batched(records, H):
    """records: (ws, dgamma, dbeta, dbias or None, partial rows) per LayerNorm -- one launch adds all their column sums"""
    items = _host_items([(w.data_ptr(), dg.data_ptr(), db.data_ptr(), dbi.data_ptr() if dbi is not None else 0, nb)
                         for w, dg, db, dbi, nb in records])
    L.call("kbner_ln_colreduce_batched", items, len(records), H, stream_ptr())


def colsum_rows_f32_batched(records, N):
    """records: (ws f32[rows, N], out f32[N], rows): out += column sums, every record in one launch"""
    items = _host_items([(w.data_ptr(), o.data_ptr(), rows) for w, o, rows in records])
    L.call("kbner_colsum_rows_f32_batched", items, len(records), N, stream_ptr())


def embed_ln_fwd(ids, pos_ids, word, pos, type0, gamma, beta, eps, h0, y, mean, rstd, drop=NO_DROP):
    _chk(ids, I32, "ids"); _chk(pos_ids, I32, "pos_ids"); _chk(word, F32, "word")
    M, H = ids.numel(), word.shape[1]
    L.call("kbner_embed_ln_fwd", ptr(ids), ptr(pos_ids), ptr(word), ptr(pos), ptr(type0), ptr(gamma), ptr(beta), eps, ptr(h0),
           ptr(y), ptr(mean), ptr(rstd), M, H, drop[0], drop[1], stream_ptr())


def embed_ln_bwd(dy, h0, mean, rstd, gamma, ids, pos_ids, dgamma, dbeta, dword, dpos, dtype0, drop=NO_DROP, row_flags=None, defer_ws=None):
    """row_flags (u8 per row of dword): set to LIVE | TOUCHED by the kernel for every row it adds a gradient to (instead of a mark_rows
    launch); defer_ws: the partial column sums stay there for ln_colreduce_batched (dgamma / dbeta / dtype0 are not touched)."""
    M, H = ids.numel(), dword.shape[1]
    if row_flags is not None:
        _chk(row_flags, torch.uint8, "row_flags")
    if row_flags is None and defer_ws is None:
        L.call("kbner_embed_ln_bwd", ptr(dy), ptr(h0), ptr(mean), ptr(rstd), ptr(gamma), ptr(ids), ptr(pos_ids), ptr(dgamma),
               ptr(dbeta), ptr(dword), ptr(dpos), ptr(dtype0), ptr(ln_ws(H, dy.device)), M, H, drop[0], drop[1], stream_ptr())
        return
    deferred = defer_ws is not None
    L.call("kbner_embed_ln_bwd_mark", ptr(dy), ptr(h0), ptr(mean), ptr(rstd), ptr(gamma), ptr(ids), ptr(pos_ids),
           None if deferred else ptr(dgamma), None if deferred else ptr(dbeta), ptr(dword), ptr(dpos), None if deferred else ptr(dtype0),
           ptr(defer_ws if deferred else ln_ws(H, dy.device)), ptr(row_flags), M, H, drop[0], drop[1], stream_ptr())


# ---------------------------------------------------------------- GEMM
FORCE_128 = False  # tests: force the 128^2 kernel
# below this many 256x256 tiles a GEMM goes to the 128x128 kernel (4x the tiles): a small micro-batch leaves most of the
# 256 CUs idle on the big tile
# (100 until round 5; 32 since the 128-row tiles of the 256-path run on the deep-ring kernel, csrc/gemm256.hip gemm128r_kernel:
# at 4 sentences per step 13.54 -> 12.97 ms, same box, alternating)
MIN_TILES_256 = int(__import__("os").environ.get("KBNER_MIN_TILES_256", "32"))


def uses_256(M, N, occupancy=False):
    """whether gemm() runs an (M, N) problem on the 256x256 persistent kernel; occupancy=True (the engine's forward / dgrad
    calls) additionally sends problems with fewer than MIN_TILES_256 big tiles to the 128x128 kernel"""
    ok = M % 256 == 0 and N % 256 == 0 and not FORCE_128
    return ok and (not occupancy or (M // 256) * (N // 256) >= MIN_TILES_256)
GEMM_HOOK = None  # bench.py sets this to a list to time every GEMM launch with HIP events on the launch stream
# Dynamic tile scheduling of the 256x256 kernel (data-parallel steps, see include/kbner.h): a ring of 8-int counter slots,
# int32 [slots, 8], zeroed by its owner (the engine, once per micro-batch); every grouped launch takes the next slot.
SCHED_RING = None
SCHED_ACTIVE = True   # with a ring installed: draw tiles dynamically NOW (sched_active); False = static launches for the moment
_sched_pos = 0


def sched_ring_reset(ring, active=True):
    """make `ring` (or None) the current scheduler ring; the caller has zeroed it on the current stream"""
    global SCHED_RING, SCHED_ACTIVE, _sched_pos
    SCHED_RING, SCHED_ACTIVE, _sched_pos = ring, bool(active), 0


def sched_active(on):
    """data-parallel steps: the GEMMs of a micro-batch draw their tiles dynamically only from the moment its first gradient
    bucket's all-reduce is enqueued (CUs may then be held by RCCL's kernels); before that they are static launches, which take the
    faster interleaved-ring loop (the dynamic draw exists in the two-stage loop only)"""
    global SCHED_ACTIVE
    SCHED_ACTIVE = bool(on)


def _addr(t):
    return t.data_ptr() if t is not None else None


def make_problem(A, B, M, N, K, C=None, C32=None, bias=None, addend=None, aux=None, out2=None, epi=0, alpha=1.0, colsum=None,
                 drop=NO_DROP, a_off=0, b_off=0):
    """a_off / b_off: element offsets into A / B (a K slice of a split-K problem keeps the parent's leading dimensions)"""
    _chk(A, BF16, "A"); _chk(B, BF16, "B")
    if drop[1]:
        epi |= L.EPI_DROP
    return L.GemmProblem(A.data_ptr() + 2 * a_off, B.data_ptr() + 2 * b_off, _addr(C), _addr(C32), _addr(bias), _addr(addend), _addr(aux),
                         _addr(out2), _addr(colsum),
                         M, N, K, A.shape[1], B.shape[1], C.shape[1] if C is not None else 0,
                         C32.shape[1] if C32 is not None else 0, addend.shape[1] if addend is not None else 0,
                         aux.shape[1] if aux is not None else 0, out2.shape[1] if out2 is not None else 0, epi, alpha,
                         drop[0], drop[1])


def gemm_variant(v=None):
    """main loop of the 256-row GEMM launches (include/kbner.h: kbner_gemm_set_variant); -> the previous value"""
    prev = int(L.load().kbner_gemm_get_variant())
    if v is not None:
        L.call("kbner_gemm_set_variant", int(v))
    return prev


def gemm_grouped(layout, problems):
    """problems: list (<= 16) of L.GemmProblem (make_problem) of one layout; 256x256x64 8-wave kernel.
    -> True when the launch used the dynamic tile draw (always 256-row tiles), False for the static kernel."""
    n = len(problems)
    arr = (L.GemmProblem * n)(*problems)
    hook = GEMM_HOOK
    if hook is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    global _sched_pos
    ring = SCHED_RING
    if ring is not None and SCHED_ACTIVE and _sched_pos < ring.shape[0]:
        slot = c_void_p(ring.data_ptr() + 32 * _sched_pos)
        _sched_pos += 1
        L.call("kbner_gemm_bf16_grouped_dyn", layout, n, ctypes.cast(arr, ctypes.c_void_p), slot, stream_ptr())
        dyn = True
    else:
        L.call("kbner_gemm_bf16_grouped", layout, n, ctypes.cast(arr, ctypes.c_void_p), stream_ptr())
        dyn = False
    if hook is not None:
        ev1.record()
        hook.append((ev0, ev1, sum(2.0 * p.M * p.N * p.K for p in problems), layout,
                     (len(problems), problems[0].M, problems[0].N, problems[0].K, problems[0].epi)))
    return dyn


def gemm(layout, A, B, M, N, K, C=None, C32=None, bias=None, addend=None, aux=None, out2=None, epi=0, splitk=1, alpha=1.0,
         lda=None, ldb=None, colsum=None, drop=NO_DROP, occupancy=False):
    """C[M,N] (bf16) or C32[M,N] += (fp32).  A/B are 2-D bf16 tensors in their memory layouts.
    Shapes divisible by 256 go to the 256^2 8-wave kernel, others to the 128^2 kernel.
    -> the tile height (rows) of the kernel that ran: 256 / 128 for the big kernel (EPI_COLSUM_WS consumers fold 2 * M / rows
    workspace lines), 128 for the small one."""
    _chk(A, BF16, "A"); _chk(B, BF16, "B")
    if uses_256(M, N, occupancy) and splitk == 1 and lda is None and ldb is None:
        dyn = gemm_grouped(layout, [make_problem(A, B, M, N, K, C, C32, bias, addend, aux, out2, epi, alpha, colsum, drop)])
        return 256 if dyn else gemm_tile_rows(layout, M, N)
    if drop[1]:
        epi |= L.EPI_DROP
    if colsum is not None:
        raise L.KbnerError("EPI_COLSUM needs the 256x256 kernel (M, N % 256 == 0)")
    lda = A.shape[1] if lda is None else lda
    ldb = B.shape[1] if ldb is None else ldb
    hook = GEMM_HOOK
    if hook is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    L.call("kbner_gemm_bf16", layout, ptr(A), lda, ptr(B), ldb, M, N, K,
           ptr(C), C.shape[1] if C is not None else 0,
           ptr(C32), C32.shape[1] if C32 is not None else 0,
           ptr(bias), ptr(addend), addend.shape[1] if addend is not None else 0,
           ptr(aux), aux.shape[1] if aux is not None else 0,
           ptr(out2), out2.shape[1] if out2 is not None else 0,
           epi, splitk, alpha, drop[0], drop[1], stream_ptr())
    if hook is not None:
        ev1.record()
        hook.append((ev0, ev1, 2.0 * M * N * K, layout, (1, M, N, K, epi)))
    return 128


def gemm_splitk(layout, A, B, M, N, K, splits, ws, C, bias=None, addend=None, drop=NO_DROP, finish=True):
    """C = bf16(dropout(A.B + bias) + addend) with K cut into `splits` problems of one grouped launch (fp32 slabs in ws
    f32[splits, M, N]) + the finish pass.  For NT / NN layouts whose output has too few tiles to fill the chip.
    finish=False: the slabs only -- the consumer folds them itself (ln_fwd_slabs / ln_bwd(dy_slabs=...))."""
    if layout not in (L.GEMM_NT, L.GEMM_NN) or K % (64 * splits) or M % 256 or N % 256:
        raise L.KbnerError("gemm_splitk: unsupported shape / layout")
    _chk(ws, F32, "ws")
    Ks = K // splits
    probs = []
    for s in range(splits):
        b_off = s * Ks if layout == L.GEMM_NT else s * Ks * B.shape[1]   # NT: B[N,K] rows; NN: Bmem[K,N] rows
        probs.append(make_problem(A, B, M, N, Ks, C32=ws[s], epi=L.EPI_STORE32, a_off=s * Ks, b_off=b_off))
    gemm_grouped(layout, probs)
    if not finish:
        return
    L.call("kbner_splitk_finish", ptr(ws), splits, ptr(bias), ptr(addend), addend.shape[1] if addend is not None else 0, ptr(C),
           C.shape[1], M, N, drop[0], drop[1], stream_ptr())


# ---------------------------------------------------------------- attention
def attn_fwd(qkv, maskbias, ctx, lse, B, S, H, A, drop=NO_DROP, ctx_lo=None):
    """ctx_lo uint8 [B*S*H] (optional): receives the residual O - bf16(O) for attn_bwd's D (include/kbner.h)"""
    L.call("kbner_attn_fwd", ptr(qkv), ptr(maskbias), ptr(ctx), ptr(ctx_lo), ptr(lse), B, S, H, A, drop[0], drop[1], stream_ptr())


def attn_bwd(qkv, ctx, dctx, maskbias, lse, dws, dqkv, B, S, H, A, drop=NO_DROP, dbias=None, ctx_lo=None):
    """dbias f32[3H] (optional): accumulates the column sums of dqkv (= d qkv.bias) inside the kernels"""
    L.call("kbner_attn_bwd", ptr(qkv), ptr(ctx), ptr(ctx_lo), ptr(dctx), ptr(maskbias), ptr(lse), ptr(dws), ptr(dqkv), B, S, H, A,
           drop[0], drop[1], ptr(dbias), stream_ptr())


# ---------------------------------------------------------------- LSTM (inference)
def lstm_step(gx, gxi, whh, h_in, h_out, c, out, outi, out_dir_stride):
    """one time step for ndir directions: gx bf16 [rows, ndir*4*Hp]; gxi / outi i32 [ndir, B]; whh bf16 [ndir, 4Hp, Hp];
    h_in / h_out bf16 [ndir, B, Hp]; c f32 [ndir, B, Hp]; out bf16 [rows_out, ldo]"""
    _chk(gx, BF16, "gx"); _chk(whh, BF16, "whh"); _chk(h_in, BF16, "h_in"); _chk(h_out, BF16, "h_out"); _chk(c, F32, "c")
    _chk(out, BF16, "out"); _chk(gxi, I32, "gxi"); _chk(outi, I32, "outi")
    ndir, B, Hp = h_in.shape
    L.call("kbner_lstm_step", ptr(gx), gx.shape[-1], ptr(gxi), ptr(whh), ptr(h_in), ptr(h_out), ptr(c), ptr(out), out.shape[-1],
           out_dir_stride, ptr(outi), B, Hp, ndir, stream_ptr())


# ---------------------------------------------------------------- optimiser
def grad_sqnorm(g, ws, out, accumulate=False):
    _chk(g, F32, "g")
    with _hbm_timed("grad_sqnorm", 4 * g.numel()):
        L.call("kbner_grad_sqnorm", ptr(g), g.numel(), ptr(ws), ptr(out), 1 if accumulate else 0, stream_ptr())


def adamw(p, g, m, v, shadow, n_shadow, step_size, lr_wd, b1, b2, eps, gnorm_sq, max_norm, grad_scale, zero_grad=True):
    # 28 B / parameter (read g, p, m, v; write p, m, v) -- the bf16 shadow write and the gradient zeroing the
    # kernel also does are not counted
    with _hbm_timed("adamw_kernel", 28 * p.numel()):
        L.call("kbner_adamw_hf", ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), p.numel(), n_shadow, step_size, lr_wd, b1, b2, eps,
               ptr(gnorm_sq), max_norm, grad_scale, 1 if zero_grad else 0, stream_ptr())


U8 = torch.uint8


def mark_rows(ids, flags):
    """flags u8[rows]: flags[ids[i]] = 1 (negative ids ignored)"""
    _chk(ids, I32, "ids"); _chk(flags, U8, "flags")
    L.call("kbner_mark_rows", ptr(ids), ids.numel(), ptr(flags), flags.numel(), stream_ptr())


def grad_sqnorm_rows(g2d, flags, ws, out, accumulate=True):
    _chk(g2d, F32, "g"); _chk(flags, U8, "flags")
    # (algorithmic bytes of the row kernels: every row counted -- bench.py times them with every row live)
    with _hbm_timed("grad_sqnorm_rows", 4 * g2d.numel()):
        L.call("kbner_grad_sqnorm_rows", ptr(g2d), ptr(flags), g2d.shape[0], g2d.shape[1], ptr(ws), ptr(out), 1 if accumulate else 0,
               stream_ptr())


def adamw_rows(p, g, m, v, flags, step_size, b1, b2, eps, gnorm_sq, max_norm, grad_scale, zero_grad=True):
    """HF AdamW (weight decay 0) on the flagged rows of an embedding table [rows, width]"""
    _chk(p, F32, "p"); _chk(flags, U8, "flags")
    with _hbm_timed("adamw_rows", 28 * p.numel()):
        L.call("kbner_adamw_hf_rows", ptr(p), ptr(g), ptr(m), ptr(v), ptr(flags), p.shape[0], p.shape[1], step_size, b1, b2, eps,
               ptr(gnorm_sq), max_norm, grad_scale, 1 if zero_grad else 0, stream_ptr())


def adamw_rows_lazy(p, g, m, v, flags, row_t, clock, hist, t, step_size, b1, b2, eps, gnorm_sq, max_norm, grad_scale):
    """step t on the TOUCHED rows only (each first brought up to t - 1), then hist[t] <- step_size, clock <- t (include/kbner.h)"""
    _chk(p, F32, "p"); _chk(flags, U8, "flags"); _chk(row_t, I32, "row_t"); _chk(clock, I32, "clock"); _chk(hist, F32, "hist")
    L.call("kbner_adamw_hf_rows_lazy", ptr(p), ptr(g), ptr(m), ptr(v), ptr(flags), ptr(row_t), ptr(clock), ptr(hist), hist.numel(), int(t),
           p.shape[0], p.shape[1], step_size, b1, b2, eps, ptr(gnorm_sq), max_norm, grad_scale, stream_ptr())


def adamw_rows_catchup(ids, p, m, v, flags, row_t, clock, hist, b1, b2, eps):
    """rows `ids` (i32, device; None: every row) of a lazily updated table brought to step clock[0]"""
    _chk(p, F32, "p"); _chk(flags, U8, "flags"); _chk(row_t, I32, "row_t"); _chk(clock, I32, "clock"); _chk(hist, F32, "hist")
    if ids is not None:
        _chk(ids, I32, "ids")
    L.call("kbner_adamw_rows_catchup", ptr(ids), ids.numel() if ids is not None else 0, ptr(p), ptr(m), ptr(v), ptr(flags), ptr(row_t),
           ptr(clock), ptr(hist), hist.numel(), p.shape[0], p.shape[1], b1, b2, eps, stream_ptr())


def f32_to_bf16(x, y):
    L.call("kbner_f32_to_bf16", ptr(x), ptr(y), x.numel(), stream_ptr())


def bf16_to_f32(x, y):
    L.call("kbner_bf16_to_f32", ptr(x), ptr(y), x.numel(), stream_ptr())


def wdiff_sum(a, b, w, out):
    L.call("kbner_wdiff_sum", ptr(a), ptr(b), ptr(w), a.numel(), ptr(out), stream_ptr())
