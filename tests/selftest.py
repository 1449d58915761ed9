"""TEST INFRASTRUCTURE: GPU self-checks of the HIP path against the oracle (oracle/ is imported HERE as the checker; this
module is used by tests/, __graft_entry__.smoke() and tools/gpu_diag.py -- it is not part of the kb-ner_amd product package)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "kb-ner_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from kbner import batch as kb  # noqa: E402
from kbner import engine, ops  # noqa: E402
from kbner.lib import EPI_ADD, EPI_ATOMIC32, EPI_BIAS, EPI_DGELU, EPI_GELU, EPI_GELU_FWD, EPI_RMW32, GEMM_NN, GEMM_NT, GEMM_TN  # noqa: E402

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
DEV = "cuda"


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


# ------------------------------------------------------------------ probes
def probe_tr():
    from kbner import lib as L
    inp = torch.arange(2048, dtype=torch.int16, device=DEV)
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    L.call("kbner_probe_tr", L.ptr(inp), L.ptr(out), L.stream_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(64, 4)
    exp = np.zeros((64, 4), np.int16)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for j in range(4):
            exp[lane, j] = (4 * g + j) * 64 + i
    return got, exp


def probe_mfma():
    from kbner import lib as L
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.randn(16, 32, generator=g).to(BF16)
    b = torch.randn(16, 32, generator=g).to(BF16)
    c = torch.zeros(16, 16, dtype=F32, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    L.call("kbner_probe_mfma", L.ptr(ad), L.ptr(bd), L.ptr(c), L.stream_ptr())
    torch.cuda.synchronize()
    ref = a.float() @ b.float().t()
    return c.cpu(), ref


# ------------------------------------------------------------------ GEMM
def check_gemm(layout, M, N, K, epi=0, splitk=1, seed=0, drop_p=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF16)
    Bm = (torch.randn(N, K, generator=g) * 0.5).to(BF16)
    ref = A.double() @ Bm.double().t()
    bias = torch.randn(N, generator=g)
    add = (torch.randn(M, N, generator=g)).to(BF16)
    aux = (torch.randn(M, N, generator=g)).to(BF16)
    Ad = A.to(DEV) if layout != GEMM_TN else A.t().contiguous().to(DEV)
    Bd = Bm.to(DEV) if layout == GEMM_NT else Bm.t().contiguous().to(DEV)
    kw = {}
    if epi & EPI_BIAS:
        ref = ref + bias.double()[None, :]
        kw["bias"] = bias.to(DEV)
    if drop_p:  # dropout(acc + bias) BEFORE the residual add; reference multiplier from kbner_dropout_mask
        kw["drop"] = (1234567 + seed, ops.drop_thresh(drop_p))
        ref = ref * ops.dropout_mask(1, M, N, *kw["drop"])[0].cpu().double()
    if epi & EPI_ADD:
        ref = ref + add.double()
        kw["addend"] = add.to(DEV)
    if epi & EPI_DGELU:  # aux is the saved derivative: plain multiply
        ref = ref * aux.double()
        kw["aux"] = aux.to(DEV)
    out2 = None
    if epi & EPI_GELU:
        out2 = torch.zeros(M, N, dtype=BF16, device=DEV)
        kw["out2"] = out2
        pre = ref            # gelu / gelu' of the fp32 pre-activation (rounds 1-3: of its bf16 rounding)
        cdf = 0.5 * (1 + torch.erf(pre / np.sqrt(2.0)))
        ref_pre = cdf + pre * torch.exp(-0.5 * pre * pre) / np.sqrt(2 * np.pi)   # out2 = gelu'(pre)
        ref = torch.nn.functional.gelu(pre)
    if epi & EPI_GELU_FWD:  # the activation alone, at the fp32 pre-activation like EPI_GELU
        ref = torch.nn.functional.gelu(ref)
    if epi & (EPI_ATOMIC32 | EPI_RMW32):
        C32 = torch.full((M, N), 1.0, dtype=F32, device=DEV)
        ops.gemm(layout, Ad, Bd, M, N, K, C32=C32, epi=epi, splitk=splitk, **kw)
        torch.cuda.synchronize()
        return rel_l2(C32.cpu() - 1.0, ref)
    C = torch.zeros(M, N, dtype=BF16, device=DEV)
    ops.gemm(layout, Ad, Bd, M, N, K, C=C, epi=epi, splitk=splitk, **kw)
    torch.cuda.synchronize()
    err = rel_l2(C.cpu().float(), ref)
    if out2 is not None:
        err = max(err, rel_l2(out2.cpu().float(), ref_pre))
    return err


def check_gemm_splitk(layout, M, N, K, splits, seed=0, drop_p=0.0):
    """split-K (fp32 slabs + finish pass) vs fp64: C = dropout(A.B^T + bias) + addend"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF16)
    Bm = (torch.randn(N, K, generator=g) * 0.5).to(BF16)
    bias = torch.randn(N, generator=g)
    add = torch.randn(M, N, generator=g).to(BF16)
    ref = A.double() @ Bm.double().t() + bias.double()[None, :]
    drop = ops.NO_DROP
    if drop_p:
        drop = (777 + seed, ops.drop_thresh(drop_p))
        ref = ref * ops.dropout_mask(1, M, N, *drop)[0].cpu().double()
    ref = ref + add.double()
    Ad = A.to(DEV)
    Bd = Bm.to(DEV) if layout == GEMM_NT else Bm.t().contiguous().to(DEV)
    ws = torch.empty(4, M, N, dtype=F32, device=DEV)
    C = torch.zeros(M, N, dtype=BF16, device=DEV)
    ops.gemm_splitk(layout, Ad, Bd, M, N, K, splits, ws, C, bias=bias.to(DEV), addend=add.to(DEV), drop=drop)
    torch.cuda.synchronize()
    return rel_l2(C.cpu().float(), ref)


def check_gemm_grouped(seed=0):
    """four TN (wgrad) problems of different shapes in one grouped launch, RMW32 epilogue."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = [(256, 512, 320), (512, 256, 320), (256, 256, 320), (768, 256, 320)]
    probs, refs, outs, keep = [], [], [], []
    for (N_, K_, M_) in shapes:
        dy = (torch.randn(M_, N_, generator=g) * 0.5).to(BF16)
        x = (torch.randn(M_, K_, generator=g) * 0.5).to(BF16)
        refs.append(dy.double().t() @ x.double())
        dyd, xd = dy.to(DEV), x.to(DEV)
        c = torch.full((N_, K_), 2.0, dtype=F32, device=DEV)
        keep += [dyd, xd]
        outs.append(c)
        probs.append(ops.make_problem(dyd, xd, N_, K_, M_, C32=c, epi=EPI_RMW32))
    ops.gemm_grouped(GEMM_TN, probs)
    torch.cuda.synchronize()
    return max(rel_l2(o.cpu() - 2.0, r) for o, r in zip(outs, refs))


# ------------------------------------------------------------------ attention
def attn_reference(qkv, maskbias, B, S, H, A, dctx=None, pmask=None):
    d = H // A
    x = qkv.detach().double().clone().requires_grad_(dctx is not None)
    q, k, v = x[:, :H], x[:, H:2 * H], x[:, 2 * H:]
    sp = lambda t: t.reshape(B, S, A, d).transpose(1, 2)  # noqa: E731
    sc = sp(q) @ sp(k).transpose(-1, -2) / np.sqrt(d) + maskbias.double()[:, None, None, :]
    pr = torch.softmax(sc, dim=-1)
    if pmask is not None:
        pr = pr * pmask.double()
    ctx = (pr @ sp(v)).transpose(1, 2).reshape(B * S, H)
    lse = torch.logsumexp(sc, dim=-1)
    if dctx is None:
        return ctx, lse, None
    ctx.backward(dctx.double())
    return ctx.detach(), lse.detach(), x.grad


def check_attention(B, S, A, seed=0, ragged=True, drop_p=0.0, residual=False, collapse=0.0, v_scale=1.0):
    """residual: the forward also stores O - bf16(O) and the backward takes D from the pair (include/kbner.h kbner_attn_bwd).
    v_scale: multiplies the V rows (activation outliers: the residual byte's e5m2 range, csrc/common.h pack2bf_res8).
    collapse > 0: the K and V rows of a head are one common row + collapse * noise (what deep layers of a freshly initialised
    encoder look like): dS = P (dP - D) cancels and the rounding of a bf16 O is amplified into dQ."""
    H = A * 64
    g = torch.Generator(device="cpu").manual_seed(seed)
    qkv = torch.randn(B * S, 3 * H, generator=g)
    if collapse:
        common = torch.randn(1, 2 * H, generator=g)
        qkv[:, H:] = common + collapse * qkv[:, H:]
    if v_scale != 1.0:
        qkv[:, 2 * H:] *= v_scale
    qkv = qkv.to(BF16)
    dctx = (torch.randn(B * S, H, generator=g)).to(BF16)
    am = torch.ones(B, S)
    if ragged:
        for b in range(B):
            am[b, S - 7 * b - (5 if b else 0):] = 0 if b else 1
    mb = ((1 - am) * -10000.0).float()
    drop = (424242 + seed, ops.drop_thresh(drop_p)) if drop_p else ops.NO_DROP
    pmask = ops.dropout_mask(B * A, S, S, *drop).view(B, A, S, S).cpu() if drop_p else None
    ctx_ref, lse_ref, dqkv_ref = attn_reference(qkv.float(), mb, B, S, H, A, dctx.float(), pmask)
    qd, dd, mbd = qkv.to(DEV), dctx.to(DEV), mb.to(DEV)
    ctx = torch.zeros(B * S, H, dtype=BF16, device=DEV)
    lse = torch.zeros(B, A, S, dtype=F32, device=DEV)
    ctx_lo = torch.zeros(B * S * H, dtype=torch.uint8, device=DEV) if residual else None
    ops.attn_fwd(qd, mbd, ctx, lse, B, S, H, A, drop=drop, ctx_lo=ctx_lo)
    dws = torch.zeros(B, A, S, dtype=F32, device=DEV)
    dqkv = torch.zeros(B * S, 3 * H, dtype=BF16, device=DEV)
    dbias = torch.zeros(3 * H, dtype=F32, device=DEV)
    ops.attn_bwd(qd, ctx, dd, mbd, lse, dws, dqkv, B, S, H, A, drop=drop, dbias=dbias, ctx_lo=ctx_lo)
    torch.cuda.synchronize()
    dq = dqkv.cpu().float()
    bias_ref = dqkv_ref.sum(0)
    return {
        "finite": bool(torch.isfinite(dq).all() and torch.isfinite(ctx.float()).all()),
        # fused d qkv.bias = column sums of dqkv; the K third is ~0 by construction (softmax shift invariance), so it is
        # compared on the absolute scale of the Q / V thirds
        "dbias": float((dbias.cpu().double() - bias_ref).abs().max() / (bias_ref.abs().max() + 1e-30)),
        "ctx": rel_l2(ctx.cpu().float(), ctx_ref),
        "lse": float((lse.cpu().double() - lse_ref).abs().max()),
        "dq": rel_l2(dq[:, :H], dqkv_ref[:, :H]),
        "dk": rel_l2(dq[:, H:2 * H], dqkv_ref[:, H:2 * H]),
        "dv": rel_l2(dq[:, 2 * H:], dqkv_ref[:, 2 * H:]),
    }


# ------------------------------------------------------------------ LayerNorm
def check_layernorm(M, H, seed=0, drop_p=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    h = (torch.randn(M, H, generator=g) * 2 + 0.3).to(BF16)
    dy = torch.randn(M, H, generator=g).to(BF16)
    gamma = torch.randn(H, generator=g) * 0.2 + 1.0
    beta = torch.randn(H, generator=g) * 0.1
    hr = h.double().clone().requires_grad_(True)
    gr = gamma.double().clone().requires_grad_(True)
    br = beta.double().clone().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(hr, (H,), gr, br, 1e-5)
    y_ref.backward(dy.double())
    hd, dyd = h.to(DEV), dy.to(DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    y = torch.zeros(M, H, dtype=BF16, device=DEV)
    mean = torch.zeros(M, dtype=F32, device=DEV)
    rstd = torch.zeros(M, dtype=F32, device=DEV)
    ops.ln_fwd(hd, gd, bd, 1e-5, y, mean, rstd)
    dh = torch.zeros(M, H, dtype=BF16, device=DEV)
    dg = torch.zeros(H, dtype=F32, device=DEV)
    db = torch.zeros(H, dtype=F32, device=DEV)
    dbias = torch.zeros(H, dtype=F32, device=DEV)
    drop = (99 + seed, ops.drop_thresh(drop_p)) if drop_p else ops.NO_DROP
    dhm = torch.zeros(M, H, dtype=BF16, device=DEV) if drop_p else None
    ops.ln_bwd(dyd, hd, mean, rstd, gd, dh, dg, db, dbias, dhm=dhm, drop=drop)
    torch.cuda.synchronize()
    res = {
        "y": rel_l2(y.cpu().float(), y_ref.detach()),
        "dh": rel_l2(dh.cpu().float(), hr.grad),
        "dgamma": rel_l2(dg.cpu(), gr.grad),
        "dbeta": rel_l2(db.cpu(), br.grad),
    }
    if drop_p:  # dhm = mask * dh (the dropped branch's dY); dbias sums the masked rows
        mref = hr.grad * ops.dropout_mask(1, M, H, *drop)[0].cpu().double()
        res["dhm"] = rel_l2(dhm.cpu().float(), mref)
        res["dbias"] = rel_l2(dbias.cpu(), mref.sum(0))
    else:
        res["dbias"] = rel_l2(dbias.cpu(), hr.grad.sum(0))
    return res


# ------------------------------------------------------------------ CRF
def check_crf(B, n, T=29, start=27, stop=28, seed=0):
    from oracle import crf as ocrf
    rng = np.random.default_rng(seed)
    trans = ocrf.init_transitions(T, start, stop, rng)
    feats = (rng.standard_normal((B, n, T)) * 2).astype(np.float32)
    lens = rng.integers(0 if B > 2 else 1, n + 1, size=B).astype(np.int32)
    lens[0] = n
    valid = [t for t in range(T) if t not in (start, stop)]
    tags = rng.choice(valid, size=(B, n)).astype(np.int32)
    fd, td = torch.from_numpy(feats).to(DEV), torch.from_numpy(trans).to(DEV)
    ld, tgd = torch.from_numpy(lens).to(DEV), torch.from_numpy(tags).to(DEV)
    vt, vc, popped = ops.crf_viterbi(fd, td, ld, start, stop, want_popped=True)
    logz, gold, alpha = ops.crf_nll_fwd(fd, td, tgd, ld, start, stop)
    dl = torch.full((B,), 1.0 / B, dtype=F32, device=DEV)
    dtr = torch.zeros(T, T, dtype=F32, device=DEV)
    demit = ops.crf_nll_bwd(fd, td, tgd, ld, alpha, logz, dl, start, stop, dtr)
    torch.cuda.synchronize()
    rt, rc = ocrf.viterbi_batch(feats, lens, trans, start, stop)
    rz = ocrf.forward_alg(feats, lens, trans, start, stop)
    rg = ocrf.score_sentence(feats, tags, lens, trans, start, stop)
    dfe, dtrr = ocrf.crf_nll_grads(feats, tags, lens, trans, start, stop, dloss=np.full(B, 1.0 / B))
    for b in range(B):
        dfe[b, lens[b]:] = 0
    return {
        "tags_equal": bool(np.array_equal(vt.cpu().numpy(), rt)),
        "popped_ok": bool(np.all(popped.cpu().numpy() == start)),
        "conf": float(np.abs(vc.cpu().numpy() - rc).max()),
        "logz": float(np.abs(logz.cpu().numpy() - rz).max() / (np.abs(rz).max() + 1e-9)),
        "gold": float(np.abs(gold.cpu().numpy() - rg).max() / (np.abs(rg).max() + 1e-9)),
        "demit": float(np.abs(demit.cpu().numpy() - dfe).max()),
        "dtrans": float(np.abs(dtr.cpu().numpy() - dtrr).max()),
    }


# ------------------------------------------------------------------ end-to-end tiny tagger step
def tiny_setup(B=2, S=64, L=2, H=128, A=2, F_=256, V=512, T=29, seed=5, std=0.08):
    cfg = engine.EncoderConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=A,
                               intermediate_size=F_, max_position_embeddings=S + 2)
    start, stop, x_idx = 27, 28, 9
    tg = engine.Tagger(cfg, T, start, stop, device=DEV)
    tg.init_random(seed=seed, std=std)
    b = kb.synthetic_batch(B, S, vocab=V, T=T, x_idx=x_idx, start=start, stop=stop, n_real=6, seed=seed)
    # make it ragged: second sentence shorter
    if B > 1:
        ids, am = b["input_ids"].copy(), b["attention_mask"].copy()
        cut = S - 9
        ids[1, cut - 1] = 2
        ids[1, cut:] = 0
        am[1, cut:] = 0
        fi = b["first_idx"].copy()
        fi[1][fi[1] >= cut - 1] = -1
        lengths = np.asarray([int((fi[r] >= 0).sum()) for r in range(B)])
        tags = b["tags"].copy()
        b = kb.assemble(ids, am, fi, tags, lengths, x_idx)
    return cfg, tg, b, (start, stop, x_idx)


def oracle_params(tg, round_gemm=True):
    """fp32 copies of the tagger's parameters under HF names (+ head), GEMM weights rounded through bf16
    so the comparison isolates kernel error from the (intended) bf16 weight quantisation."""
    sd = {k: v.cpu() for k, v in tg.hf_state_dict().items()}
    if round_gemm:
        for k in list(sd):
            if k.endswith("dense.weight") or k.endswith("query.weight") or k.endswith("key.weight") or k.endswith("value.weight"):
                sd[k] = sd[k].to(BF16).float()
    for k in ("linear.weight", "linear.bias", "transitions"):
        sd[k] = tg.arena.param(k).detach().cpu().clone()
    return sd


def dropout_masks_of(tg, B, S):
    """The multipliers the last training-mode forward applied, rebuilt from its saved seeds with kbner_dropout_mask."""
    cfg = tg.cfg
    H, A = cfg.hidden_size, cfg.num_attention_heads
    M = B * S
    d_emb, d_layers = tg._enc_saved[5], tg._enc_saved[6]
    hid = lambda d: ops.dropout_mask(1, tg.acts(B, S).Mp, H, d[0], d[1])[0, :M].view(B, S, H).cpu()  # noqa: E731
    masks = {}
    if d_emb[1]:
        masks["emb"] = hid(d_emb)
    for i, (d_att, d_o, d_f) in enumerate(d_layers):
        if d_att[1]:
            masks[("attn", i)] = ops.dropout_mask(B * A, S, S, d_att[0], d_att[1]).view(B, A, S, S).cpu()
        if d_o[1]:
            masks[("o", i)] = hid(d_o)
        if d_f[1]:
            masks[("ffn", i)] = hid(d_f)
    return masks


def check_dropout_mask(p=0.1, Z=3, M=512, N=512, seed=12345):
    """statistics of the counter-based mask: keep rate, row / column keep-rate spread, replay determinism, seed sensitivity"""
    th = ops.drop_thresh(p)
    m = ops.dropout_mask(Z, M, N, seed, th)
    m2 = ops.dropout_mask(Z, M, N, seed, th)
    m3 = ops.dropout_mask(Z, M, N, seed + 1, th)
    torch.cuda.synchronize()
    keep = (m > 0).float()
    vals = torch.unique(m).cpu().tolist()
    k3 = (m3 > 0).float()
    # correlation between the masks of two seeds and between adjacent rows / columns of one mask
    def corr(a, b):
        a, b = a.flatten() - a.mean(), b.flatten() - b.mean()
        return float((a @ b) / (a.norm() * b.norm() + 1e-30))
    return {"keep_rate": float(keep.mean()), "values": vals, "scale": 1.0 / (1.0 - p),
            "row_rate_min": float(keep.mean(2).min()), "row_rate_max": float(keep.mean(2).max()),
            "col_rate_min": float(keep.mean(1).min()), "col_rate_max": float(keep.mean(1).max()),
            "replay_equal": bool(torch.equal(m, m2)), "seed_corr": corr(keep, k3),
            "adj_row_corr": corr(keep[:, 1:], keep[:, :-1]), "adj_col_corr": corr(keep[:, :, 1:], keep[:, :, :-1])}


def check_step(dropout=False, H=128, A=2, F_=256, L=2, S=64, V=512, std=0.08, bf16_oracle=False):
    """One micro-batch fwd+bwd on the HIP path vs the oracle's autograd (fp32 CPU).  dropout=True: training mode with
    p=0.1 at every encoder site + WordDropout 0.1; the oracle is fed the very masks the kernels generated.
    bf16_oracle=True (attribution of the gradient error): a SECOND oracle pass rounds to bf16 at the
    points where the HIP path stores bf16 (oracle/encoder.py bf16_points); the result then also carries, per gradient tensor,
    HIP vs that rounded oracle and rounded oracle vs fp32 oracle -- if the kernels are right, the first is small and the second
    reproduces the HIP-vs-fp32 distance."""
    from oracle import encoder as oenc
    from oracle import train_step as ots
    cfg, tg, b, (start, stop, x_idx) = tiny_setup(H=H, A=A, F_=F_, L=L, S=S, V=V, std=std)
    bd = kb.to_device(b, DEV)
    masks, word_keep = None, None
    if dropout:
        tg.train(True)
        tg.word_dropout = 0.1
        tg.seed_dropout(7)
    loss = tg.forward_loss(bd, loss_scale=1.0, backward=True)
    torch.cuda.synchronize()
    if dropout:
        masks = dropout_masks_of(tg, b["B"], b["S"])
        word_keep = torch.from_numpy(~tg._last_word_dropped)
        tg.train(False)
    ocfg = oenc.EncoderConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                              num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                              max_position_embeddings=cfg.max_position_embeddings)
    params = {k: v.clone().requires_grad_(True) for k, v in oracle_params(tg).items()}
    ob = dict(input_ids=torch.from_numpy(b["input_ids"]), attention_mask=torch.from_numpy(b["attention_mask"]),
              first_idx=torch.from_numpy(b["first_idx"]), tags=torch.from_numpy(b["tags"].astype(np.int64)),
              lengths=torch.from_numpy(b["lengths"].astype(np.int64)))
    oloss, oem = ots.tagger_forward_loss(params, ocfg, ob, start, stop, x_idx, masks=masks, word_keep=word_keep)
    oloss.backward()
    oloss_f = float(oloss.detach())
    res = {"loss_hip": float(loss), "loss_oracle": oloss_f, "loss_rel": abs(float(loss) - oloss_f) / abs(oloss_f)}
    if dropout:
        res["n_sites"] = len(masks)
        res["word_dropped"] = int((~word_keep).sum())
        with torch.no_grad():
            oem = ots.tagger_forward_loss(params, ocfg, ob, start, stop, x_idx)[1]
    # hidden / emissions parity (forward only, all tokens; evaluation mode)
    em = tg.forward_features(bd)
    torch.cuda.synchronize()
    n = b["first_idx"].shape[1]
    valid = torch.from_numpy(b["first_idx"] >= 0)
    res["emissions_rel"] = rel_l2(em.cpu()[valid], oem.detach()[valid])
    # gradients.  attention key biases are excluded: softmax is invariant to a per-query constant, so
    # d loss / d key.bias is exactly 0 in exact arithmetic and both sides hold only roundoff noise.
    nm = engine.hf_name_map(cfg)
    worst, worst_name, coss = 0.0, "", 1.0
    table = []
    gscale = max(float(v.grad.abs().max()) for v in params.values() if v.grad is not None)
    for hf, (mine, sl) in nm.items():
        gh = tg.arena.grad(mine)
        gh = (gh[sl[0]:sl[1]] if sl is not None else gh).cpu()
        go = params[hf].grad
        if go is None or hf.endswith("key.bias") or float(go.abs().max()) < 1e-7 * gscale:
            continue
        e = rel_l2(gh, go)
        c = cosine(gh, go)
        table.append((e, c, hf))
        coss = min(coss, c)
        if e > worst:
            worst, worst_name = e, hf
    res["grad_table_top"] = ["%s rel=%.3g cos=%.5f" % (n_, e_, c_) for e_, c_, n_ in sorted(table, reverse=True)[:6]]
    res["grad_worst_rel"], res["grad_worst_name"], res["grad_min_cos"] = worst, worst_name, coss
    for k in ("linear.weight", "linear.bias", "transitions"):
        res["grad_" + k] = rel_l2(tg.arena.grad(k).cpu(), params[k].grad)
    # bf16_oracle: True = one pass with bf16_points=True; a tuple of modes (True | "flash" | "flash_split") = one pass each
    for mode in ((bf16_oracle if isinstance(bf16_oracle, (tuple, list)) else (bf16_oracle,)) if bf16_oracle else ()):
        tag = "bf16_oracle" if mode is True else "%s_oracle" % mode
        p16 = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        l16, _ = ots.tagger_forward_loss(p16, ocfg, ob, start, stop, x_idx, masks=masks, word_keep=word_keep, bf16_points=mode)
        l16.backward()
        res["loss_rel_vs_" + tag] = abs(float(loss) - float(l16.detach())) / abs(float(l16.detach()))
        t16, w_hip, w_rnd, per_layer = [], (0.0, "", 1.0), (0.0, "", 1.0), {}
        for hf, (mine, sl) in nm.items():
            go, g16 = params[hf].grad, p16[hf].grad
            if go is None or g16 is None or hf.endswith("key.bias") or float(go.abs().max()) < 1e-7 * gscale:
                continue
            gh = tg.arena.grad(mine)
            gh = (gh[sl[0]:sl[1]] if sl is not None else gh).cpu()
            e_h16, c_h16 = rel_l2(gh, g16), cosine(gh, g16)          # HIP vs the rounding oracle
            e_16o, c_16o = rel_l2(g16, go), cosine(g16, go)          # what the rounding alone does to the fp32 gradient
            t16.append((e_h16, c_h16, e_16o, c_16o, hf))
            if e_h16 > w_hip[0]:
                w_hip = (e_h16, hf, c_h16)
            if e_16o > w_rnd[0]:
                w_rnd = (e_16o, hf, c_16o)
            if hf.endswith("attention.self.query.weight"):
                per_layer[int(hf.split(".")[2])] = (round(rel_l2(gh, go), 4), round(e_h16, 4), round(e_16o, 4))
        res["grad_worst_rel_vs_" + tag], res["grad_worst_name_vs_" + tag], res["grad_cos_vs_" + tag] = w_hip
        res["grad_min_cos_vs_" + tag] = min(t[1] for t in t16)
        res[tag + "_vs_fp32_worst_rel"], res[tag + "_vs_fp32_worst_name"], _ = w_rnd
        # per layer, query.weight: (HIP vs fp32 oracle, HIP vs this oracle, this oracle vs fp32 oracle) relative L2
        res["query_weight_rel_by_layer_" + tag] = [per_layer[k] for k in sorted(per_layer)]
    # Viterbi on the HIP emissions == oracle Viterbi on the SAME emissions (bit-exact indices)
    from oracle import crf as ocrf
    lens = torch.from_numpy(b["lengths"]).to(DEV)
    vt, vc = tg.viterbi(em, lens)
    torch.cuda.synchronize()
    rt, rc = ocrf.viterbi_batch(em.cpu().numpy(), b["lengths"], tg.arena.param("transitions").cpu().numpy(), start, stop)
    res["viterbi_equal"] = bool(np.array_equal(vt.cpu().numpy(), rt))
    return res


def check_windows(W=64, stride=32, n_content=150, H=128, A=2, F_=256, V=512, T=29, seed=11):
    """A sentence longer than one window (flair/embeddings.py:3203-3227,3292-3299) next to a short one: the HIP path
    gathers each word token from (window row, position); the oracle STITCHES the window states with the reference's seam
    rule and walks the concatenation.  Both must agree on loss / emissions."""
    from oracle import encoder as oenc
    from oracle import train_step as ots
    rng = np.random.default_rng(seed)
    cfg = engine.EncoderConfig(vocab_size=V, hidden_size=H, num_hidden_layers=2, num_attention_heads=A, intermediate_size=F_,
                               max_position_embeddings=W + 2)
    start, stop, x_idx = 27, 28, 9
    tg = engine.Tagger(cfg, T, start, stop, device=DEV)
    tg.init_random(seed=seed, std=0.08)
    Wc = W - 2
    content = rng.integers(5, V, size=n_content)
    starts = [0]
    while starts[-1] + Wc < n_content:
        starts.append(starts[-1] + Wc - stride)
    rows = [[0] + list(content[lo:lo + Wc]) + [2] for lo in starts]
    short = [0] + list(rng.integers(5, V, size=20)) + [2]
    all_rows = [short] + rows
    S0 = max(len(r) for r in all_rows)
    ids = np.zeros((len(all_rows), S0), np.int64)
    am = np.zeros_like(ids)
    for i, r in enumerate(all_rows):
        ids[i, :len(r)], am[i, :len(r)] = r, 1
    # word tokens: every 1-3 sub-tokens; positions in CONTENT coordinates
    def words(n):
        f, g = [], 0
        while g < n:
            f.append(g)
            g += int(rng.choice([1, 2, 3], p=[0.6, 0.3, 0.1]))
        return f
    f_short, f_long = words(20), words(n_content)
    n = max(len(f_short), len(f_long))
    first = np.full((2, n), -1, np.int64)
    frow = np.zeros((2, n), np.int64)
    first[0, :len(f_short)] = np.asarray(f_short) + 1
    half = stride // 2
    for k, g in enumerate(f_long):
        for w, lo in enumerate(starts):
            if w == len(starts) - 1 or g < lo + Wc - half:
                frow[1, k], first[1, k] = 1 + w, g - lo + 1
                break
    lengths = np.asarray([len(f_short), len(f_long)])
    tags = rng.integers(1, 9, size=(2, n))
    b = kb.assemble(ids, am, first, tags, lengths, x_idx, first_row=frow)
    bd = kb.to_device(b, DEV)
    loss = tg.forward_loss(bd, backward=False)
    em = tg.forward_features(bd)
    torch.cuda.synchronize()
    # oracle: per-row encoder, reference stitching, first sub-token by walking the stitched sequence
    ocfg = oenc.EncoderConfig(vocab_size=V, hidden_size=H, num_hidden_layers=2, num_attention_heads=A, intermediate_size=F_,
                              max_position_embeddings=W + 2)
    params = oracle_params(tg)
    with torch.no_grad():
        hid = oenc.encoder_forward(params, ocfg, torch.from_numpy(b["input_ids"]), torch.from_numpy(b["attention_mask"]))
        stitched = oenc.stitch_windows([hid[1 + w, :len(rows[w])] for w in range(len(rows))], stride)
        assert stitched.shape[0] == n_content + 2
        pooled = torch.zeros(2, n, H)
        pooled[0, :len(f_short)] = hid[0, torch.as_tensor(f_short) + 1]
        pooled[1, :len(f_long)] = stitched[torch.as_tensor(f_long) + 1]
        oem = torch.nn.functional.linear(pooled, params["linear.weight"], params["linear.bias"])
        ob = dict(input_ids=torch.from_numpy(b["input_ids"]), attention_mask=torch.from_numpy(b["attention_mask"]),
                  first_idx=torch.from_numpy(first), first_row=torch.from_numpy(frow), tags=torch.from_numpy(tags.astype(np.int64)),
                  lengths=torch.from_numpy(lengths.astype(np.int64)))
        oloss, oem2 = ots.tagger_forward_loss(params, ocfg, ob, start, stop, x_idx)
    valid = torch.from_numpy(first >= 0)
    return {"windows": len(rows), "emissions_rel": rel_l2(em.cpu()[valid], oem[valid]),
            "oracle_gather_vs_stitch": rel_l2(oem2[valid], oem[valid]),
            "loss_rel": abs(float(loss) - float(oloss)) / abs(float(oloss))}


def check_train_steps(steps=3, accum=2, lr=2e-4, lr_rate=50.0, t_total=10):
    """Whole optimiser steps (finetune_trainer.py:876-1023: accumulate -> clip_grad_norm_(5.0) -> HF AdamW with the two lr
    groups -> linear decay) on the HIP engine vs the oracle trainer from identical parameters and micro-batches."""
    from oracle import encoder as oenc
    from oracle import train_step as ots
    cfg, tg, b0, (start, stop, x_idx) = tiny_setup()
    batches = [b0] + [tiny_setup(seed=5 + k)[2] for k in range(1, accum)]
    ocfg = oenc.EncoderConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                              num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                              max_position_embeddings=cfg.max_position_embeddings)
    p0 = oracle_params(tg, round_gemm=False)
    tr = ots.OracleTrainer(p0, ocfg, start, stop, x_idx, lr=lr, lr_rate=lr_rate, accum=accum, t_total=t_total)
    # the same steps from the same parameters through the storage-rounding oracle (the HIP path's GELU placement): reported
    tr16 = ots.OracleTrainer(p0, ocfg, start, stop, x_idx, lr=lr, lr_rate=lr_rate, accum=accum, t_total=t_total,
                             bf16_points=True, gelu_stored="acc")
    opt = engine.FusedAdamW(tg.arena, lr=lr, lr_rate=lr_rate, t_total=t_total)
    dev_b = [kb.to_device(b, DEV) for b in batches]
    ob = [dict(input_ids=torch.from_numpy(b["input_ids"]), attention_mask=torch.from_numpy(b["attention_mask"]),
               first_idx=torch.from_numpy(b["first_idx"]), tags=torch.from_numpy(b["tags"].astype(np.int64)),
               lengths=torch.from_numpy(b["lengths"].astype(np.int64))) for b in batches]
    lh, lo, nh, no, n16 = [], [], [], [], []
    for _ in range(steps):
        for k in range(accum):
            lh.append(float(tg.forward_loss(dev_b[k], loss_scale=1.0 / accum, backward=True)))
            lo.append(tr.micro_batch(ob[k]))
            tr16.micro_batch(ob[k])
        nh.append(float(opt.step().sqrt()))
        no.append(tr.optimizer_step(max_norm=5.0))
        n16.append(tr16.optimizer_step(max_norm=5.0))
    torch.cuda.synchronize()
    p1 = oracle_params(tg, round_gemm=False)
    res = {"loss_hip": lh, "loss_oracle": lo, "norm_hip": nh, "norm_oracle": no,
           "loss_rel_max": max(abs(a - b_) / abs(b_) for a, b_ in zip(lh, lo)),
           "norm_rel_max": max(abs(a - b_) / abs(b_) for a, b_ in zip(nh, no)),
           "loss_decreased": lh[-accum] < lh[0]}
    # attribution of the clip-norm distance (assert_train_steps): what storage rounding alone does to this figure
    spread = ots.clip_norm_rounding_spread(steps=2, accum=accum, lr=lr, lr_rate=lr_rate, t_total=t_total)
    res["norm_sigma_storage_rounding"] = float(np.sqrt((spread["acc"] ** 2).mean()))
    res["norm_sigma_storage_rounding_pre"] = float(np.sqrt((spread["pre"] ** 2).mean()))
    res["norm_rel_sigmas"] = res["norm_rel_max"] / res["norm_sigma_storage_rounding"]
    res["norm_storage_oracle"] = n16
    res["norm_rel_storage_oracle_vs_fp32"] = [(a - b_) / b_ for a, b_ in zip(n16, no)]
    res["norm_rel_hip_vs_fp32"] = [(a - b_) / b_ for a, b_ in zip(nh, no)]
    # parameter movement: direction agreement of (after - before) for the tensors that carry most of the update
    worst = 1.0
    for k in ("transitions", "linear.weight", "encoder.layer.1.output.dense.weight", "encoder.layer.0.attention.self.value.weight",
              "embeddings.LayerNorm.weight"):
        dh = (p1[k] - p0[k]).double().flatten()
        do = (tr.params[k].detach() - p0[k]).double().flatten()
        live = do.abs() > 0.2 * do.abs().max()          # Adam's first steps move every weight by ~lr: compare the clear ones
        c = float((dh[live] @ do[live]) / (dh[live].norm() * do[live].norm() + 1e-30))
        res["dcos_" + k] = c
        worst = min(worst, c)
    res["delta_cos_min"] = worst
    tmask = p0["transitions"] > -1e11
    res["transitions_maxabs"] = float((p1["transitions"] - tr.params["transitions"].detach())[tmask].abs().max())
    res["transitions_moved"] = float((tr.params["transitions"].detach() - p0["transitions"])[tmask].abs().max())
    return res


def check_adamw(n=4096 + 64, seed=0):
    from oracle import optim as oopt
    rng = np.random.default_rng(seed)
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    pd = torch.from_numpy(p.copy()).to(DEV)
    md = torch.zeros(n, dtype=F32, device=DEV)
    vd = torch.zeros(n, dtype=F32, device=DEV)
    sh = torch.zeros(n, dtype=BF16, device=DEV)
    from kbner import lib as L
    ws = torch.zeros(L.load().kbner_sqnorm_ws_floats(), dtype=F32, device=DEV)
    nsq = torch.zeros(1, dtype=F32, device=DEV)
    worst = 0.0
    for step in range(1, 4):
        g = (rng.standard_normal(n) * (3.0 if step == 2 else 0.01)).astype(np.float32)
        gd = torch.from_numpy(g.copy()).to(DEV)
        ops.grad_sqnorm(gd, ws, nsq)
        lr = 1e-3
        import math
        bc = math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        ops.adamw(pd, gd, md, vd, sh, n, lr * bc, 0.0, 0.9, 0.999, 1e-6, nsq, 5.0, 1.0, True)
        torch.cuda.synchronize()
        norm = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        coef = oopt.clip_coef(norm, 5.0)
        oopt.adamw_hf_step(p, g * np.float32(coef), m, v, step, lr)
        worst = max(worst, float(np.abs(pd.cpu().numpy() - p).max()))
        assert float(gd.abs().max()) == 0.0
        worst = max(worst, float(np.abs(float(nsq) - norm * norm) / (norm * norm)))
    shadow_err = float((sh.float().cpu() - torch.from_numpy(p)).abs().max())
    return {"p_abs": worst, "shadow_abs": shadow_err}


# ---- ONE definition of every tolerance that smoke() and the pytest wrappers (tests/test_gpu_kernels.py) share: both call
# the assert_* functions below, so the two copies of a threshold that diverged in round 4 (GPUTEST_r04: smoke red, pytest
# green) cannot exist.
STEP_TOL = {            # check_step(): 3x what the round-1 driver run observed (loss 2.3e-4, emissions 5.6e-3, worst gradient
    "loss_rel": 7e-4,   # 0.0121 / cosine 0.99993, head 4.5e-3, transitions 5.1e-4)
    "emissions_rel": 1.7e-2, "grad_min_cos": 0.9998, "grad_worst_rel": 0.037, "grad_linear.weight": 1.4e-2,
    "grad_transitions": 1.6e-3}
TRAIN_TOL = {           # check_train_steps(): 3x round 1's driver observations (loss 4.7e-4, update cosine 0.9995,
    "loss_rel_max": 1.5e-3, "delta_cos_min": 0.9985, "transitions_rel": 2.1e-3,   # transitions 7e-4 of the move), and
    "norm_sigmas": 4.0,          # the clip norm within 4 sigma of what bf16 storage rounding ALONE does to it (below)
    "placement_ratio": 2.0}      # ... where the two GELU placements must have the same sigma within this factor


def assert_step(r):
    assert r["loss_rel"] < STEP_TOL["loss_rel"], r
    assert r["emissions_rel"] < STEP_TOL["emissions_rel"], r
    assert r["grad_min_cos"] > STEP_TOL["grad_min_cos"] and r["grad_worst_rel"] < STEP_TOL["grad_worst_rel"], r
    assert r["grad_linear.weight"] < STEP_TOL["grad_linear.weight"] and r["grad_transitions"] < STEP_TOL["grad_transitions"], r
    assert r["viterbi_equal"], r


def assert_train_steps(r):
    """The clip-norm tolerance is DERIVED, not picked: oracle/train_step.py clip_norm_rounding_spread runs the fp32 oracle
    trainer next to storage-rounding oracle trainers over 16 independent (weights, batches) draws of this tiny tagger.  The
    norm's relative deviation is a zero-mean draw with sigma = 2.6-2.8e-4 (max over 32 draws 6.6-6.9e-4) for BOTH placements
    of the GELU rounding -- so the HIP path's 1.2-1.7e-4 of rounds 1-3 and its 5.2e-4 since round 4 evaluates GELU on the fp32
    accumulator are two ordinary draws of one distribution (0.5 and 1.9 sigma); round 4's story of a rounding bias that had
    been cancelling the others is refuted by the same ensemble (equal sigmas, means -1.5e-4 / -6e-5 of either sign)."""
    assert r["loss_rel_max"] < TRAIN_TOL["loss_rel_max"], r
    sa, sp = r["norm_sigma_storage_rounding"], r["norm_sigma_storage_rounding_pre"]
    assert 1.0 / TRAIN_TOL["placement_ratio"] < sa / sp < TRAIN_TOL["placement_ratio"], r      # the attribution
    assert 1e-4 < sa < 6e-4, r                                                                 # the ensemble itself is sane
    assert r["norm_rel_max"] < TRAIN_TOL["norm_sigmas"] * sa, r
    # ... and the storage-rounding oracle run from the SAME parameters and batches lands where the HIP path does (GPU box,
    # round 5: oracle -4.2e-4 / -3.1e-4, HIP -5.2e-4 / -1.7e-4 against the fp32 trainer: same sign, ratio of the maxima 0.80)
    # ONE-SIDED (round 6): the HIP path may be closer to the fp32 trainer than the rounding model is -- rounds 1-3's kernels (1.2-1.7e-4
    # against the oracle's 4.2e-4) were; a two-sided band would have failed them for being accurate.  What is bounded is how far
    # the HIP distance may EXCEED what storage rounding explains from the same parameters: twice the oracle's distance, or the
    # 4-sigma bound above when the oracle's own draw happens to be small.
    so = max(abs(x) for x in r["norm_rel_storage_oracle_vs_fp32"])
    assert r["norm_rel_max"] < max(TRAIN_TOL["norm_sigmas"] * sa, 2.0 * so), r
    assert r["loss_decreased"], r
    assert r["delta_cos_min"] > TRAIN_TOL["delta_cos_min"], r
    assert r["transitions_maxabs"] < TRAIN_TOL["transitions_rel"] * r["transitions_moved"], r


def smoke():
    r = check_step()
    print("smoke:", r)
    assert_step(r)
    t = check_train_steps(steps=2)
    print("smoke train steps:", {k: v for k, v in t.items() if not k.startswith("dcos_")})
    assert_train_steps(t)
