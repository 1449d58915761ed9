"""TEST INFRASTRUCTURE -- CPU fp32 restatement (torch, autograd) of one fine-tuning step of the
hot path, used (a) as the gradient oracle in tests and (b) as bench.py's `cpu_baseline` leg
(kind "port"), timed on the GPU box's host cores.  Product code never imports this.

Follows flair/trainers/finetune_trainer.py:927-1023 (forward_loss, /accum, backward, every
`accum` micro-batches clip_grad_norm_(5.0) + AdamW.step + zero_grad + scheduler.step) and
flair/models/sequence_tagger_model.py:844-1052,1899-1921,2426-2506 (forward, linear head,
remove_x compaction, forward_score - gold_score, mean).
"""
import math

import torch
import torch.nn.functional as F

from . import encoder as enc

NEG = -1e12


def crf_nll_torch(feats, tags, lens, trans, start, stop):
    """Differentiable (forward - gold) per sentence; same recurrences as oracle/crf.py."""
    B, n, T = feats.shape
    alpha = torch.full((B, T), NEG, dtype=feats.dtype)
    alpha[:, start] = 0.0
    lens = torch.as_tensor(lens, dtype=torch.int64)
    final = alpha.clone()
    final_set = lens == 0
    for i in range(n):
        tag_var = (feats[:, i, :, None] + trans[None, :, :]) + alpha[:, None, :]
        alpha = torch.logsumexp(tag_var, dim=2)
        sel = (lens == i + 1)[:, None]
        final = torch.where(sel, alpha, final)
    logz = torch.logsumexp(final + trans[stop][None, :], dim=1)
    mask = (torch.arange(n)[None, :] < lens[:, None])
    tags = torch.as_tensor(tags, dtype=torch.int64)
    emis = (torch.gather(feats, 2, tags[:, :, None])[:, :, 0] * mask).sum(1)
    frm = torch.cat([torch.full((B, 1), start, dtype=torch.int64), tags], 1)
    to = torch.cat([tags, torch.full((B, 1), stop, dtype=torch.int64)], 1)
    m2 = torch.cat([mask, torch.zeros(B, 1, dtype=torch.bool)], 1)
    to = torch.where(m2, to, torch.full_like(to, stop))
    tmask = torch.cat([torch.ones(B, 1, dtype=torch.bool), mask], 1)
    ts = (trans[to, frm] * tmask).sum(1)
    return logz - (ts + emis)


def tagger_forward_loss(params, cfg, batch, start, stop, x_idx, masks=None, word_keep=None, bf16_points=False, gelu_stored=None):
    """batch: dict(input_ids[B,S], attention_mask[B,S], first_idx[B,n], tags[B,n], lengths[B]).
    params additionally holds 'linear.weight' [T,H], 'linear.bias' [T], 'transitions' [T,T].
    masks: explicit encoder dropout multipliers (encoder_forward); word_keep: bool[n], flair.nn.WordDropout's per-POSITION
    mask (flair/nn.py:176-183: one Bernoulli per token position shared by the whole batch, no rescale)."""
    hidden = enc.encoder_forward(params, cfg, batch["input_ids"], batch["attention_mask"], masks=masks, bf16_points=bf16_points,
                                 gelu_stored=gelu_stored)
    pooled = enc.gather_first_subtoken(hidden, batch["first_idx"], batch.get("first_row"))
    if bf16_points and bf16_points != "flash_exact":   # the pooled rows are a bf16 tensor on the HIP path; the head runs in fp32
        pooled = enc.round_bf16(pooled)
    if word_keep is not None:
        pooled = pooled * word_keep.to(pooled.dtype)[None, :, None]
    emis = F.linear(pooled, params["linear.weight"], params["linear.bias"])
    tags = batch["tags"]
    lengths = batch["lengths"]
    B, n, T = emis.shape
    keep = (torch.arange(n)[None, :] < lengths[:, None])
    if x_idx is not None:
        keep = keep & (tags != x_idx)
    lens = keep.sum(1)
    nmax = int(lens.max())
    cf = emis.new_zeros(B, nmax, T)
    ct = torch.zeros(B, nmax, dtype=torch.int64)
    for b in range(B):
        idx = torch.nonzero(keep[b])[:, 0]
        cf[b, :len(idx)] = emis[b, idx]
        ct[b, :len(idx)] = tags[b, idx]
    nll = crf_nll_torch(cf, ct, lens, params["transitions"], start, stop)
    return nll.mean(), emis


class OracleTrainer:
    """AdamW state + step for a dict of fp32 leaf tensors (two param groups as the reference
    builds them, finetune_trainer.py:552-571: `transitions` at lr*lr_rate, the rest at lr)."""

    def __init__(self, params, cfg, start, stop, x_idx, lr=5e-6, lr_rate=10000, accum=1, t_total=1000, bf16_points=False,
                 gelu_stored=None):
        """bf16_points / gelu_stored: run every micro-batch through the storage-rounding pass of oracle/encoder.py (the
        attribution of the HIP path's distance from the fp32 trainer; the optimiser itself stays fp32 as on the HIP path)."""
        self.bf16_points, self.gelu_stored = bf16_points, gelu_stored
        self.params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        self.cfg, self.start, self.stop, self.x_idx = cfg, start, stop, x_idx
        self.lr, self.lr_rate, self.accum, self.t_total = lr, lr_rate, accum, t_total
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.step_count = 0

    def micro_batch(self, batch):
        loss, _ = tagger_forward_loss(self.params, self.cfg, batch, self.start, self.stop, self.x_idx,
                                      bf16_points=self.bf16_points, gelu_stored=self.gelu_stored)
        (loss / self.accum).backward()
        return float(loss.detach())

    @torch.no_grad()
    def optimizer_step(self, max_norm=5.0):
        sq = 0.0
        for p in self.params.values():
            if p.grad is not None:
                sq += float((p.grad.double() ** 2).sum())
        norm = math.sqrt(sq)
        coef = max_norm / (norm + 1e-6)
        lam = max(0.0, (self.t_total - self.step_count) / max(1, self.t_total))
        self.step_count += 1
        t = self.step_count
        b1, b2, eps = 0.9, 0.999, 1e-6
        for k, p in self.params.items():
            if p.grad is None:
                continue
            g = p.grad * coef if coef < 1.0 else p.grad
            lr = self.lr * lam * (self.lr_rate if k == "transitions" else 1.0)
            m, v = self.m[k], self.v[k]
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            ss = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            p.addcdiv_(m, v.sqrt().add_(eps), value=-ss)
            p.grad = None
        return norm


def clip_norm_rounding_spread(n_seeds=16, steps=2, accum=2, V=512, H=128, L=2, A=2, F_=256, S=64, T=29, std=0.08, start=27, stop=28,
                              x_idx=9, lr=2e-4, lr_rate=50.0, t_total=10, first_seed=100):
    """How far bf16 STORAGE ROUNDING alone moves the clip norm (the gradient norm `optimizer_step` returns) of whole optimiser
    steps on the tiny tagger tests/selftest.py check_train_steps uses: for `n_seeds` independent (weights, batches) draws the fp32
    trainer is run next to two storage-rounding trainers (bf16_points=True with the FFN activation handled as the HIP epilogue
    does: gelu_stored "acc" = GELU / GELU' on the fp32 accumulator, the round-4 placement; "pre" = on its bf16 rounding, rounds
    1-3).  Returns the relative norm deviations {placement: float64[n_seeds * steps]} -- their RMS is the scale any bf16-storing
    implementation of this step sits at, which is what check_train_steps' clip-norm tolerance is derived from, and the two
    placements having the same RMS is the evidence that the placement is not what moved the HIP path's figure in round 4."""
    import numpy as np
    from kbner import batch as kb   # the SURVEY 8(d) synthetic generator (integers only; no device code)
    cfg = enc.EncoderConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=A, intermediate_size=F_,
                            max_position_embeddings=S + 2)
    out = {"acc": [], "pre": []}
    for seed in range(first_seed, first_seed + n_seeds):
        p = enc.init_params(cfg, seed=seed, std=std)
        g = torch.Generator().manual_seed(seed)
        p["linear.weight"] = (torch.rand(T, H, generator=g) * 2 - 1) / math.sqrt(H)
        p["linear.bias"] = torch.zeros(T)
        t = torch.randn(T, T, generator=g)
        t[start, :] = NEG
        t[:, stop] = NEG
        p["transitions"] = t
        bs = []
        for k in range(accum):
            b = kb.synthetic_batch(2, S, vocab=V, T=T, x_idx=x_idx, start=start, stop=stop, n_real=6, seed=seed * 7 + k)
            bs.append(dict(input_ids=torch.from_numpy(b["input_ids"]), attention_mask=torch.from_numpy(b["attention_mask"]),
                           first_idx=torch.from_numpy(b["first_idx"]), tags=torch.from_numpy(b["tags"].astype(np.int64)),
                           lengths=torch.from_numpy(b["lengths"].astype(np.int64))))
        norms = {}
        for mode in (None, "acc", "pre"):
            tr = OracleTrainer(p, cfg, start, stop, x_idx, lr=lr, lr_rate=lr_rate, accum=accum, t_total=t_total,
                               bf16_points=mode is not None, gelu_stored=mode)
            ns = []
            for _ in range(steps):
                for k in range(accum):
                    tr.micro_batch(bs[k])
                ns.append(tr.optimizer_step(max_norm=5.0))
            norms[mode] = np.asarray(ns)
        for mode in ("acc", "pre"):
            out[mode] += list((norms[mode] - norms[None]) / norms[None])
    return {k: np.asarray(v) for k, v in out.items()}
