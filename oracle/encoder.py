"""TEST INFRASTRUCTURE -- CPU fp32 restatement (plain torch ops) of the encoder the reference calls.

The encoder arithmetic is third-party: HuggingFace `transformers==3.0.0` `XLMRobertaModel`
(requirements.txt:30), loaded at flair/embeddings.py:2951-2953 and called at
flair/embeddings.py:3269.  That package is not vendored under /root/reference, so this file
restates its published algorithm (modeling_bert.BertModel with modeling_roberta position ids):

  position_ids = cumsum(ids != pad) * (ids != pad) + pad                (pad_token_id = 1)
  x0 = LN(word[ids] + pos[position_ids] + type[0]), eps = 1e-5
  per layer:  q,k,v = x W{q,k,v}^T + b ; P = softmax(q k^T / sqrt(d) + (1-mask) * -10000) ; c = P v
              x  = LN(x + c Wo^T + bo) ; x = LN(x + gelu_erf(x W1^T + b1) W2^T + b2)

Pinned against transformers 5.15 `XLMRobertaModel` (eager, fp32) run in the build container on
random-init weights: tests/golden/encoder_tiny.npz (oracle/gen_golden.py).  The reference's own
tests hold nothing for this boundary -> parity for the encoder is pinned only by
those container-generated vectors.

Only tests/, smoke() and bench.py's cpu_baseline leg may import this; product code never does.
Weight names are HF state_dict names without the model prefix.
"""
import math

import torch
import torch.nn.functional as F


class EncoderConfig:
    def __init__(self, vocab_size=250002, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                 intermediate_size=4096, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
                 layer_norm_eps=1e-5):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.pad_token_id = pad_token_id
        self.layer_norm_eps = layer_norm_eps

    @staticmethod
    def large():
        return EncoderConfig()

    @staticmethod
    def base():
        return EncoderConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)


def param_shapes(cfg):
    """HF state_dict names -> shapes (encoder only; the pooler is unused on this path because
    sentence_feat is False, flair/embeddings.py:3270-3271)."""
    H, F_, V, P, TV = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size
    shapes = {
        "embeddings.word_embeddings.weight": (V, H),
        "embeddings.position_embeddings.weight": (P, H),
        "embeddings.token_type_embeddings.weight": (TV, H),
        "embeddings.LayerNorm.weight": (H,),
        "embeddings.LayerNorm.bias": (H,),
    }
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layer.%d." % i
        shapes[p + "attention.self.query.weight"] = (H, H)
        shapes[p + "attention.self.query.bias"] = (H,)
        shapes[p + "attention.self.key.weight"] = (H, H)
        shapes[p + "attention.self.key.bias"] = (H,)
        shapes[p + "attention.self.value.weight"] = (H, H)
        shapes[p + "attention.self.value.bias"] = (H,)
        shapes[p + "attention.output.dense.weight"] = (H, H)
        shapes[p + "attention.output.dense.bias"] = (H,)
        shapes[p + "attention.output.LayerNorm.weight"] = (H,)
        shapes[p + "attention.output.LayerNorm.bias"] = (H,)
        shapes[p + "intermediate.dense.weight"] = (F_, H)
        shapes[p + "intermediate.dense.bias"] = (F_,)
        shapes[p + "output.dense.weight"] = (H, F_)
        shapes[p + "output.dense.bias"] = (H,)
        shapes[p + "output.LayerNorm.weight"] = (H,)
        shapes[p + "output.LayerNorm.bias"] = (H,)
    return shapes


def init_params(cfg, seed=20220711, std=0.02, device="cpu"):
    """Random-init weights of the architecture: N(0, 0.02) matrices / embeddings, zero biases,
    unit LayerNorm gains (HF BertPreTrainedModel._init_weights), pad row of word/position
    embeddings zero (nn.Embedding padding_idx)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("LayerNorm.weight"):
            t = torch.ones(shp)
        elif name.endswith(".bias"):
            t = torch.zeros(shp)
        else:
            t = torch.empty(shp).normal_(0.0, std, generator=g)
            if name in ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight"):
                t[cfg.pad_token_id].zero_()
        out[name] = t.to(device)
    return out


def position_ids_from_input_ids(input_ids, pad_id):
    m = (input_ids != pad_id).to(torch.int64)
    return torch.cumsum(m, dim=1) * m + pad_id


class _RoundBf16(torch.autograd.Function):
    """value AND the gradient flowing back through it rounded to bfloat16 (round-to-nearest-even): a tensor the HIP path keeps
    in bf16 in both directions (activation forward, its dY backward)"""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def round_bf16(t):
    return _RoundBf16.apply(t)


def round_bf16_weight(w):
    """the bf16 shadow of an fp32 master weight: rounded value forward, fp32 gradient (straight through) backward"""
    return w + (w.to(torch.bfloat16).to(w.dtype) - w).detach()


class _AttnCoreFlash(torch.autograd.Function):
    """softmax(Q K^T / sqrt(d) + ext) (* dropout multiplier) . V with the BACKWARD the HIP kernels run (csrc/attention.hip
    attn_bwd_dq2 / attn_bwd_dkv2, the FlashAttention-2 recomputation): P recomputed in fp32, the softmax-backward correction
    taken as D = rowdot(dO, O) from the STORED bf16 O (instead of sum_j P dP), dS and P rounded to bf16 for the dQ / dK / dV
    products.  o_split: how O is stored for D -- False = bf16 (round 3), True = bf16 + the e5m2 byte of the residual * 2^14
    (round 4, kbner_attn_fwd's ctx_lo)."""

    @staticmethod
    def forward(ctx, q, k, v, ext, mult, o_split):
        """o_split: False / True as above; "exact" = no rounding anywhere (the self-check of these backward formulas against
        autograd, tests/test_oracle_golden.py)"""
        d = q.shape[-1]
        p = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d) + ext, dim=-1)
        pm = p if mult is None else p * mult
        pb = pm if o_split == "exact" else pm.to(torch.bfloat16).to(q.dtype)
        o = torch.matmul(pb, v)
        ctx.save_for_backward(q, k, v, p, pb, o, mult if mult is not None else torch.tensor(0.0))
        ctx.has_mult, ctx.o_split = mult is not None, o_split
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p, pb, o, mult = ctx.saved_tensors
        d = q.shape[-1]
        rb = (lambda t: t) if ctx.o_split == "exact" else (lambda t: t.to(torch.bfloat16).to(q.dtype))   # noqa: E731
        o_st = rb(o)
        if ctx.o_split is True:
            o_st = o_st + ((o - o_st) * 16384.0).to(torch.float32).to(torch.float8_e5m2).to(o.dtype) / 16384.0
        dd = (do * o_st).sum(-1, keepdim=True)
        dp = torch.matmul(do, v.transpose(-1, -2))
        if ctx.has_mult:
            dp = dp * mult
        ds = rb(p * (dp - dd))
        dq = torch.matmul(ds, k) / math.sqrt(d)
        dk = torch.matmul(ds.transpose(-1, -2), q) / math.sqrt(d)
        dv = torch.matmul(pb.transpose(-1, -2), do)
        return dq, dk, dv, None, None, None


class _GeluStored(torch.autograd.Function):
    """erf-GELU the way the FFN-up GEMM epilogue runs it (csrc/gemm256.hip, EPI_GELU): the activation AND gelu'(pre) leave the
    epilogue as bf16 tensors, backward is dpre = bf16(dact * stored gelu') (EPI_DGELU).  pre_rounded=False: both evaluated on the
    fp32 accumulator (round 4 onwards); True: on its bf16 rounding, as if the pre-activation had been stored (rounds 1-3)."""

    @staticmethod
    def forward(ctx, pre, pre_rounded):
        rb = lambda t: t.to(torch.bfloat16).to(t.dtype)   # noqa: E731
        p = rb(pre) if pre_rounded else pre
        gp = 0.5 * (1.0 + torch.erf(p * (1.0 / math.sqrt(2.0)))) + p * torch.exp(-0.5 * p * p) * (1.0 / math.sqrt(2.0 * math.pi))
        ctx.save_for_backward(rb(gp))
        return rb(F.gelu(p))

    @staticmethod
    def backward(ctx, g):
        (gp,) = ctx.saved_tensors
        return (g * gp).to(torch.bfloat16).to(g.dtype), None


def encoder_forward(params, cfg, input_ids, attention_mask, return_all=False, masks=None, bf16_points=False, gelu_stored=None):
    """input_ids int64[B,S], attention_mask {0,1}[B,S] -> last hidden state f32[B,S,H]
    (== hidden_states[-1], the only layer the path uses: `layers: '-1'`).
    gelu_stored (with bf16_points): None = autograd through F.gelu with the activation rounded (the default storage-rounding
    pass); "acc" / "pre" = _GeluStored on the fp32 accumulator / on its bf16 rounding (attribution of the clip-norm distance,
    tests/selftest.py check_train_steps).

    masks (training-mode dropout, transformers 3.0.0 modeling_bert: BertEmbeddings.forward dropout after the LayerNorm,
    BertSelfAttention `attention_probs = self.dropout(attention_probs)`, BertSelfOutput / BertOutput dense -> dropout ->
    LayerNorm(h + input)): optional dict of explicit MULTIPLIERS (0 or 1/(1-p)) so a test can feed the exact mask the
    HIP kernels regenerate from their seeds: "emb" [B,S,H], ("attn", i) [B,A,S,S], ("o", i) / ("ffn", i) [B,S,H]."""
    masks = masks or {}
    mul = lambda t, key: t * masks[key] if key in masks else t  # noqa: E731
    # bf16_points (tests/selftest.py check_step(bf16_oracle=True): attribution of the HIP path's gradient error): round to bf16
    # wherever the HIP path stores bf16 -- GEMM weights (shadow), every activation tensor between kernels (and, through autograd,
    # the gradient that comes back through it), the attention probabilities fed to P.V -- and nowhere else (fp32 accumulation,
    # fp32 LayerNorm statistics, fp32 softmax as in the kernels)
    r = round_bf16 if bf16_points and bf16_points != "flash_exact" else (lambda t: t)
    rw = round_bf16_weight if bf16_points and bf16_points != "flash_exact" else (lambda w: w)
    H, A = cfg.hidden_size, cfg.num_attention_heads
    d = H // A
    B, S = input_ids.shape
    eps = cfg.layer_norm_eps
    pos = position_ids_from_input_ids(input_ids, cfg.pad_token_id)
    x = (params["embeddings.word_embeddings.weight"][input_ids]
         + params["embeddings.position_embeddings.weight"][pos]
         + params["embeddings.token_type_embeddings.weight"][0])
    x = F.layer_norm(x, (H,), params["embeddings.LayerNorm.weight"], params["embeddings.LayerNorm.bias"], eps)
    x = r(mul(x, "emb"))
    ext = (1.0 - attention_mask.to(x.dtype))[:, None, None, :] * -10000.0
    hs = [x]
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layer.%d." % i
        q = r(F.linear(x, rw(params[p + "attention.self.query.weight"]), params[p + "attention.self.query.bias"]))
        k = r(F.linear(x, rw(params[p + "attention.self.key.weight"]), params[p + "attention.self.key.bias"]))
        v = r(F.linear(x, rw(params[p + "attention.self.value.weight"]), params[p + "attention.self.value.bias"]))
        q = q.view(B, S, A, d).transpose(1, 2)
        k = k.view(B, S, A, d).transpose(1, 2)
        v = v.view(B, S, A, d).transpose(1, 2)
        if bf16_points in ("flash", "flash_split", "flash_exact"):    # + the backward formulas of the attention kernels (see _AttnCoreFlash)
            c = _AttnCoreFlash.apply(q, k, v, ext, masks.get(("attn", i)),
                                     "exact" if bf16_points == "flash_exact" else bf16_points == "flash_split")
            c = r(c.transpose(1, 2).reshape(B, S, H))
        else:
            sc = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d) + ext
            pr = r(mul(torch.softmax(sc, dim=-1), ("attn", i)))
            c = r(torch.matmul(pr, v).transpose(1, 2).reshape(B, S, H))
        o = mul(F.linear(c, rw(params[p + "attention.output.dense.weight"]), params[p + "attention.output.dense.bias"]), ("o", i))
        x = r(F.layer_norm(r(o + x), (H,), params[p + "attention.output.LayerNorm.weight"],
                           params[p + "attention.output.LayerNorm.bias"], eps))
        pre = F.linear(x, rw(params[p + "intermediate.dense.weight"]), params[p + "intermediate.dense.bias"])
        h = _GeluStored.apply(pre, gelu_stored == "pre") if (gelu_stored and bf16_points) else r(F.gelu(pre))
        o = mul(F.linear(h, rw(params[p + "output.dense.weight"]), params[p + "output.dense.bias"]), ("ffn", i))
        x = r(F.layer_norm(r(o + x), (H,), params[p + "output.LayerNorm.weight"], params[p + "output.LayerNorm.bias"], eps))
        hs.append(x)
    if return_all:
        return x, hs
    return x


def stitch_windows(window_states, stride):
    """Seam rule for a sentence encoded as several overlapping windows (flair/embeddings.py:3292-3299): drop the last
    1 + stride//2 positions of the states accumulated so far (</s> + half the overlap) and the first 1 + stride//2 of
    the next window (<s> + the other half), concatenate.  window_states: list of f32[S_w,H] (real positions only)."""
    acc = window_states[0]
    for nxt in window_states[1:]:
        acc = torch.cat((acc[:-1 - stride // 2], nxt[1 + stride // 2:]), 0)
    return acc


def gather_first_subtoken(hidden, first_idx, first_row=None):
    """first-subtoken pooling (flair/embeddings.py:3288-3345 with pooling_operation 'first') +
    assign_batch_features zero padding (:108-124).  hidden f32[R,S,H]; first_idx int64[B,n] holds
    the subtoken position of each word token's first piece, or -1 for padding / tokens with zero
    subtokens (-> zero vector, :3306-3308); first_row int64[B,n] the encoder row it lives in (default b:
    one row per sentence; differs when long sentences were split into windows).  Returns f32[B,n,H]."""
    R, S, H = hidden.shape
    B = first_idx.shape[0]
    rows = torch.arange(B)[:, None] if first_row is None else first_row
    flat = (rows * S + first_idx.clamp(min=0)).reshape(-1)
    out = hidden.reshape(R * S, H)[flat].reshape(B, -1, H)
    return out * (first_idx >= 0).to(hidden.dtype)[:, :, None]
