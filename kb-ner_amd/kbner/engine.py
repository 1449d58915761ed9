"""Host-side engine of the hot path: parameter arena, XLM-R encoder forward/backward, emission
head + CRF loss, fused AdamW -- all arithmetic in libkbner_hip.so (kbner.ops); torch only owns
device memory and the stream.  No autograd: backward is explicit, mirroring what
loss.backward() does in flair/trainers/finetune_trainer.py:957.

Data layout in HBM (sized for 288 GB: everything stays resident, nothing is recomputed):
  * ONE flat fp32 arena for parameters with identically laid-out arenas for grads / Adam m / v,
    so clip-norm and AdamW are single streaming launches.  GEMM weights come first and have a
    bf16 "shadow" (what MFMA reads), rewritten by the AdamW kernel itself.
  * activations are token-major bf16 [M_pad, width] (M = B*S padded to 128 rows); per layer the
    engine keeps x, qkv, ctx, h1, x1, gelu'(pre), act, h2 (+ fp32 LN stats, softmax lse) for backward.
"""
import logging
import math

import os

import torch

from . import lib as L_
from . import ops
from .dp import world_size as dp_world_size
from .lib import (EPI_ADD, EPI_ATOMIC32, EPI_BIAS, EPI_COLSUM, EPI_COLSUM_WS, EPI_DGELU, EPI_GELU, EPI_GELU_FWD, EPI_RMW32, EPI_STORE32, GEMM_NN, GEMM_NT,
                  GEMM_TN)

log = logging.getLogger("kbner")

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


class EncoderConfig:
    """Same fields/meaning as the HF XLMRobertaConfig the reference loads (flair/embeddings.py:2952)."""

    def __init__(self, vocab_size=250002, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                 intermediate_size=4096, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
                 layer_norm_eps=1e-5, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.pad_token_id = pad_token_id
        self.layer_norm_eps = layer_norm_eps
        self.hidden_dropout_prob = hidden_dropout_prob  # active only while Tagger.training (model.train())
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        if hidden_size != num_attention_heads * 64:
            raise ValueError("kbner attention kernels need head_dim 64 (XLM-R base/large)")
        if hidden_size % 128 or intermediate_size % 128:
            raise ValueError("hidden/intermediate sizes must be multiples of 128")

    @staticmethod
    def large(**kw):
        return EncoderConfig(**kw)

    @staticmethod
    def base(**kw):
        return EncoderConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, **kw)


def _round_up(x, m):
    return (x + m - 1) // m * m


class Arena:
    """Flat fp32 parameter storage + grads + Adam state; tensors padded to 8 elements."""

    def __init__(self, specs, device, with_grad=True):
        self.device = device
        self.offsets = {}
        self.shapes = {}
        off = 0
        self.n_shadow = 0
        seen_plain = False
        for name, shape, shadow in specs:
            n = 1
            for s in shape:
                n *= s
            if shadow:
                assert not seen_plain, "shadowed (GEMM) tensors must come first"
            else:
                if not seen_plain:
                    self.n_shadow = off
                seen_plain = True
            self.offsets[name] = off
            self.shapes[name] = tuple(shape)
            off += _round_up(n, 8)
        if not seen_plain:
            self.n_shadow = off
        self.n = off
        self.p = torch.zeros(self.n, dtype=F32, device=device)
        self.g = torch.zeros(self.n, dtype=F32, device=device) if with_grad else None   # frozen (inference) replicas hold no gradients
        self.m = None
        self.v = None
        self.shadow = torch.zeros(max(self.n_shadow, 8), dtype=BF16, device=device)
        # u8[V]: bit 0 = the word-embedding row has EVER received a gradient, bit 1 = it has received one since the optimizer last
        # zeroed it (both set by the embedding backward and by the data-parallel row exchange; include/kbner.h KBNER_ROW_*).  The
        # optimizer skips rows with no bit set -- their g, m, v are exactly 0 -- and the gradient of rows without bit 1, which is.
        self.emb_flags = None
        if with_grad and "emb.word" in self.shapes:
            self.emb_flags = torch.zeros(self.shapes["emb.word"][0], dtype=torch.uint8, device=device)
        # The GEMM-weight gradients g[:n_shadow] need no zeroing when the backward pass that follows an optimizer step OVERWRITES
        # them (the grouped weight-gradient launch with KBNER_EPI_STORE32 instead of the fp32 read-modify-write): 4 B / parameter
        # less written by AdamW and 4 B / parameter less read by the first micro-batch's weight-gradient epilogues.
        # wgrad_overwrite_ok: the owner (Tagger) guarantees that every weight gradient goes through that launch.
        # wgrad_stale: g[:n_shadow] holds the PREVIOUS step's gradients (set by FusedAdamW.step, cleared by encoder_backward).
        self.wgrad_overwrite_ok = False
        self.wgrad_stale = False
        # FusedAdamW.lazy_rows: the state of the lazily updated word-embedding table (row_t i32[V], clock i32[1], hist f32[cap], b1,
        # b2, eps) or None.  While it is set a row of emb.word / its moments may lag behind the optimizer's step count: the encoder
        # forward brings the rows it looks up to date (catch_up_rows), every other reader calls materialize_rows() first.
        self.lazy = None

    def _emb_rows(self, buf):
        lo = self.offsets["emb.word"]
        V, H = self.shapes["emb.word"]
        return buf[lo:lo + V * H].view(V, H)

    def catch_up_rows(self, ids):
        """before an embedding lookup of `ids` (i32, device): the zero-gradient optimizer steps those rows still owe"""
        z = self.lazy
        if z is not None:
            ops.adamw_rows_catchup(ids, self._emb_rows(self.p), self._emb_rows(self.m), self._emb_rows(self.v), self.emb_flags,
                                   z["row_t"], z["clock"], z["hist"], z["b1"], z["b2"], z["eps"])

    def materialize_rows(self):
        """every row brought to the optimizer's step count: p / m / v are the eager optimizer's (before reading them elsewhere)"""
        z = self.lazy
        if z is not None and z["dirty"]:
            self.catch_up_rows(None)
            z["dirty"] = False

    def finalize_grads(self):
        """ONE place for the stale-gradient rule: make `g` what a consumer of "this step's gradients" may read.  If no backward
        pass has overwritten the GEMM-weight gradients since the last optimizer step (wgrad_stale), they are still that step's:
        zero them.  Called by FusedAdamW.step and by the data-parallel exchange (GradReducer.begin_exchange) -- any new consumer
        of arena.g between an optimizer step and the next full backward pass calls it too.  Idempotent."""
        if self.wgrad_stale:
            self.g[:self.n_shadow].zero_()
            self.wgrad_stale = False

    def _view(self, buf, name):
        off, shape = self.offsets[name], self.shapes[name]
        n = 1
        for s in shape:
            n *= s
        return buf[off:off + n].view(shape)

    # views are created once per (buffer, name): the engine asks for ~700 of them per micro-batch
    def _cached(self, tag, buf, name):
        cache = self.__dict__.setdefault("_views", {})
        v = cache.get((tag, name))
        if v is None:
            v = cache[(tag, name)] = self._view(buf, name)
        return v

    def param(self, name):
        return self._cached("p", self.p, name)

    def grad(self, name):
        return self._cached("g", self.g, name)

    def bf(self, name):
        assert self.offsets[name] < self.n_shadow
        return self._cached("s", self.shadow, name)

    def refresh_shadow(self):
        if self.n_shadow:
            ops.f32_to_bf16(self.p[:self.n_shadow], self.shadow[:self.n_shadow])

    def ensure_state(self):
        if self.m is None:
            self.m = torch.zeros_like(self.p)
            self.v = torch.zeros_like(self.p)


def tagger_specs(cfg, T):
    """Arena order: GEMM weights (bf16-shadowed) | embeddings, LN, biases, head | transitions (own lr group,
    flair/trainers/finetune_trainer.py:552-571)."""
    H, F_, V, P, TV = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size
    specs = []
    for i in range(cfg.num_hidden_layers):
        specs += [("l%d.qkv.weight" % i, (3 * H, H), True), ("l%d.o.weight" % i, (H, H), True),
                  ("l%d.ffn1.weight" % i, (F_, H), True), ("l%d.ffn2.weight" % i, (H, F_), True)]
    specs += [("emb.word", (V, H), False), ("emb.pos", (P, H), False), ("emb.type", (TV, H), False),
              ("emb.ln.g", (H,), False), ("emb.ln.b", (H,), False)]
    for i in range(cfg.num_hidden_layers):
        specs += [("l%d.qkv.bias" % i, (3 * H,), False), ("l%d.o.bias" % i, (H,), False),
                  ("l%d.ln1.g" % i, (H,), False), ("l%d.ln1.b" % i, (H,), False),
                  ("l%d.ffn1.bias" % i, (F_,), False), ("l%d.ffn2.bias" % i, (H,), False),
                  ("l%d.ln2.g" % i, (H,), False), ("l%d.ln2.b" % i, (H,), False)]
    specs += [("linear.weight", (T, H), False), ("linear.bias", (T,), False), ("transitions", (T, T), False)]
    return specs


# HF state_dict name <-> arena name (+ row slice for the fused qkv)
def hf_name_map(cfg):
    H = cfg.hidden_size
    m = {
        "embeddings.word_embeddings.weight": ("emb.word", None),
        "embeddings.position_embeddings.weight": ("emb.pos", None),
        "embeddings.token_type_embeddings.weight": ("emb.type", None),
        "embeddings.LayerNorm.weight": ("emb.ln.g", None),
        "embeddings.LayerNorm.bias": ("emb.ln.b", None),
    }
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layer.%d." % i
        for j, nm in enumerate(("query", "key", "value")):
            m[p + "attention.self.%s.weight" % nm] = ("l%d.qkv.weight" % i, (j * H, (j + 1) * H))
            m[p + "attention.self.%s.bias" % nm] = ("l%d.qkv.bias" % i, (j * H, (j + 1) * H))
        m[p + "attention.output.dense.weight"] = ("l%d.o.weight" % i, None)
        m[p + "attention.output.dense.bias"] = ("l%d.o.bias" % i, None)
        m[p + "attention.output.LayerNorm.weight"] = ("l%d.ln1.g" % i, None)
        m[p + "attention.output.LayerNorm.bias"] = ("l%d.ln1.b" % i, None)
        m[p + "intermediate.dense.weight"] = ("l%d.ffn1.weight" % i, None)
        m[p + "intermediate.dense.bias"] = ("l%d.ffn1.bias" % i, None)
        m[p + "output.dense.weight"] = ("l%d.ffn2.weight" % i, None)
        m[p + "output.dense.bias"] = ("l%d.ffn2.bias" % i, None)
        m[p + "output.LayerNorm.weight"] = ("l%d.ln2.g" % i, None)
        m[p + "output.LayerNorm.bias"] = ("l%d.ln2.b" % i, None)
    return m


class _Acts:
    """Activation / workspace buffers for one (B, S) shape; zero-initialised so the 128-row padding
    tail never feeds NaNs into a wgrad reduction."""

    def __init__(self, cfg, B, S, device):
        H, F_, A, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers
        self.B, self.S, self.M = B, S, B * S
        self.Mp = Mp = _round_up(B * S, 256)
        z = lambda *shape, dt=BF16: torch.zeros(shape, dtype=dt, device=device)  # noqa: E731
        self.h0 = z(Mp, H)
        self.emb_mean, self.emb_rstd = z(Mp, dt=F32), z(Mp, dt=F32)
        self.x = [z(Mp, H) for _ in range(L + 1)]
        self.qkv = [z(Mp, 3 * H) for _ in range(L)]
        self.ctx = [z(Mp, H) for _ in range(L)]
        self.ctx_lo = [None] * L   # O - bf16(O) in bytes, allocated by the first training forward (Engine.ATTN_RESIDUAL)
        self.infer_graph = None    # Engine.encoder_forward(need_grad=False): static inputs + the captured HIP graph of this shape
        self.lse = [z(B, A, S, dt=F32) for _ in range(L)]
        self.h1 = [z(Mp, H) for _ in range(L)]
        self.x1 = [z(Mp, H) for _ in range(L)]
        self.dact = [z(Mp, F_) for _ in range(L)]  # gelu'(pre-activation), saved by the FFN-up epilogue for backward
        self.act = [z(Mp, F_) for _ in range(L)]
        self.h2 = [z(Mp, H) for _ in range(L)]
        self.st1 = [(z(Mp, dt=F32), z(Mp, dt=F32)) for _ in range(L)]
        self.st2 = [(z(Mp, dt=F32), z(Mp, dt=F32)) for _ in range(L)]
        # backward workspaces (shared across layers)
        # dY buffers are kept for WGRAD_GROUP layers so their weight-gradient GEMMs can be launched
        # together (4 layers x 4 GEMMs = 768 tiles of 256x256 = 3 full waves of the 256 CUs)
        self.dx = z(Mp, H)
        self.dx1 = z(Mp, H)
        self.dctx = z(Mp, H)
        self.dh = [z(Mp, H) for _ in range(WGRAD_GROUP)]
        self.dh1 = [z(Mp, H) for _ in range(WGRAD_GROUP)]
        self.dpre = [z(Mp, F_) for _ in range(WGRAD_GROUP)]
        self.dqkv = [z(Mp, 3 * H) for _ in range(WGRAD_GROUP)]
        self.dws = z(B, A, S, dt=F32)
        self._z, self._H = z, H
        self.dhm = self.dh1m = None
        self.splitk_ws = None  # f32 [4, Mp, H] split-K slabs, allocated on first use (small micro-batches only)
        self.colsum_ws = None  # f32 [2 * Mp/256, F] column-sum lines of the FFN-down dgrad epilogue (EPI_COLSUM_WS)
        self.defer_ln_ws = None     # small batches: one partial-sum workspace per LayerNorm / per FFN-up bias (Tagger.encoder_backward)
        self.colsum_ws_all = None

    def drop_buffers(self):
        """masked copies of dh / dh1 (the dY of the two GEMMs whose outputs were dropped); allocated on first training use"""
        if self.dhm is None:
            self.dhm = [self._z(self.Mp, self._H) for _ in range(WGRAD_GROUP)]
            self.dh1m = [self._z(self.Mp, self._H) for _ in range(WGRAD_GROUP)]
        return self.dhm, self.dh1m


WGRAD_GROUP = 4


def _splitk(tiles, Mp):
    """split the token (reduction) dimension of a wgrad GEMM so the launch has >= ~768 workgroups"""
    sk = 1
    while tiles * sk < 768 and Mp % (64 * sk * 2) == 0 and sk < 64:
        sk *= 2
    return sk


class Tagger:
    """XLM-R encoder + linear emission head + CRF, parameters in one Arena.

    Reference path: TransformerWordEmbeddings (flair/embeddings.py:2906) -> FastSequenceTagger.forward /
    forward_loss / _calculate_loss (flair/models/sequence_tagger_model.py:844,1899,2426)."""

    def __init__(self, cfg, num_tags, start_idx, stop_idx, device="cuda", inference=False):
        self.cfg, self.T, self.start, self.stop = cfg, num_tags, start_idx, stop_idx
        self.device = torch.device(device)
        self.arena = Arena(tagger_specs(cfg, num_tags), self.device, with_grad=not inference)
        # every weight gradient of the encoder is a tile of the grouped 256 x 256 launch (_wgrads): it may overwrite
        self.arena.wgrad_overwrite_ok = (not inference and cfg.hidden_size % 256 == 0 and cfg.intermediate_size % 256 == 0
                                         and os.environ.get("KBNER_WGRAD_OVERWRITE", "1") != "0")   # (A/B switch: README "Switches")
        self._acts = {}
        self._saved = None
        # Dropout (active only while `training`): the encoder's three HF sites (embeddings, attention probabilities,
        # the two sub-layer outputs) from cfg, plus the tagger's WordDropout (flair/nn.py:166-183: whole token POSITIONS
        # zeroed across the batch, no rescale).  Masks are counter-based (include/kbner.h): only seeds are stored.
        self.training = False
        self.word_dropout = 0.0
        self.seed_dropout(20220711)
        # data-parallel runs: the persistent GEMM draws its tiles dynamically (kbner_gemm_bf16_grouped_dyn) so that CUs taken
        # by an overlapped RCCL collective cost their share of the launch, not a second pass; off for a single process.
        # True: from the moment a micro-batch's first gradient bucket is handed to the reducer (encoder_backward) to the end of
        # its backward pass -- the forward pass and the first WGRAD_GROUP layers of backward stay static launches, i.e. the
        # interleaved-ring loop; "always": every GEMM launch (bench.py --dynamic-tiles, the N = 1 A/B)
        self.dynamic_tiles = False
        self._sched_ring = None

    def seed_dropout(self, seed):
        import numpy as np
        self._drop_rng = np.random.default_rng(int(seed))

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def _site_seeds(self):
        """per forward: (embedding site, [(attention, attn-output, ffn-output) per layer]) as (seed, thresh) pairs"""
        cfg = self.cfg
        L = cfg.num_hidden_layers
        th = ops.drop_thresh(cfg.hidden_dropout_prob) if self.training else 0
        ta = ops.drop_thresh(cfg.attention_probs_dropout_prob) if self.training else 0
        if not (th or ta):
            return ops.NO_DROP, [(ops.NO_DROP, ops.NO_DROP, ops.NO_DROP)] * L
        sd = self._drop_rng.integers(0, 2 ** 32, size=1 + 3 * L, dtype="uint64")
        emb = (int(sd[0]), th)
        return emb, [((int(sd[1 + 3 * l]), ta), (int(sd[2 + 3 * l]), th), (int(sd[3 + 3 * l]), th)) for l in range(L)]

    # ---------------------------------------------------------------- parameters
    def init_random(self, seed=20220711, std=0.02):
        """HF-style random init (N(0,std) matrices/embeddings, zero biases, unit LN gains, zero pad rows); linear head as
        torch.nn.Linear's default; transitions as sequence_tagger_model.py:402-410.  Generated ON the device from a seeded
        generator: every DP rank builds the identical replica without 560 M host-side randn per process."""
        a, cfg = self.arena, self.cfg
        on_gpu = self.device.type == "cuda"
        g = torch.Generator(device=self.device if on_gpu else "cpu").manual_seed(seed)
        dev = self.device if on_gpu else "cpu"
        for name, shape in a.shapes.items():
            dst = a.param(name)
            if name.endswith("ln.g") or name.endswith("ln1.g") or name.endswith("ln2.g"):
                dst.fill_(1.0)
            elif name.endswith(".bias") or name.endswith("ln.b") or name.endswith("ln1.b") or name.endswith("ln2.b"):
                dst.zero_()
            elif name == "transitions":
                t = torch.randn(shape, generator=g, device=dev)
                t[self.start, :] = -1e12
                t[:, self.stop] = -1e12
                dst.copy_(t)
            elif name == "linear.weight":
                bound = 1.0 / math.sqrt(shape[1])
                dst.copy_((torch.rand(shape, generator=g, device=dev) * 2 - 1) * bound)
            else:
                dst.normal_(0.0, std, generator=g) if on_gpu else dst.copy_(torch.empty(shape).normal_(0.0, std, generator=g))
        a.param("emb.word")[cfg.pad_token_id].zero_()
        a.param("emb.pos")[cfg.pad_token_id].zero_()
        a.refresh_shadow()

    def load_hf_state_dict(self, sd, prefix=""):
        """Load encoder weights given under HF names (e.g. from XLMRobertaModel.state_dict())."""
        nm = hf_name_map(self.cfg)
        self.arena.materialize_rows()
        for hf, (mine, sl) in nm.items():
            t = sd[prefix + hf].to(device=self.device, dtype=F32)
            dst = self.arena.param(mine)
            if sl is not None:
                dst = dst[sl[0]:sl[1]]
            dst.copy_(t)
        self.arena.refresh_shadow()

    def hf_state_dict(self):
        self.arena.materialize_rows()
        out = {}
        for hf, (mine, sl) in hf_name_map(self.cfg).items():
            t = self.arena.param(mine)
            out[hf] = (t[sl[0]:sl[1]] if sl is not None else t).detach().clone()
        return out

    def set_param(self, name, value):
        self.arena.materialize_rows()
        self.arena.param(name).copy_(torch.as_tensor(value).to(device=self.device, dtype=F32))
        if self.arena.offsets[name] < self.arena.n_shadow:
            self.arena.refresh_shadow()

    # the attention forward also keeps O - bf16(O) (one byte per element and layer) and the backward takes its softmax
    # correction D from the pair: include/kbner.h kbner_attn_bwd.  KBNER_ATTN_RESIDUAL=0: D from the bf16 O alone (round 3).
    ATTN_RESIDUAL = os.environ.get("KBNER_ATTN_RESIDUAL", "1") != "0"
    ACTS_BUDGET_BYTES = 120 << 30  # resident activation sets (one per (B, S) shape), least-recently-used first out

    def acts(self, B, S):
        """Activation / workspace buffers for a (B, S) shape.  Length-sorted batches cycle through a handful of S values;
        their buffer sets stay resident (LRU within a byte budget) instead of being re-allocated and zero-filled."""
        key = (B, S)
        if key in self._acts:
            self._acts[key] = self._acts.pop(key)  # move to the most-recent end
            return self._acts[key]
        cfg = self.cfg
        per_row = 2 * (cfg.num_hidden_layers * (8.5 * cfg.hidden_size + 2 * cfg.intermediate_size) + 16 * cfg.hidden_size
                       + 4 * cfg.intermediate_size)
        need = int(_round_up(B * S, 256) * per_row)
        while self._acts and sum(a.nbytes for a in self._acts.values()) + need > self.ACTS_BUDGET_BYTES:
            self._acts.pop(next(iter(self._acts)))
        ac = _Acts(self.cfg, B, S, self.device)
        ac.nbytes = need
        self._acts[key] = ac
        return ac

    SPLITK_MAX_TILES = 64  # outputs with at most this many 256x256 tiles split a long K (small micro-batches)
    FUSE_SPLITK_LN = True  # the LayerNorm behind a split-K GEMM folds the fp32 slabs itself (no kbner_splitk_finish launch)
    DEFER_REDUCE_MAX_TOKENS = 16384  # micro-batches up to this many (padded) sub-tokens batch their column-sum reductions (encoder_backward)

    def _long_k_gemm(self, layout, A, W, Mp, N, K, C, ac, bias=None, addend=None, drop=ops.NO_DROP, finish=True):
        """C = bf16(dropout(A.W + bias) + addend) for the three K >= 3H GEMMs of a layer (FFN-down forward, its two dgrad
        siblings).  With a small micro-batch the [Mp, H] output has a few dozen tiles, each a serial chain of K/64 DMA
        round trips (86 us at K=4096 whatever Mp): K is then cut over up to 4 workgroups per tile (ops.gemm_splitk)."""
        tiles = (Mp // 256) * (N // 256) if Mp % 256 == 0 and N % 256 == 0 else 0
        splits = 0
        if 0 < tiles <= self.SPLITK_MAX_TILES and not ops.FORCE_128:
            for sp in (4, 3, 2):
                if K % (64 * sp) == 0 and K // sp >= 512:
                    splits = sp
                    break
        if splits:
            if ac.splitk_ws is None:
                ac.splitk_ws = torch.empty((4, Mp, N), dtype=F32, device=self.device)
            fold_later = not finish and self.FUSE_SPLITK_LN and ops.GEMM_HOOK is None and ops.HBM_HOOK is None
            ops.gemm_splitk(layout, A, W, Mp, N, K, splits, ac.splitk_ws, C, bias=bias, addend=addend, drop=drop, finish=not fold_later)
            if fold_later:
                return ac.splitk_ws, splits      # the LayerNorm that consumes C folds the slabs itself (C is NOT written)
        else:
            epi = (EPI_BIAS if bias is not None else 0) | (EPI_ADD if addend is not None else 0)
            ops.gemm(layout, A, W, Mp, N, K, C=C, bias=bias, addend=addend, epi=epi, drop=drop, occupancy=True)

    # ---------------------------------------------------------------- encoder
    # Forward-only passes (evaluate, predict, the frozen encoders of a stack) are ~170 launches of ~50 us of Python + ctypes each:
    # 9.7 ms of host time per batch of 32 next to 11.8 ms of device time -- device-bound by a narrow margin on a fast host,
    # host-bound on a slower one (the driver's round-3 box: 1154 instead of 2400-2700 sentences/s).  Nothing in such a pass
    # depends on the host (no dropout seeds, fixed shapes, buffers that live as long as the (B, S) activation set), so the third
    # pass over a shape is captured into a HIP graph and every later one is three small input copies + ONE graph launch.
    # KBNER_INFER_GRAPH=0 keeps the eager launches (A/B).
    INFER_GRAPH = os.environ.get("KBNER_INFER_GRAPH", "1") != "0"
    INFER_GRAPH_DP = os.environ.get("KBNER_INFER_GRAPH_DP", "0") == "1"    # opt in on data-parallel ranks (untested over RCCL)

    def encoder_forward(self, ids, pos_ids, maskbias, B, S, need_grad=True):
        if (need_grad or not self.INFER_GRAPH or self.training or ops.SCHED_RING is not None or ops.GEMM_HOOK is not None
                or self.device.type != "cuda"):
            return self._encoder_forward(ids, pos_ids, maskbias, B, S, need_grad)
        ac = self.acts(B, S)
        st = ac.infer_graph
        if st is None:
            st = ac.infer_graph = {"calls": 0, "graph": None, "ids": torch.empty_like(ids), "pos": torch.empty_like(pos_ids),
                                   "mb": torch.empty_like(maskbias), "variant": ops.gemm_variant(), "lazy": self.arena.lazy is not None}
        if (st["graph"] is False or st["ids"].shape != ids.shape or st["mb"].shape != maskbias.shape
                or st["variant"] != ops.gemm_variant() or st.get("lazy", False) != (self.arena.lazy is not None)
                or torch.cuda.is_current_stream_capturing()):
            return self._encoder_forward(ids, pos_ids, maskbias, B, S, False)
        st["ids"].copy_(ids)
        st["pos"].copy_(pos_ids)
        st["mb"].copy_(maskbias)
        if st["graph"] is None:
            st["calls"] += 1
            if st["calls"] < 3:   # two eager passes first: every kernel's one-time attribute call and workspace allocation is behind us
                return self._encoder_forward(st["ids"], st["pos"], st["mb"], B, S, False)
            if dp_world_size() > 1 and not self.INFER_GRAPH_DP:
                # a data-parallel rank has a live RCCL watchdog thread and collectives of other ranks' making in flight; the
                # capture has not been exercised there (1-GPU boxes), so sharded evaluation keeps the eager launches
                st["graph"] = False
                return self._encoder_forward(st["ids"], st["pos"], st["mb"], B, S, False)
            g = torch.cuda.CUDAGraph()
            prev = L_._stream_cached
            try:
                # thread_local: another thread's HIP call (a watchdog, a loader) must not invalidate this capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    L_._stream_cached = L_.c_void_p(torch.cuda.current_stream().cuda_stream)   # the capture stream
                    st["out"] = self._encoder_forward(st["ids"], st["pos"], st["mb"], B, S, False)
            except Exception as e:   # a failed capture disables the graph for this shape ONCE; the pass itself runs eagerly
                L_._stream_cached = prev
                st["graph"] = False
                st.pop("out", None)
                log.warning("HIP-graph capture of the forward-only encoder pass failed for shape (%d, %d): %s -- this shape "
                            "keeps eager launches", B, S, e)
                torch.cuda.synchronize()
                return self._encoder_forward(st["ids"], st["pos"], st["mb"], B, S, False)
            finally:
                L_._stream_cached = prev
            st["graph"] = g
        self._enc_saved = None
        st["graph"].replay()
        return st["out"]

    def _encoder_forward(self, ids, pos_ids, maskbias, B, S, need_grad=True):
        """ids/pos_ids i32[Mp], maskbias f32[B,S] -> last hidden state bf16 [Mp,H] (rows >= B*S are padding).
        need_grad=False (inference: evaluate, the frozen encoders of an embedding stack) runs the FFN-up epilogue without the
        gelu' output that only encoder_backward reads; a backward after such a forward fails loudly."""
        cfg, a = self.cfg, self.arena
        H, F_, A, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers
        ac = self.acts(B, S)
        Mp = ac.Mp
        eps = cfg.layer_norm_eps
        d_emb, d_layers = self._site_seeds()
        a.catch_up_rows(ids)   # FusedAdamW.lazy_rows: the looked-up rows owe the zero-gradient steps since they were last touched
        ops.embed_ln_fwd(ids, pos_ids, a.param("emb.word"), a.param("emb.pos"), a.param("emb.type")[0], a.param("emb.ln.g"),
                         a.param("emb.ln.b"), eps, ac.h0, ac.x[0], ac.emb_mean, ac.emb_rstd, drop=d_emb)
        for l in range(L):
            p = "l%d." % l
            x = ac.x[l]
            d_att, d_o, d_f = d_layers[l]
            ops.gemm(GEMM_NT, x, a.bf(p + "qkv.weight"), Mp, 3 * H, H, C=ac.qkv[l], bias=a.param(p + "qkv.bias"), epi=EPI_BIAS, occupancy=True)
            if need_grad and self.ATTN_RESIDUAL and ac.ctx_lo[l] is None:
                ac.ctx_lo[l] = torch.empty((Mp * H,), dtype=torch.uint8, device=self.device)   # what the backward reads is written
            ops.attn_fwd(ac.qkv[l], maskbias, ac.ctx[l], ac.lse[l], B, S, H, A, drop=d_att,
                         ctx_lo=ac.ctx_lo[l] if need_grad and self.ATTN_RESIDUAL else None)
            ops.gemm(GEMM_NT, ac.ctx[l], a.bf(p + "o.weight"), Mp, H, H, C=ac.h1[l], bias=a.param(p + "o.bias"), addend=x,
                     epi=EPI_BIAS | EPI_ADD, drop=d_o, occupancy=True)
            ops.ln_fwd(ac.h1[l], a.param(p + "ln1.g"), a.param(p + "ln1.b"), eps, ac.x1[l], ac.st1[l][0], ac.st1[l][1])
            if need_grad:
                ops.gemm(GEMM_NT, ac.x1[l], a.bf(p + "ffn1.weight"), Mp, F_, H, C=ac.act[l], out2=ac.dact[l],
                         bias=a.param(p + "ffn1.bias"), epi=EPI_BIAS | EPI_GELU, occupancy=True)
            else:
                ops.gemm(GEMM_NT, ac.x1[l], a.bf(p + "ffn1.weight"), Mp, F_, H, C=ac.act[l],
                         bias=a.param(p + "ffn1.bias"), epi=EPI_BIAS | EPI_GELU_FWD, occupancy=True)
            slabs = self._long_k_gemm(GEMM_NT, ac.act[l], a.bf(p + "ffn2.weight"), Mp, H, F_, ac.h2[l], ac, bias=a.param(p + "ffn2.bias"),
                                      addend=ac.x1[l], drop=d_f, finish=False)
            if slabs is not None:   # small micro-batches: h2 = fold of the split-K slabs, written by the LayerNorm kernel itself
                ops.ln_fwd_slabs(slabs[0], slabs[1], a.param(p + "ffn2.bias"), ac.x1[l], d_f, ac.h2[l], a.param(p + "ln2.g"),
                                 a.param(p + "ln2.b"), eps, ac.x[l + 1], ac.st2[l][0], ac.st2[l][1])
            else:
                ops.ln_fwd(ac.h2[l], a.param(p + "ln2.g"), a.param(p + "ln2.b"), eps, ac.x[l + 1], ac.st2[l][0], ac.st2[l][1])
        self._enc_saved = (ids, pos_ids, maskbias, B, S, d_emb, d_layers) if need_grad else None
        return ac.x[L]

    def encoder_backward(self, dx_top, grad_ready=None):
        """dx_top bf16 [Mp,H] = d loss / d last hidden state; accumulates into arena.g.
        grad_ready(lo, hi): called right after the launches that finalise arena.g[lo:hi] were enqueued -- the GEMM-weight
        gradients of a whole WGRAD_GROUP of layers (contiguous in the arena) -- so a data-parallel trainer can start that
        bucket's all-reduce while backward continues below (kbner.dp.GradReducer)."""
        cfg, a = self.cfg, self.arena
        H, F_, A, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers
        if self._enc_saved is None:
            raise RuntimeError("encoder_backward needs an encoder_forward(need_grad=True) before it (the last forward kept no gelu')")
        ids, pos_ids, maskbias, B, S, d_emb, d_layers = self._enc_saved
        ac = self.acts(B, S)
        Mp = ac.Mp
        dx = dx_top
        pending = []
        dropping = any(d[1][1] for d in d_layers)
        dhm_ring, dh1m_ring = ac.drop_buffers() if dropping else (ac.dh, ac.dh1)
        # DEFER_REDUCE_MAX_TOKENS: below it every LayerNorm backward keeps its partial column sums in a workspace of its own and the
        # FFN-up bias gradients their epilogue lines; both are reduced for the whole pass at its end (ops.ln_colreduce_batched /
        # colsum_rows_f32_batched) instead of by 3 L launches of ~4.7 us
        defer = Mp <= self.DEFER_REDUCE_MAX_TOKENS and ops.HBM_HOOK is None
        ln_items, cs_items, defer_ln, ln_blocks = [], [], None, 0
        dx_slabs = None     # (split-K slabs, slices, residual) standing in for `dx` when the producing GEMM left its fold to LN2 backward
        if defer:
            ln_blocks = ops.ln_bwd_blocks(Mp)
            if ac.defer_ln_ws is None:
                ac.defer_ln_ws = torch.empty((2 * L + 1, ln_blocks * 3 * H), dtype=F32, device=self.device)
                ac.colsum_ws_all = torch.empty((L, 2 * (Mp // 128), F_), dtype=F32, device=self.device)
            defer_ln = ac.defer_ln_ws
        for l in range(L - 1, -1, -1):
            p = "l%d." % l
            r = l % WGRAD_GROUP
            dh, dh1, dpre, dqkv = ac.dh[r], ac.dh1[r], ac.dpre[r], ac.dqkv[r]
            d_att, d_o, d_f = d_layers[l]
            # with hidden dropout the sub-layer GEMMs see dY = mask * dh (dhm / dh1m); the residual branch keeps dh / dh1
            dhm = dhm_ring[r] if d_f[1] else dh
            dh1m = dh1m_ring[r] if d_o[1] else dh1
            # LN2 backward; fused: d ffn2.bias = column sums of dh
            ops.ln_bwd(dx, ac.h2[l], ac.st2[l][0], ac.st2[l][1], a.param(p + "ln2.g"), dh, a.grad(p + "ln2.g"),
                       a.grad(p + "ln2.b"), a.grad(p + "ffn2.bias"), dhm=dhm if d_f[1] else None, drop=d_f,
                       defer_ws=defer_ln[2 * l + 1] if defer else None, dy_slabs=dx_slabs)
            dx_slabs = None
            if defer:
                ln_items.append((defer_ln[2 * l + 1], a.grad(p + "ln2.g"), a.grad(p + "ln2.b"), a.grad(p + "ffn2.bias"), ln_blocks))
            # FFN down dgrad: dpre = (dh W2) * gelu'(pre)   (the derivative itself was saved by the forward epilogue)
            # (its column sums = d ffn1.bias are accumulated by the same epilogue when the 256^2 kernel runs)
            fused = ops.uses_256(Mp, F_, occupancy=True)
            if fused:
                # column sums (= d ffn1.bias) leave the epilogue as plain stores into a [2 * Mp/256, F] workspace, folded by a
                # small reduce kernel: per-tile atomics onto the same 4096 addresses were what made this the slowest GEMM
                if ac.colsum_ws is None:
                    ac.colsum_ws = torch.empty((2 * (Mp // 128), F_), dtype=F32, device=self.device)
                # ops.gemm reports the tile height of the kernel that ran (256 under dynamic tile draw, the static kernel's pick --
                # also when the scheduler ring is used up -- otherwise): it decides how many workspace lines hold partial sums
                cws = ac.colsum_ws_all[l] if defer else ac.colsum_ws
                trows = ops.gemm(GEMM_NN, dhm, a.bf(p + "ffn2.weight"), Mp, F_, H, C=dpre, aux=ac.dact[l],
                                 epi=EPI_DGELU | EPI_COLSUM | EPI_COLSUM_WS, colsum=cws, occupancy=True)
                if defer and 2 * (Mp // trows) <= 64:
                    cs_items.append((cws, a.grad(p + "ffn1.bias"), 2 * (Mp // trows)))
                else:
                    ops.colsum_rows_f32(cws, 2 * (Mp // trows), a.grad(p + "ffn1.bias"))
            else:
                ops.gemm(GEMM_NN, dhm, a.bf(p + "ffn2.weight"), Mp, F_, H, C=dpre, aux=ac.dact[l], epi=EPI_DGELU, occupancy=True)
                ops.colsum(dpre, a.grad(p + "ffn1.bias"))
            dx1_slabs = self._long_k_gemm(GEMM_NN, dpre, a.bf(p + "ffn1.weight"), Mp, H, F_, ac.dx1, ac, addend=dh, finish=False)
            if dx1_slabs is not None:
                dx1_slabs = (dx1_slabs[0], dx1_slabs[1], dh)
            # LN1 backward; fused: d o.bias
            ops.ln_bwd(ac.dx1, ac.h1[l], ac.st1[l][0], ac.st1[l][1], a.param(p + "ln1.g"), dh1, a.grad(p + "ln1.g"),
                       a.grad(p + "ln1.b"), a.grad(p + "o.bias"), dhm=dh1m if d_o[1] else None, drop=d_o,
                       defer_ws=defer_ln[2 * l] if defer else None, dy_slabs=dx1_slabs)
            if defer:
                ln_items.append((defer_ln[2 * l], a.grad(p + "ln1.g"), a.grad(p + "ln1.b"), a.grad(p + "o.bias"), ln_blocks))
            # attention output projection
            ops.gemm(GEMM_NN, dh1m, a.bf(p + "o.weight"), Mp, H, H, C=ac.dctx, occupancy=True)
            # attention core (+ d qkv.bias = column sums of dqkv, accumulated inside the kernels)
            ops.attn_bwd(ac.qkv[l], ac.ctx[l], ac.dctx, maskbias, ac.lse[l], ac.dws, dqkv, B, S, H, A, drop=d_att,
                         dbias=a.grad(p + "qkv.bias"), ctx_lo=ac.ctx_lo[l] if self.ATTN_RESIDUAL else None)
            # QKV projection
            # (the next layer's LN2 backward folds this GEMM's slabs itself -- the first launch of that layer, before anything
            #  else writes the slab buffer; layer 0's gradient goes to the embedding LayerNorm as bf16)
            dx_slabs = self._long_k_gemm(GEMM_NN, dqkv, a.bf(p + "qkv.weight"), Mp, H, 3 * H, ac.dx, ac, addend=dh1, finish=l == 0)
            if dx_slabs is not None:
                dx_slabs = (dx_slabs[0], dx_slabs[1], dh1)
            # weight gradients dW += dY^T X are deferred and launched for WGRAD_GROUP layers at once
            # (no split-K, no atomics; the dY buffers rotate so they stay live until the group is flushed)
            pending += [(dhm, ac.act[l], H, F_, p + "ffn2.weight"), (dpre, ac.x1[l], F_, H, p + "ffn1.weight"),
                        (dh1m, ac.ctx[l], H, H, p + "o.weight"), (dqkv, ac.x[l], 3 * H, H, p + "qkv.weight")]
            if l % WGRAD_GROUP == 0:
                self._wgrads(pending, Mp)
                pending = []
                if grad_ready is not None:
                    hi_l = min(l + WGRAD_GROUP, L)
                    lo_off = a.offsets["l%d.qkv.weight" % l]
                    hi_off = a.offsets["l%d.qkv.weight" % hi_l] if hi_l < L else a.offsets["emb.word"]
                    grad_ready(lo_off, hi_off)
                    ops.sched_active(True)   # a collective is in flight from here to the end of this backward pass
            dx = ac.dx
        a.wgrad_stale = False   # every GEMM-weight gradient has been written by this pass
        # small batches: the column-sum reductions of the pass (2 L LayerNorm partials -> gamma / beta / bias gradients, L FFN-up bias
        # workspaces) in two launches instead of 3 L
        # (the embedding LayerNorm's kernel also sets the optimizer's row flags of the rows it writes, and its partial sums join the batch)
        ops.embed_ln_bwd(dx, ac.h0, ac.emb_mean, ac.emb_rstd, a.param("emb.ln.g"), ids, pos_ids, a.grad("emb.ln.g"),
                         a.grad("emb.ln.b"), a.grad("emb.word"), a.grad("emb.pos"), a.grad("emb.type")[0], drop=d_emb,
                         row_flags=a.emb_flags, defer_ws=defer_ln[2 * L] if defer else None)
        if defer:
            ln_items.append((defer_ln[2 * L], a.grad("emb.ln.g"), a.grad("emb.ln.b"), a.grad("emb.type")[0], ln_blocks))
        for k in range(0, len(ln_items), 64):
            ops.ln_colreduce_batched(ln_items[k:k + 64], H)
        for k in range(0, len(cs_items), 64):
            ops.colsum_rows_f32_batched(cs_items[k:k + 64], F_)

    def _wgrads(self, pairs, Mp):
        a = self.arena
        if all(n_ % 256 == 0 and k_ % 256 == 0 for _, _, n_, k_, _ in pairs):
            # the first backward pass after an optimizer step overwrites the stale gradients (Arena.wgrad_stale), later ones add
            epi = EPI_STORE32 if (a.wgrad_stale and a.wgrad_overwrite_ok) else EPI_RMW32
            ops.gemm_grouped(GEMM_TN, [ops.make_problem(dy, x, n_, k_, Mp, C32=a.grad(nm), epi=epi)
                                       for dy, x, n_, k_, nm in pairs])
        else:
            for dy, x, n_, k_, nm in pairs:
                ops.gemm(GEMM_TN, dy, x, n_, k_, Mp, C32=a.grad(nm), epi=EPI_ATOMIC32,
                         splitk=_splitk((n_ // 128) * (k_ // 128), Mp))

    # ---------------------------------------------------------------- tagger head
    def emissions(self, hidden, row_idx, B, n):
        """gather rows (first sub-token of each word token, -1 -> zeros) then the linear head.
        -> (features f32 [B,n,T], pooled bf16 [B*n,H])"""
        pooled = ops.gather_rows(hidden, row_idx)
        em = ops.head_fwd(pooled, self.arena.param("linear.weight"), self.arena.param("linear.bias"))
        return em.view(B, n, self.T), pooled

    def forward_loss(self, batch, loss_scale=1.0, backward=True, weights=None, grad_ready=None):
        """One micro-batch: encoder -> kept-token gather -> head -> CRF NLL (mean over sentences,
        sequence_tagger_model.py:2499-2506) and, if `backward`, the full backward pass accumulating
        loss_scale * d loss into arena.g.  Returns the loss as a 0-d device tensor (no host sync).
        weights: optional f32[B] per-sentence weights replacing the 1/B of the mean (the trainer uses it to run the
        micro-batches of one gradient-accumulation group as ONE batch with weights 1/(accumulate * |micro-batch|))."""
        return self._launch(self._forward_loss, batch, loss_scale, backward, weights, grad_ready)

    def _launch(self, fn, *args):
        """run one micro-batch's launches on the current stream, with a fresh tile-scheduling ring when the GEMMs draw their
        tiles dynamically (data-parallel steps)"""
        with L_.stream_scope():
            if not self.dynamic_tiles:
                return fn(*args)
            if self._sched_ring is None:
                self._sched_ring = torch.zeros((512, 8), dtype=I32, device=self.device)
            self._sched_ring.zero_()     # one memset per micro-batch covers its ~200 GEMM launches
            # dynamic_tiles True: from the first gradient bucket's all-reduce on (encoder_backward); "always": every launch (A/B)
            ops.sched_ring_reset(self._sched_ring, active=(self.dynamic_tiles == "always"))
            try:
                return fn(*args)
            finally:
                ops.sched_ring_reset(None)

    def _emit(self, batch):
        """encoder -> (WordDropout) kept-token gather -> head: (em f32[B,nc,T], pooled bf16[B*nc,H], crow_idx, B, nc, R, S)"""
        B, S = batch["B"], batch["S"]
        R = batch.get("R", B)  # encoder rows (> B when long sentences were split into sliding windows)
        hidden = self.encoder_forward(batch["ids"], batch["pos_ids"], batch["maskbias"], R, S)
        nc = batch["ctags"].shape[1]
        crow_idx = batch["crow_idx"]
        if self.training and self.word_dropout > 0.0 and "cpos" in batch:
            # WordDropout (flair/nn.py:176-183) on the [n,B,D] sentence tensor: token position k is zeroed for EVERY sentence
            # of the batch, survivors are not rescaled.  Zero row = gather index -1 (and no gradient scattered back).
            n_pos = int(batch["n_tokens"])
            self._last_word_dropped = self._drop_rng.random(n_pos) < self.word_dropout
            dropped = torch.from_numpy(self._last_word_dropped).to(self.device)
            cpos = batch["cpos"].long()
            crow_idx = torch.where(dropped[cpos.clamp(min=0)] & (cpos >= 0), torch.full_like(crow_idx, -1), crow_idx)
        em, pooled = self.emissions(hidden, crow_idx, B, nc)
        return em, pooled, crow_idx, B, nc, R, S

    def _backprop_emissions(self, demit, pooled, crow_idx, B, nc, R, S, grad_ready, l2=None, l2_scale=1.0):
        """d loss / d emissions f32[B,nc,T] -> head, scatter to the encoder rows, encoder backward (all into arena.g).
        l2 = (other view's pooled rows bf16[B*nc,H], row weights f32[B*nc]): the representation term of multi-view training is
        added to the pooled-row gradient here; its value is returned (0-d tensor), else None."""
        a = self.arena
        dpooled = ops.head_bwd(demit.view(B * nc, self.T), pooled, a.param("linear.weight"), a.grad("linear.weight"),
                               a.grad("linear.bias"))
        l2_val = None
        if l2 is not None:
            part = torch.zeros((1,), dtype=F32, device=self.device)
            ops.l2_rows(pooled, l2[0], l2[1], part, da=dpooled, gscale=float(l2_scale))
            l2_val = part[0]
        ac = self.acts(R, S)
        ac.dx.zero_()
        ops.scatter_rows(dpooled, crow_idx, ac.dx)
        self.encoder_backward(ac.dx, grad_ready)
        return l2_val

    def _sentence_weights(self, weights, B):
        if weights is None:   # (read-only constants: one tensor per batch size, not one fill launch per micro-batch)
            cache = self.__dict__.setdefault("_mean_weights", {})
            w = cache.get(B)
            if w is None:
                w = cache[B] = torch.full((B,), 1.0 / B, dtype=F32, device=self.device)
            return w
        w = torch.as_tensor(weights, dtype=F32, device=self.device).contiguous()
        if w.numel() != B:
            raise ValueError("weights must hold one value per sentence")
        return w

    def _forward_loss(self, batch, loss_scale, backward, weights, grad_ready=None):
        em, pooled, crow_idx, B, nc, R, S = self._emit(batch)
        self.last_emissions = em   # f32 [B, nc, T] at the kept (non-S-X) tokens: the teacher view of multi-view training
        self.last_pooled = pooled.view(B, nc, -1)   # bf16 [B, nc, H]: the same view's token representations (calculate_l2_loss)
        a = self.arena
        trans = a.param("transitions")
        w = self._sentence_weights(weights, B)
        loss = torch.empty((1,), dtype=F32, device=self.device)
        if not getattr(self, "use_crf", True):
            # softmax head (FastSequenceTagger(use_crf=False), sequence_tagger_model.py:2523-2539): token-level cross entropy at the
            # kept tokens, sum_b w[b] * (sum over the sentence's tokens); the caller's weights carry the / B or / token-count
            per, demit = ops.softmax_ce(em, batch["ctags"], batch["clens"], w * loss_scale)
            ops.wdiff_sum(per, torch.zeros_like(per), w, loss)
            if backward:
                self._backprop_emissions(demit, pooled, crow_idx, B, nc, R, S, grad_ready)
            return loss[0]
        logz, gold, alpha = ops.crf_nll_fwd(em, trans, batch["ctags"], batch["clens"], self.start, self.stop)
        ops.wdiff_sum(logz, gold, w, loss)
        if backward:
            if weights is None:    # the constant 1 / B weights: their scaled copy is a constant per (B, loss_scale) too
                cache = self.__dict__.setdefault("_mean_weights_scaled", {})
                dl = cache.get((B, float(loss_scale)))
                if dl is None:
                    if len(cache) > 64:
                        cache.clear()
                    dl = cache[(B, float(loss_scale))] = w * loss_scale
            else:
                dl = w * loss_scale
            demit = ops.crf_nll_bwd(em, trans, batch["ctags"], batch["clens"], alpha, logz, dl, self.start, self.stop,
                                    a.grad("transitions"))
            self._backprop_emissions(demit, pooled, crow_idx, B, nc, R, S, grad_ready)
        return loss[0]

    def distill_loss(self, batch, teacher_emissions, tau, loss_scale=1.0, backward=True, weights=None, grad_ready=None,
                     mode="posterior", teacher_pooled=None, l2_only=False):
        """Multi-view distillation term (FastSequenceTagger._calculate_multi_view_loss, sequence_tagger_model.py:1958-2107): `batch`
        is the STUDENT view (the bare sentences), teacher_emissions f32[B,n_t,T] the emissions of the same sentences' real tokens in
        the context view (a constant: the reference detaches it).
          mode "posterior" (:2088-2101)  sum_b weights[b] * T^2 * sum_i KL(teacher || student tempered token marginals)
          mode "exact"     (:2049-2087)  sum_b weights[b] * max(0, -(E_teacher[path score / T] - logZ_T) T^2), the teacher being the
                                         context view's tempered pairwise posteriors under the shared transitions
          teacher_pooled bf16[B,n_t,H]   calculate_l2_loss (:1988-1996,2026-2035): + sum_b weights[b] / H * |student token
                                         representations - the context view's|^2 at the real tokens; l2_only (:2038): only that
        (default weights 1/B = the reference's `.sum() / shape[0]`).  Returns a 0-d device tensor and, if `backward`, accumulates
        loss_scale * its gradient into arena.g."""
        return self._launch(self._distill_loss, batch, teacher_emissions, tau, loss_scale, backward, weights, grad_ready, mode,
                            teacher_pooled, l2_only)

    def distill_terms(self, em, te, clens, w, tau, mode, loss_scale, dtrans):
        """the CRF part of the multi-view term from the two views' emissions on (f32 [B,nc,T] each; te constant): -> (sum_b w[b] *
        loss_b as a 0-d tensor, d(loss_scale * that) / d em); the transition gradient is ADDED to dtrans"""
        trans = self.arena.param("transitions")
        B = em.shape[0]
        if mode == "none":
            return torch.zeros((), dtype=F32, device=self.device), torch.zeros_like(em)
        if mode == "exact":
            pair, _, _ = ops.crf_pair_posterior(te, trans, clens, tau, self.start, self.stop)
            s_sc = ((te[:, 0, :] + trans[:, self.start][None, :]) / tau).contiguous()          # :2065 (no backward variable)
            e_sc = (trans[self.stop, :] / tau)[None, :].expand(B, self.T).contiguous()         # :2066 (no forward variable)
            per, demit = ops.crf_exact_kd(em, trans, clens, pair, s_sc, e_sc, w * loss_scale, tau, self.start, self.stop, dtrans)
        elif mode == "posterior":
            per, demit = ops.crf_posterior_kl(em, te, trans, clens, w * loss_scale, tau, self.start, self.stop, dtrans)
        else:
            raise ValueError("distill_loss mode must be 'posterior' or 'exact'")
        return (per * w).sum(), demit

    def _distill_loss(self, batch, te, tau, loss_scale, backward, weights, grad_ready, mode="posterior", tp=None, l2_only=False):
        em, pooled, crow_idx, B, nc, R, S = self._emit(batch)
        if te.shape[0] != B or te.shape[2] != self.T:
            raise ValueError("teacher emissions must be [B, n, T] for the same B sentences")
        if te.shape[1] < nc:   # (a real-token count mismatch between the views is a data error; the kernel only reads < lens)
            raise ValueError("the teacher view has fewer real tokens (%d) than the student view (%d)" % (te.shape[1], nc))
        te = te[:, :nc].contiguous()
        a = self.arena
        trans = a.param("transitions")
        w = self._sentence_weights(weights, B)
        dtr = a.grad("transitions") if backward else torch.zeros((self.T, self.T), dtype=F32, device=self.device)
        if l2_only and tp is None:
            raise ValueError("l2_only needs the teacher view's token representations")
        loss, demit = self.distill_terms(em, te, batch["clens"], w, tau, "none" if l2_only else mode, loss_scale, dtr)
        l2 = None
        if tp is not None:
            H = pooled.shape[1]
            if tp.shape[0] != B or tp.shape[1] < nc or tp.shape[2] != H:
                raise ValueError("teacher token representations must be [B, >= %d, %d]" % (nc, H))
            valid = torch.arange(nc, device=self.device)[None, :] < batch["clens"][:, None]
            wrow = (valid * (w / H)[:, None]).reshape(B * nc).to(F32).contiguous()
            l2 = (tp[:, :nc].contiguous().view(B * nc, H), wrow)
            if not backward:
                part = torch.zeros((1,), dtype=F32, device=self.device)
                ops.l2_rows(pooled, l2[0], wrow, part)
                loss = loss + part[0]
        if backward:
            part = self._backprop_emissions(demit, pooled, crow_idx, B, nc, R, S, grad_ready, l2=l2, l2_scale=loss_scale)
            if part is not None:
                loss = loss + part
        return loss

    # ---------------------------------------------------------------- teacher-student knowledge distillation
    def kd_loss(self, batch, kd, interpolation, tau, loss_scale=1.0, backward=True, weights=None, grad_ready=None):
        """One micro-batch of `distill_mode` training (FastSequenceTagger.simple_forward_distillation_loss,
        sequence_tagger_model.py:2110-2372, for a CRF student):
            interpolation * (posterior + crf + exact) + (1 - interpolation) * NLL
        The KD terms are defined on the emissions of ALL word tokens (`features = self.forward(data_points)`, mask = the length
        mask), the NLL on the tokens left after the remove_x compaction (`_calculate_loss`) -- one encoder pass serves both.
        kd: the batch's teacher targets as device tensors, any of
            "scores"   list (one per teacher) of f32[B,n,T] teacher forward-backward scores          (distill_posterior)
            "targets"  i32[B,n,K] teacher n-best tag sequences, K = best_k * teachers                 (distill_crf)
            "weights"  f32[B,K] their path weights + "att_nums" (sentence, teacher) pairs             (crf_attention)
            "exact"    (pair f32[B,n-1,T*T], start_score f32[B,T], end_score f32[B,T])                (distill_exact)
            "emission" (teacher f32[B,n,T], rows are probabilities: bool)                             (distill_emission / _prob)
        weights: optional per-sentence weights replacing the 1/B of the batch means.  Returns the loss (0-d device tensor);
        self.last_kd_parts = (kd terms, NLL), both unweighted by the interpolation."""
        return self._launch(self._kd_loss, batch, kd, interpolation, tau, loss_scale, backward, weights, grad_ready)

    def _kd_loss(self, batch, kd, interpolation, tau, loss_scale, backward, weights, grad_ready):
        B, S = batch["B"], batch["S"]
        R = batch.get("R", B)
        hidden = self.encoder_forward(batch["ids"], batch["pos_ids"], batch["maskbias"], R, S)
        n = batch["row_idx"].numel() // B
        row_idx = batch["row_idx"]
        if self.training and self.word_dropout > 0.0:
            # WordDropout (flair/nn.py:176-183): token position k zeroed for EVERY sentence of the batch (see _emit)
            self._last_word_dropped = self._drop_rng.random(n) < self.word_dropout
            dropped = torch.from_numpy(self._last_word_dropped).to(self.device).repeat(B)
            row_idx = torch.where(dropped, torch.full_like(row_idx, -1), row_idx)
        em, pooled = self.emissions(hidden, row_idx, B, n)
        self.last_emissions = em
        loss, demit = self.kd_crf_terms(em, batch, kd, interpolation, tau, loss_scale, backward, weights)
        if backward:
            self._backprop_emissions(demit, pooled, row_idx, B, n, R, S, grad_ready)
        return loss

    def kd_crf_terms(self, em, batch, kd, interpolation, tau, loss_scale=1.0, backward=True, weights=None, dtrans=None):
        """everything of kd_loss downstream of the all-token emissions em f32[B,n,T]: -> (loss 0-d tensor, d loss / d em or None).
        The transition gradient is accumulated into arena.g (or into `dtrans` when given: tests)."""
        a = self.arena
        B, n, T = em.shape
        trans = a.param("transitions")
        w = self._sentence_weights(weights, B)
        if dtrans is None:
            dtrans = a.grad("transitions") if backward else torch.zeros((T, T), dtype=F32, device=self.device)
        lens = batch["lengths"]
        ip = float(interpolation)
        kd_val = torch.zeros((), dtype=F32, device=self.device)
        demit = None

        def acc(d):
            nonlocal demit
            demit = d if demit is None else demit.add_(d)

        scores = kd.get("scores") or []
        for st in scores:                                                # :2120-2136, mean over the teachers
            wk = w * (ip / len(scores))
            per, d = ops.crf_posterior_kl_scores(em, st, trans, lens, wk * loss_scale, tau, self.start, self.stop, dtrans)
            kd_val = kd_val + (per * w).sum() / len(scores)
            acc(d)
        if kd.get("emission") is not None:                               # :2311-2365, 2384-2398: token-level KL on the emissions
            teach, is_prob = kd["emission"]
            per, d = ops.emission_kl(em, teach, lens, w * (ip * loss_scale), tau, is_prob)
            kd_val = kd_val + (per * w).sum()
            acc(d)
        if kd.get("exact") is not None:                                  # :2139-2244
            pair, s_sc, e_sc = kd["exact"]
            per, d = ops.crf_exact_kd(em, trans, lens, pair, s_sc, e_sc, w * (ip * loss_scale), tau, self.start, self.stop, dtrans)
            kd_val = kd_val + (per * w).sum()
            acc(d)
        if kd.get("targets") is not None:                                # :2249-2309: every teacher path as a gold sequence
            tg = kd["targets"]
            K = tg.shape[2]
            em_k = em.repeat_interleave(K, 0)                            # row b * K + k, as the reference's repeat / permute
            tags_k = tg.permute(0, 2, 1).reshape(B * K, n).contiguous()
            lens_k = lens.repeat_interleave(K)
            logz, gold, alpha = ops.crf_nll_fwd(em_k, trans, tags_k, lens_k, self.start, self.stop)
            if kd.get("weights") is not None:                            # crf_attention: sum(nll * att) / att_nums
                wk = (kd["weights"] * (w[:, None] * (B / float(kd["att_nums"])))).reshape(B * K).contiguous()
            else:                                                        # mean over the B * K paths
                wk = (w[:, None] / K).expand(B, K).reshape(B * K).contiguous()
            part = torch.empty((1,), dtype=F32, device=self.device)
            ops.wdiff_sum(logz, gold, wk, part)
            kd_val = kd_val + part[0]
            d = ops.crf_nll_bwd(em_k, trans, tags_k, lens_k, alpha, logz, wk * (ip * loss_scale), self.start, self.stop, dtrans)
            acc(d.view(B, K, n, T).sum(1))
        # gold-label NLL on the compacted rows (:2371, _calculate_loss)
        nc = batch["ctags"].shape[1]
        comp = ops.gather_rows_f32(em.view(B * n, T), batch["cfeat_idx"]).view(B, nc, T)
        logz, gold, alpha = ops.crf_nll_fwd(comp, trans, batch["ctags"], batch["clens"], self.start, self.stop)
        nll = torch.empty((1,), dtype=F32, device=self.device)
        ops.wdiff_sum(logz, gold, w, nll)
        self.last_kd_parts = (kd_val, nll[0])
        loss = ip * kd_val + (1.0 - ip) * nll[0]
        if not backward:
            return loss, None
        dc = ops.crf_nll_bwd(comp, trans, batch["ctags"], batch["clens"], alpha, logz, w * ((1.0 - ip) * loss_scale), self.start,
                             self.stop, dtrans)
        if demit is None:
            demit = torch.zeros_like(em)
        ops.scatter_add_rows_f32(dc.view(B * nc, T), batch["cfeat_idx"], demit.view(B * n, T))
        return loss, demit

    def forward_features(self, batch):
        """FastSequenceTagger.forward (sequence_tagger_model.py:844): emissions for ALL word tokens."""
        B, S = batch["B"], batch["S"]
        hidden = self.encoder_forward(batch["ids"], batch["pos_ids"], batch["maskbias"], batch.get("R", B), S, need_grad=False)
        n = batch["row_idx"].numel() // B
        em, _ = self.emissions(hidden, batch["row_idx"], B, n)
        return em

    def viterbi(self, emissions, lens):
        return ops.crf_viterbi(emissions.contiguous(), self.arena.param("transitions"), lens, self.start, self.stop)


class FusedAdamW:
    """HF AdamW (eps 1e-6, wd 0, correct_bias) + clip_grad_norm_(5.0) + linear decay, two lr groups
    (transitions at lr * lr_rate) -- flair/trainers/finetune_trainer.py:552-571,686-688,1010-1023."""

    def __init__(self, arena, lr=5e-6, lr_rate=10000.0, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, max_norm=5.0,
                 t_total=None, warmup=0):
        self.arena = arena
        self.lr, self.lr_rate, self.betas, self.eps, self.wd = lr, lr_rate, betas, eps, weight_decay
        self.max_norm, self.t_total, self.warmup = max_norm, t_total, warmup
        self.t = 0
        arena.ensure_state()
        from . import lib as L
        self.ws = torch.zeros(L.load().kbner_sqnorm_ws_floats(), dtype=F32, device=arena.device)
        self.norm_sq = torch.zeros(1, dtype=F32, device=arena.device)
        self.split = arena.offsets["transitions"]
        self.sparse_embedding = True   # skip word-embedding rows that never received a gradient (exact; see step())
        self._lazy_rows = False
        if getattr(arena, "lazy", None) is not None:
            # an earlier optimizer of this arena left its table lazy: bring every row up to date and retire that clock -- this
            # optimizer's (eager until lazy_rows is set) would otherwise step rows that still owe the old one
            arena.materialize_rows()
            arena.lazy["row_t"].fill_(-1)
            arena.lazy = None

    # lazy_rows (opt-in; the trainer and bench.py switch it on).  A live embedding row that receives no gradient in a step is
    # moved by an update that reads nothing but its own p / m / v: instead of streaming all live rows through HBM every step
    # (24 B per element: 1.0 ms of the YAML regime's 11-ms step once every row of XLM-R's table is live) the step updates the rows
    # that DID receive a gradient, and the others are brought up to date -- the same fp32 operations in the same order, in
    # registers -- when the encoder next looks them up (Arena.catch_up_rows) or a later step touches them.  Bit-identical to the
    # eager update; every reader of emb.word / its moments other than the encoder forward calls Arena.materialize_rows() first
    # (hf_state_dict, set_param, load_hf_state_dict, state_dict and load_state_dict here do).
    LAZY_HIST = 8192          # steps of step_size history (a power of two) ...
    LAZY_FULL_EVERY = 4096    # ... and every this many steps all rows are brought up to date: bounds a row's catch-up loop

    @property
    def lazy_rows(self):
        return self._lazy_rows

    @lazy_rows.setter
    def lazy_rows(self, on):
        a = self.arena
        on = bool(on) and a.emb_flags is not None and a.p.is_cuda and a.shapes["emb.word"][1] <= 1024
        if on == self._lazy_rows:
            return
        if on:
            dev = a.p.device
            z = a.__dict__.get("_lazy_bufs")
            if z is None:   # allocated once per arena: a captured forward pass (encoder_forward's HIP graph) holds these pointers
                z = a._lazy_bufs = {"row_t": torch.empty(a.emb_flags.numel(), dtype=torch.int32, device=dev),
                                    "clock": torch.empty(1, dtype=torch.int32, device=dev),
                                    "hist": torch.zeros(self.LAZY_HIST, dtype=F32, device=dev)}
            z["row_t"].copy_(torch.where(a.emb_flags != 0, self.t, -1).to(torch.int32))   # live rows are current, the others never moved
            z["clock"].fill_(self.t)
            z.update({"b1": self.betas[0], "b2": self.betas[1], "eps": self.eps, "dirty": False, "last_full": self.t})
            a.lazy = z
        else:
            a.materialize_rows()
            a.lazy["row_t"].fill_(-1)   # a captured catch-up launch that is replayed from now on finds nothing to do
            a.lazy = None
        self._lazy_rows = on

    def lazy_rows_for(self, tokens_per_step):
        """lazy_rows where it pays: when a step looks up a small part of the table (at most an eighth of its rows -- the YAMLs' 4
        sentences: 0.8 %).  With 256 sentences per step 41 % of the rows are visited, each owing 1.4 steps: the per-row catch-up
        launches then cost what the eager update's extra bytes do (255.9 against 255.3 ms per step)."""
        a = self.arena
        self.lazy_rows = a.emb_flags is not None and 8 * int(tokens_per_step) <= a.emb_flags.numel()
        return self.lazy_rows

    def materialize(self):
        """arena.p / m / v are the eager optimizer's after this (no-op unless lazy_rows)"""
        self.arena.materialize_rows()

    def state_dict(self):
        """what a resume needs (the reference stores optimizer.state_dict(), finetune_trainer.py:1261-1277): the step count and
        the two Adam moment arenas (host copies; 2 x 2.24 GB for XLM-R-large, like the reference's exp_avg / exp_avg_sq)"""
        a = self.arena
        a.materialize_rows()
        return {"t": self.t, "n": a.n, "m": a.m.detach().cpu(), "v": a.v.detach().cpu()}

    def load_state_dict(self, sd):
        a = self.arena
        if int(sd["n"]) != a.n:
            raise ValueError("optimizer state holds %d elements, the arena %d" % (int(sd["n"]), a.n))
        a.ensure_state()
        lazy = self._lazy_rows
        self.lazy_rows = False          # (the restored moments are every row's: the lazy clock restarts from them below)
        a.m.copy_(sd["m"].to(a.device))
        a.v.copy_(sd["v"].to(a.device))
        self.t = int(sd["t"])
        if a.emb_flags is not None:
            # a row is live iff it has ever received a gradient, i.e. iff one of its moments is not identically zero (v alone
            # can underflow to 0 for tiny gradients while m is still non-zero: the dense update would then apply m / eps)
            lo = a.offsets["emb.word"]
            V, H = a.shapes["emb.word"]
            live = (a.v[lo:lo + V * H].view(V, H) != 0).any(1) | (a.m[lo:lo + V * H].view(V, H) != 0).any(1)
            # (LIVE | TOUCHED: whatever arena.g holds for these rows at this point is read by the next step)
            a.emb_flags.copy_(live.to(torch.uint8) * 3)
        if lazy:
            self.lazy_rows = True

    def lr_lambda(self):
        if self.t_total is None:
            return 1.0
        if self.t < self.warmup:
            return float(self.t) / float(max(1, self.warmup))
        return max(0.0, float(self.t_total - self.t) / float(max(1, self.t_total - self.warmup)))

    def step(self, grad_scale=1.0):
        a = self.arena
        lam = self.lr_lambda()
        self.t += 1
        b1, b2 = self.betas
        bc = math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        s = self.split
        # the word-embedding table (46 % of XLM-R's parameters) is updated row-sparsely: rows that never received a gradient
        # have g = m = v = 0 and, with weight decay 0, HF AdamW leaves them where they are -- they are not read at all
        sparse = self.sparse_embedding and a.emb_flags is not None and self.wd == 0.0
        # GEMM-weight gradients: left in place for the next backward pass to overwrite (Arena.wgrad_overwrite_ok); if no backward
        # pass has written them since the last step they are that step's: a step without gradients sees zeros, as it always did
        keep = a.wgrad_overwrite_ok and a.n_shadow > 0
        a.finalize_grads()
        if sparse:
            e0 = a.offsets["emb.word"]
            V, H = a.shapes["emb.word"]
            e1 = e0 + V * H
            rows = lambda buf: buf[e0:e1].view(V, H)   # noqa: E731
            ops.grad_sqnorm(a.g[:e0], self.ws, self.norm_sq)
            ops.grad_sqnorm_rows(rows(a.g), a.emb_flags, self.ws, self.norm_sq, accumulate=True)
            ops.grad_sqnorm(a.g[e1:], self.ws, self.norm_sq, accumulate=True)
            ranges = [(0, e0, self.lr * lam), (e1, s, self.lr * lam), (s, a.n, self.lr * self.lr_rate * lam)]
        else:
            ops.grad_sqnorm(a.g, self.ws, self.norm_sq)
            ranges = [(0, s, self.lr * lam), (s, a.n, self.lr * self.lr_rate * lam)]
        if keep:   # the GEMM weights as a range of their own (they are the first n_shadow elements)
            ranges = [r for lo, hi, lr in ranges
                      for r in (((lo, a.n_shadow, lr), (a.n_shadow, hi, lr)) if lo < a.n_shadow < hi else ((lo, hi, lr),))]
        for lo, hi, lr in ranges:
            if hi <= lo:
                continue
            nsh = min(a.n_shadow, hi) - lo if lo < a.n_shadow else 0
            ops.adamw(a.p[lo:hi], a.g[lo:hi], a.m[lo:hi], a.v[lo:hi], a.shadow[lo:] if nsh > 0 else None, max(nsh, 0),
                      lr * bc, lr * self.wd, b1, b2, self.eps, self.norm_sq, self.max_norm, grad_scale,
                      not (keep and hi <= a.n_shadow))
        a.wgrad_stale = keep
        z = a.lazy if self._lazy_rows else None
        if z is not None and not sparse:      # (weight decay switched on / the sparse path switched off under a lazy table)
            self.lazy_rows = False
            z = None
        if sparse and z is not None:
            ops.adamw_rows_lazy(rows(a.p), rows(a.g), rows(a.m), rows(a.v), a.emb_flags, z["row_t"], z["clock"], z["hist"],
                                self.t, self.lr * lam * bc, b1, b2, self.eps, self.norm_sq, self.max_norm, grad_scale)
            z["dirty"] = True
            if self.t - z["last_full"] >= self.LAZY_FULL_EVERY:
                a.materialize_rows()
                z["last_full"] = self.t
        elif sparse:
            ops.adamw_rows(rows(a.p), rows(a.g), rows(a.m), rows(a.v), a.emb_flags, self.lr * lam * bc, b1, b2, self.eps,
                           self.norm_sq, self.max_norm, grad_scale, True)
        return self.norm_sq

