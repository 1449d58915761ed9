"""Data-parallel helpers: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in
CPU tests).  The reference has NO distributed training (single process, commented-out DataParallel at
flair/trainers/finetune_trainer.py:699-700); this is the new capability the north-star asks for.

Semantics pinned here:
  * sentences are independent units -> pure DP, replicas of the whole model, no data-path collective in fwd/bwd;
  * after `chunk_batches` + a rank-SHARED shuffle of the batch order, rank r takes micro-batches r, r+W, r+2W, ...; the tail
    is padded by wrapping around so every rank runs the same number of micro-batches / optimizer steps;
  * ONE sum-all-reduce of the flat fp32 gradient arena per optimizer step, issued after the last local micro-batch and before
    clip_grad_norm_, so the clip norm is taken on the averaged global gradient (= single-process semantics at W x the batch);
    the 1/W is folded into the AdamW kernel's grad_scale, the arena is never rescaled in a separate pass;
  * the LR schedule's t_total counts GLOBAL optimizer steps."""
import math
import os

import torch
import torch.distributed as dist

# the optimizer's embedding-row flags (include/kbner.h: KBNER_ROW_LIVE / KBNER_ROW_TOUCHED): a row whose gradient this exchange writes
# has received one (LIVE) and may hold a non-zero one now (TOUCHED)
ROW_LIVE, ROW_TOUCHED = 1, 2


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def init_from_env(backend=None):
    """torchrun-style init (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); no-op for a single process."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    if w <= 1 or is_dist():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group(backend)


def shard_indices(n_items, r=None, w=None):
    """indices of the micro-batches rank r runs: r, r+w, ... padded (wrap-around) to ceil(n/w) items on every rank"""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per = math.ceil(n_items / w) if n_items else 0
    return [(r + i * w) % n_items for i in range(per)]


def steps_per_epoch(n_batches, accum, w=None):
    """GLOBAL optimizer steps per epoch (what t_total multiplies by max_epochs, finetune_trainer.py:679)"""
    w = world_size() if w is None else w
    return math.ceil(math.ceil(n_batches / w) / accum)


def all_reduce_sum_(flat):
    """in-place sum over ranks of a flat tensor (the gradient arena); returns the 1/W factor AdamW must apply"""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def all_reduce_scalars(values, op="sum"):
    """small host-side statistics (loss sums, counters): returns a python list"""
    if world_size() == 1:
        return list(values)
    dev = "cuda" if (dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return t.cpu().tolist()


def broadcast_object(obj, src=0):
    if world_size() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def all_gather_object(obj):
    """every rank's (small, picklable) object, in rank order"""
    if world_size() == 1:
        return [obj]
    out = [None] * world_size()
    dist.all_gather_object(out, obj)
    return out


def barrier():
    if world_size() > 1:
        dist.barrier()


class _HipRowOps:
    """device-side pieces of the exchange, all in libkbner_hip.so (the CPU gloo test injects torch stand-ins instead)"""

    @staticmethod
    def gather_rows(src2d, idx):
        from . import ops
        return ops.gather_rows_f32(src2d, idx)

    @staticmethod
    def scatter_rows(rows, idx, dst2d):
        from . import ops
        ops.scatter_rows_f32(rows, idx, dst2d)

    @staticmethod
    def to_bf16(x):
        from . import ops
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        ops.f32_to_bf16(x, y)
        return y

    @staticmethod
    def from_bf16(y, out):
        from . import ops
        ops.bf16_to_f32(y, out)


class GradReducer:
    """The gradient exchange of ONE optimizer step, overlapped with backward.

    The gradient arena is laid out so that what backward finishes first is contiguous: the GEMM weights of layers
    [4g, 4g+4) become final when their grouped weight-gradient launch has been enqueued (engine.encoder_backward), i.e.
    6 x 201 MB buckets for XLM-R-large that complete one by one while backward still runs ~5/6 .. 0 of its length.

      * `bucket_ready(lo, hi)` -- called by the engine right after it enqueued the kernels that finalise arena.g[lo:hi]:
        issues `all_reduce(async_op=True)` on that slice.  ProcessGroupNCCL runs the collective on its own stream after an
        event on the current (compute) stream, so RCCL moves bucket g over xGMI while the GPU computes layers below it.
      * `finish()` -- after backward: exchanges what is left (embeddings, LayerNorm / bias vectors, head, transitions) and
        makes the compute stream wait for every outstanding bucket.  The word-embedding gradient ([V=250002, H], 46 % of
        all bytes) only becomes final at the very end of backward and cannot be hidden, but at most B*S of its rows are
        non-zero per rank: `begin(touched_ids)` lets the ranks agree on the union of touched rows on the HOST (gloo) before
        the forward pass, and when that union is below `sparse_threshold` of the vocabulary only those rows travel
        (gather -> all-reduce [U, H] -> scatter back); otherwise the dense slice is reduced (optionally as bf16:
        `compress_embedding=True`, a documented deviation from the fp32 mean gradient).
    The result is the SUM over ranks in arena.g; the 1/W is applied by the AdamW kernel (grad_scale).  With world_size 1
    everything is a no-op."""

    def __init__(self, arena_g, emb_range=None, emb_width=None, sparse_threshold=0.5, compress_embedding=False, row_ops=None,
                 emb_flags=None, coalesce=1, delay=0, time_buckets=False, finalize=None):
        self.g = arena_g
        # `finalize`: the gradient owner's "make g this step's gradients" hook (kbner.engine.Arena.finalize_grads: the GEMM-weight
        # gradients are not zeroed by the optimizer -- a rank whose step had NO backward pass would otherwise send the previous
        # step's).  Called before the first byte of a step travels.
        self.finalize = finalize
        self._finalized = False
        self.emb_flags = emb_flags          # u8[V] "row has received a gradient" flags of the optimizer (kbner.engine.Arena)
        self.emb_range = emb_range          # (lo, hi) element range of emb.word inside the arena, or None
        self.emb_width = emb_width
        self.sparse_threshold = sparse_threshold
        self.compress_embedding = compress_embedding
        self.ops = row_ops or _HipRowOps
        self.works, self.covered = [], []
        # A/B knobs for the first multi-GPU run (bench.py --bucket-layers / --exchange-delay): `coalesce` consecutive ready ranges
        # travel as ONE all-reduce (2 = 8-layer buckets of 402 MB instead of 4-layer ones); `delay` = a bucket is issued only when
        # `delay` later ones have become ready (its all-reduce then starts one grouped weight-gradient launch later: fewer CUs taken
        # from the layers right behind it); `time_buckets` = every bucket all-reduce is blocking and timed with events on the compute
        # stream (diagnostic step: the isolated duration of each collective, nothing overlapped)
        self.coalesce, self.delay, self.time_buckets = max(1, int(coalesce)), max(0, int(delay)), bool(time_buckets)
        self._pending, self._held = [], []
        self.bucket_events = []
        self.stats = {"buckets": 0, "bytes_overlapped": 0, "bytes_tail": 0, "emb_rows": None, "emb_mode": None}

    def _cpu_group(self):
        """host-side group for the touched-row ids (control plane): with the nccl backend a gloo group is created next to it,
        so the id exchange never synchronises the host with the GPU stream"""
        if getattr(self, "_cpu_pg", None) is None:
            self._cpu_pg = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD
        return self._cpu_pg

    def begin(self, touched_ids=None):
        """start of an optimizer step.  touched_ids: the word ids (host integers, any shape) this rank will look up in the
        step's micro-batches -- known before the forward pass, so the union over ranks is agreed on the host while the GPU
        works, and only its rows of the word-embedding gradient travel in finish()."""
        self.works, self.covered = [], []
        self._pending, self._held = [], []
        self.bucket_events = []
        self._finalized = False
        self.stats = {"buckets": 0, "bytes_overlapped": 0, "bytes_tail": 0, "emb_rows": None, "emb_mode": None}
        self._union = None
        if world_size() == 1 or self.emb_range is None or touched_ids is None or self.sparse_threshold <= 0:
            return
        import numpy as np
        V = (self.emb_range[1] - self.emb_range[0]) // self.emb_width
        mine = np.unique(np.asarray(touched_ids, dtype=np.int64).ravel())
        pg = self._cpu_group()
        cnt = torch.tensor([mine.size], dtype=torch.int64)
        cnts = [torch.zeros(1, dtype=torch.int64) for _ in range(world_size())]
        dist.all_gather(cnts, cnt, group=pg)
        cap = max(int(c) for c in cnts)
        if cap >= self.sparse_threshold * V:   # one rank alone already touches too much: dense exchange
            return
        pad = torch.full((cap,), -1, dtype=torch.int64)
        pad[:mine.size] = torch.from_numpy(mine)
        allp = [torch.empty_like(pad) for _ in range(world_size())]
        dist.all_gather(allp, pad, group=pg)
        union = np.unique(np.concatenate([t.numpy() for t in allp]))
        union = union[union >= 0]
        if union.size < self.sparse_threshold * V:
            self._union = torch.from_numpy(union.astype(np.int32)).to(self.g.device, non_blocking=True)

    def _issue(self, lo, hi):
        if self.time_buckets:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM)
            e1.record()
            self.bucket_events.append((4 * (hi - lo), e0, e1))
        else:
            self.works.append(dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.covered.append((lo, hi))
        self.stats["buckets"] += 1
        self.stats["bytes_overlapped"] += 4 * (hi - lo)

    def _flush(self, everything=False):
        """merge `coalesce` pending ranges into one bucket (contiguous ones; a gap starts a new bucket), hold `delay` of them back"""
        while self._pending and (everything or len(self._pending) >= self.coalesce):
            take, self._pending = self._pending[:self.coalesce], self._pending[self.coalesce:]
            take.sort()
            cur = list(take[0])
            for lo, hi in take[1:]:
                if lo == cur[1]:
                    cur[1] = hi
                else:
                    self._held.append(tuple(cur))
                    cur = [lo, hi]
            self._held.append(tuple(cur))
        while self._held and (everything or len(self._held) > self.delay):
            self._issue(*self._held.pop(0))

    def _finalize_once(self):
        if not self._finalized:
            self._finalized = True
            if self.finalize is not None:
                self.finalize()

    def bucket_ready(self, lo, hi):
        if world_size() == 1 or hi <= lo:
            return
        # (a bucket is announced by the backward pass that has just written it: the stale flag is already clear then, and
        # finalize is a no-op; it matters in finish() for a step that ran no backward pass at all)
        self._pending.append((lo, hi))
        self._flush()

    def _complement(self, skip):
        """ranges of [0, n) neither reduced by a bucket nor listed in `skip`"""
        done = sorted(self.covered + list(skip))
        out, cur = [], 0
        for lo, hi in done:
            if lo > cur:
                out.append((cur, lo))
            cur = max(cur, hi)
        if cur < self.g.numel():
            out.append((cur, self.g.numel()))
        return out

    def _exchange_embedding(self):
        lo, hi = self.emb_range
        H = self.emb_width
        V = (hi - lo) // H
        g2 = self.g[lo:lo + V * H].view(V, H)
        if self._union is not None:
            mode = "sparse"
            idx = self._union
            rows = self.ops.gather_rows(g2, idx)
            dist.all_reduce(rows, op=dist.ReduceOp.SUM)
            self.ops.scatter_rows(rows, idx, g2)
            if self.emb_flags is not None:      # rows other ranks touched now carry a gradient here too
                self.emb_flags[idx.long()] = ROW_LIVE | ROW_TOUCHED
            self.stats["emb_rows"] = int(idx.numel())
            self.stats["bytes_tail"] += 4 * H * int(idx.numel())
        elif self.compress_embedding:
            mode = "dense_bf16"
            half = self.ops.to_bf16(self.g[lo:hi])
            dist.all_reduce(half, op=dist.ReduceOp.SUM)
            self.ops.from_bf16(half, self.g[lo:hi])
            self.stats["bytes_tail"] += 2 * (hi - lo)
        else:
            mode = "dense"
            dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM)
            self.stats["bytes_tail"] += 4 * (hi - lo)
        if mode != "sparse" and self.emb_flags is not None:
            self.emb_flags.fill_(ROW_LIVE | ROW_TOUCHED)             # dense exchange: any row may have received a gradient from another rank
        self.stats["emb_mode"] = mode

    def finish(self):
        """after the step's last backward: -> the 1/W factor AdamW must apply"""
        w = world_size()
        if w == 1:
            return 1.0
        if not self.covered and not self._pending and not self._held:
            self._finalize_once()      # no backward pass announced anything: g may still hold the previous step's weight gradients
        self._flush(everything=True)   # buckets held back by `coalesce` / `delay`
        skip = []
        if self.emb_range is not None:
            self._exchange_embedding()
            skip.append(tuple(self.emb_range))
        elif self.emb_flags is not None:
            self.emb_flags.fill_(ROW_LIVE | ROW_TOUCHED)             # the whole arena is reduced densely: every row may carry a gradient now
        for lo, hi in self._complement(skip):
            dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM)
            self.stats["bytes_tail"] += 4 * (hi - lo)
        for wk in self.works:
            wk.wait()   # nccl: the compute stream waits on the collective's event (no host block); gloo: host wait
        self.works = []
        return 1.0 / w


def broadcast_params_(flat, src=0):
    """make every replica start from rank `src`'s parameters (the tagger head / transitions are drawn from each process's
    own torch seed otherwise -- ADVICE r1, high)"""
    if world_size() > 1:
        dist.broadcast(flat, src=src)
