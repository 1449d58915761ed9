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


def barrier():
    if world_size() > 1:
        dist.barrier()
