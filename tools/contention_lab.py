#!/usr/bin/env python
"""What the dynamic tile draw is FOR, measured on one GPU: the training step while `k` CUs are held by
another kernel on a second stream -- the situation of a data-parallel step whose gradient buckets are in flight (RCCL's ring
kernels own a few dozen workgroups) -- with the GEMMs launched (a) static, ring loop (the N = 1 path), (b) static, two-stage loop,
(c) dynamic tile draw on the two-stage loop (rounds 2-4's N > 1 path), (d) the ring kernel with one workgroup per tile (round 5:
what N > 1 runs from the first bucket on -- the hardware dispatcher is the scheduler).
A persistent 256-workgroup GEMM launch whose grid does not fit next to the squatters runs its late workgroups' whole static share
after everybody else has finished; with the draw a late workgroup finds what is left.

  python tools/contention_lab.py [--micro-batch 128] [--steps 4] [--cus 0,16,32]
"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from kbner import batch as kb
from kbner import engine, ops

ap = argparse.ArgumentParser()
ap.add_argument("--micro-batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--cus", default="0,16,32")
a = ap.parse_args()
occ = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "liboccupy.so"))
occ.occupy_launch.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
T, start, stop, x_idx = 29, 27, 28, 9
cfg = engine.EncoderConfig(vocab_size=250002, max_position_embeddings=514)
tg = engine.Tagger(cfg, T, start, stop, device=dev)
tg.init_random(seed=kb.SEED)
tg.arena.emb_flags.fill_(1)
opt = engine.FusedAdamW(tg.arena, lr=5e-6, lr_rate=10000.0, t_total=1000)
mb = kb.to_device(kb.synthetic_batch(a.micro_batch, 512, vocab=cfg.vocab_size, T=T, x_idx=x_idx, start=start, stop=stop, seed=kb.SEED), dev)
side = torch.cuda.Stream()
sink = torch.zeros(4, dtype=torch.int32, device=dev)

def step():
    tg.forward_loss(mb, loss_scale=1.0, backward=True)
    opt.step()

def measure(k, cycles):
    times = []
    for i in range(a.steps + 1):
        torch.cuda.synchronize()
        if k:
            occ.occupy_launch(k, ctypes.c_ulonglong(cycles), ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record()
        torch.cuda.synchronize()
        if i:
            times.append(e0.elapsed_time(e1))
    return sum(times) / len(times)

modes = (("static ring (variant 3)", 3, False), ("static two-stage (variant 0)", 0, False),
         ("dynamic: tile draw on the two-stage loop, every launch (variant 0)", 0, "always"),
         ("dynamic: the ring kernel, one workgroup per tile (variant 3; what N > 1 runs)", 3, "always"))
step(); step(); torch.cuda.synchronize()
base = {}
for name, variant, dyn in modes:
    ops.gemm_variant(variant)
    tg.dynamic_tiles = dyn
    base[name] = measure(0, 0)
for k in [int(x) for x in a.cus.split(",")]:
    for name, variant, dyn in modes:
        ops.gemm_variant(variant)
        tg.dynamic_tiles = dyn
        # the squatters stay for 1.3 x the undisturbed step of this mode (2.1 GHz)
        ms = measure(k, int(base[name] * 1.3e-3 * 2.1e9)) if k else base[name]
        print(json.dumps({"cus_held": k, "gemm": name, "ms_per_step": round(ms, 2), "sentences_per_s": round(a.micro_batch / ms * 1e3, 1),
                          "vs_undisturbed": round(ms / base[name], 3)}), flush=True)
ops.gemm_variant(3)
