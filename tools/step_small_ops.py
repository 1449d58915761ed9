#!/usr/bin/env python
"""Which torch-native launches (fills, copies, elementwise) one 4-sentence training step makes, and from where (torch.profiler with
stacks): the launches of the step that are not kbner kernels.   python tools/step_small_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from kbner import batch as kb, engine
T, start, stop, x_idx = 29, 27, 28, 9
cfg = engine.EncoderConfig.large()
tg = engine.Tagger(cfg, T, start, stop, device="cuda"); tg.init_random(seed=1)
opt = engine.FusedAdamW(tg.arena, lr=5e-6, lr_rate=10000.0, t_total=1000)
tg.arena.emb_flags.fill_(1)
opt.lazy_rows = True
mb = kb.to_device(kb.synthetic_batch(4, 512, vocab=cfg.vocab_size, T=T, x_idx=x_idx, start=start, stop=stop, seed=1), "cuda")
for _ in range(3):
    tg.forward_loss(mb, backward=True); opt.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tg.forward_loss(mb, backward=True); opt.step()
    torch.cuda.synchronize()
import collections
agg = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::select", "aten::as_strided", "aten::view", "aten::empty", "aten::slice", "aten::empty_strided",
                                                         "aten::reshape", "aten::_unsafe_view", "aten::unsqueeze", "aten::expand", "aten::item", "aten::_local_scalar_dense"):
        st = [f for f in (ev.stack or []) if "kb-ner_amd" in f or "bench" in f]
        agg[(ev.name, tuple(s.split("kb-ner_amd/")[-1] for s in st[:2]), str(ev.input_shapes)[:60])] += 1
for (n, st, shp), c in sorted(agg.items(), key=lambda kv: -kv[1])[:40]:
    print(c, n, shp, " | ".join(st))
