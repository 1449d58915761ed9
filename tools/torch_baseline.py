#!/usr/bin/env python
"""Measurement only (never the product path): the SAME training step -- XLM-R-large (random init, 24 x 1024 x 16 x 4096, V = 250 002),
512 sub-tokens per sentence, token-classification head, clip 5.0, AdamW -- through the software the reference is built on, stock
PyTorch + Hugging Face transformers on ROCm (bf16 autocast, fp32 master weights, SDPA attention, fused AdamW, dropout off), on the
same MI355X.  A generous stand-in for the reference's own 2020 stack (flair 0.4.3 + transformers 3.0.0, fp32, eager attention), which
does not install here: what a user gets today without this library.  The CRF is left out (a [B, n, 29] scan: < 1 % of the step).
    python tools/torch_baseline.py [--batch 32] [--steps 5]"""
import argparse, time
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--seq", type=int, default=512)
a = ap.parse_args()
from transformers import XLMRobertaConfig, XLMRobertaModel
cfg = XLMRobertaConfig(vocab_size=250002, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                       max_position_embeddings=514, type_vocab_size=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       layer_norm_eps=1e-5, pad_token_id=1)
try:
    cfg._attn_implementation = "sdpa"
except Exception:
    pass
torch.manual_seed(0)
dev = "cuda"
model = XLMRobertaModel(cfg, add_pooling_layer=False).to(dev)
head = torch.nn.Linear(1024, 29).to(dev)
params = list(model.parameters()) + list(head.parameters())
opt = torch.optim.AdamW(params, lr=5e-6, eps=1e-6, weight_decay=0.0, fused=True)
B, S = a.batch, a.seq
ids = torch.randint(5, 250002, (B, S), device=dev)
ids[:, 0] = 0; ids[:, -1] = 2
tags = torch.randint(0, 29, (B, S), device=dev)
mask = torch.ones(B, S, dtype=torch.long, device=dev)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h = model(input_ids=ids, attention_mask=mask).last_hidden_state
        logits = head(h)
    loss = torch.nn.functional.cross_entropy(logits.float().view(-1, 29), tags.view(-1))
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print("torch + transformers %s on ROCm: batch %d x %d: %.1f ms per step = %.1f sentences/s (peak memory %.1f GB)" % (
    __import__("transformers").__version__, B, S, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 1e9), flush=True)
