"""TEST INFRASTRUCTURE -- CPU restatement of the optimiser arithmetic on the fine-tune path.

Third-party: `transformers==3.0.0` `AdamW` and `get_linear_schedule_with_warmup`
(requirements.txt:30; constructed at flair/trainers/finetune_trainer.py:566-571,686-688), plus
`torch.nn.utils.clip_grad_norm_(params, 5.0)` (finetune_trainer.py:1010).  Not vendored under
/root/reference; this restates the published algorithm of that release:

  m = b1*m + (1-b1)*g ;  v = b2*v + (1-b2)*g*g
  step_size = lr * sqrt(1-b2^t) / (1-b1^t)          (correct_bias=True)
  p = p - step_size * m / (sqrt(v) + eps)           (eps = 1e-6)
  p = p - lr * wd * p                               (wd = 0 on this path)

Pinned by tests/golden/adamw.npz (generated in-container by oracle/gen_golden.py running the
reference trainer's optimiser construction against the shimmed AdamW, and cross-checked against
torch.optim.Adam with matching eps placement).  Only tests/smoke/bench cpu_baseline import this.
"""
import math

import numpy as np


def adamw_hf_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-6, wd=0.0, correct_bias=True):
    """In-place on float32 numpy arrays; `step` is the 1-based step count AFTER increment."""
    f = np.float32
    m *= f(b1)
    m += f(1.0 - b1) * g
    v *= f(b2)
    v += f(1.0 - b2) * g * g
    denom = np.sqrt(v) + f(eps)
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p -= f(step_size) * (m / denom)
    if wd > 0.0:
        p -= f(lr * wd) * p
    return p, m, v


def linear_schedule(step, t_total, warmup=0):
    """get_linear_schedule_with_warmup lambda: warmup ramp then linear decay to 0."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(t_total - step) / float(max(1, t_total - warmup)))


def clip_coef(total_norm, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), applied iff < 1."""
    c = max_norm / (total_norm + 1e-6)
    return c if c < 1.0 else 1.0
