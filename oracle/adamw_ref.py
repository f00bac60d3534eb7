"""Oracle: AdamW step + warm-up schedule.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Restates what reference transduction_model.py:178 (torch.optim.AdamW, betas (.9,.999), eps 1e-8,
weight_decay=FLAGS.l2=1e-7, decoupled) and :186-189 (linear warm-up lr = it*1e-3/500) compute."""
import math
import torch


def warmup_lr(batch_idx, target_lr=1e-3, warmup=500):
    it = batch_idx + 1
    return it * target_lr / warmup if it <= warmup else None   # None: leave lr unchanged


def adamw_step_ref(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=1e-7):
    """step is 1-based.  Returns new (p, m, v); fp32."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
