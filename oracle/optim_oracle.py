"""CPU restatement (numpy, float64) of the optimizer step of the reference's training loop.  TEST INFRASTRUCTURE:
only tests/ may import this.

Follows DistributedTrainer.train_epoch, src/training/distributed.py:207-234:
  total_norm = sqrt(sum_p ||grad_p||^2)                                  (:210-215)
  NaN norm -> the batch is skipped (no optimizer step)                   (:218-221, :230)
  clip_grad_norm_(parameters, max_norm), src/training/utils.py:653-665   -> grads *= min(1, max_norm / (norm + 1e-6))
  torch.optim.AdamW(lr, betas=(0.9, 0.999), weight_decay), :92-98        -> decoupled decay, bias-corrected moments
  LearningRateScheduler.step, src/training/utils.py:245-263
The arithmetic of clip_grad_norm_ and AdamW lives in torch (pinned 2.1.0 by the reference's
pytorch_lmu_environment.yml:9; 2.10 here); tests/test_optim_oracle.py pins this restatement to the installed torch.
"""
import math

import numpy as np


def global_norm(grads):
    return math.sqrt(sum(float(np.sum(np.asarray(g, dtype=np.float64) ** 2)) for g in grads))


def adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
               max_grad_norm=None):
    """One step in place on lists of float64 arrays; `step` counts from 1.  Returns the gradient norm before clipping
    (None without clipping) and whether the step was applied."""
    norm = None
    coef = 1.0
    if max_grad_norm is not None:
        norm = global_norm(grads)
        if math.isnan(norm):
            return norm, False
        coef = min(1.0, max_grad_norm / (norm + 1e-6))
    b1, b2 = betas
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        g = np.asarray(g, dtype=np.float64) * coef
        p *= 1.0 - lr * weight_decay
        m *= b1
        m += (1.0 - b1) * g
        v *= b2
        v += (1.0 - b2) * g * g
        p -= (lr / bc1) * (m / (np.sqrt(v) / math.sqrt(bc2) + eps))
    return norm, True


def scheduled_lr(step_count, warmup_steps=1000, max_lr=1e-3, decay_steps=10000, decay_rate=0.96):
    """Learning rate after the `step_count`-th call of LearningRateScheduler.step (src/training/utils.py:245-263)."""
    if step_count <= warmup_steps:
        return max_lr * (step_count / warmup_steps)
    return max_lr * decay_rate ** ((step_count - warmup_steps) // decay_steps)


def pack_targets(targets, target_lengths):
    """Labels of all utterances back to back (src/models/asr_model.py:147-150)."""
    out = []
    for i, length in enumerate(target_lengths):
        out.extend(int(t) for t in targets[i][:int(length)])
    return np.asarray(out, dtype=np.int64)
