"""CPU pins of oracle/optim_oracle.py: the AdamW / clipping restatement against the installed torch (the library
the reference calls, src/training/distributed.py:92-98 and src/training/utils.py:664), the scheduler's formula,
and the host-side scheduler class."""
import math

import numpy as np
import torch

from oracle import optim_oracle as oo


def _run_torch(shapes, steps, lr, wd, max_norm, seed):
    g = torch.Generator().manual_seed(seed)
    params = [torch.nn.Parameter(torch.randn(s, generator=g, dtype=torch.float64)) for s in shapes]
    opt = torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.999), weight_decay=wd)
    grads_seq, norms = [], []
    for _ in range(steps):
        grads = [torch.randn(s, generator=g, dtype=torch.float64) * 3 for s in shapes]
        grads_seq.append([x.numpy().copy() for x in grads])
        for p, x in zip(params, grads):
            p.grad = x.clone()
        if max_norm is not None:
            norms.append(float(torch.nn.utils.clip_grad_norm_(params, max_norm)))
        opt.step()
    return [p.detach().numpy() for p in params], grads_seq, norms, opt


def test_adamw_and_clipping_match_torch():
    shapes = [(7, 5), (11,), (3, 4, 2), (1,)]
    for max_norm in (None, 10.0, 0.5):
        ref, grads_seq, norms, _ = _run_torch(shapes, 6, 1.5e-3, 0.01, max_norm, seed=5)
        g = torch.Generator().manual_seed(5)
        params = [torch.randn(s, generator=g, dtype=torch.float64).numpy().copy() for s in shapes]
        m = [np.zeros(s) for s in shapes]
        v = [np.zeros(s) for s in shapes]
        for t, grads in enumerate(grads_seq, 1):
            norm, applied = oo.adamw_step(params, grads, m, v, t, 1.5e-3, weight_decay=0.01, max_grad_norm=max_norm)
            assert applied
            if max_norm is not None:
                assert abs(norm - norms[t - 1]) <= 1e-12 * norms[t - 1]
        for a, b in zip(params, ref):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)


def test_nan_norm_skips_the_step():
    p = [np.ones(4)]
    m, v = [np.zeros(4)], [np.zeros(4)]
    norm, applied = oo.adamw_step(p, [np.array([1.0, np.nan, 0.0, 2.0])], m, v, 1, 1e-3, max_grad_norm=10.0)
    assert math.isnan(norm) and not applied
    assert (p[0] == 1).all() and (m[0] == 0).all() and (v[0] == 0).all()


def test_scheduler_formula_and_class():
    from asr_amd.optim import LearningRateScheduler

    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0}]

    opt = Opt()
    sch = LearningRateScheduler(opt, warmup_steps=4, max_lr=1.5e-3, decay_steps=3, decay_rate=0.5)
    got = [sch.step() for _ in range(12)]
    want = [oo.scheduled_lr(i, 4, 1.5e-3, 3, 0.5) for i in range(1, 13)]
    assert got == want
    # hand-checked points: linear warm-up, plateau for decay_steps steps, then halving
    assert got[0] == 1.5e-3 / 4 and got[3] == 1.5e-3 and got[5] == 1.5e-3 and got[6] == 0.75e-3 and got[9] == 0.375e-3
    assert all(g["lr"] == got[-1] for g in opt.param_groups)
    assert sch.step_count == 12


def test_pack_targets_restatement():
    t = np.array([[5, 6, 7, 0], [1, 0, 0, 0], [9, 8, 7, 6]])
    assert oo.pack_targets(t, [3, 1, 4]).tolist() == [5, 6, 7, 1, 9, 8, 7, 6]
    assert oo.pack_targets(t, [0, 0, 0]).tolist() == []
