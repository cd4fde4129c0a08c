"""CPU pin of the attention oracle's training form: forward and backward (no dropout, and dropout with an
explicit mask) against torch autograd of the reference's formula (src/models/lmu_encoder.py:72-90: scores,
masked_fill(mask == 0, -1e9), softmax, weights * mask / (1 - p), weights @ V), plus the hash's keep rate."""
import math

import numpy as np
import torch

from oracle import attention_oracle as ao


def _torch_reference(q, k, v, lengths, H, keep, p):
    B, T, C = q.shape
    dk = C // H
    q, k, v = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (q, k, v))
    split = lambda t: t.view(B, T, H, dk).transpose(1, 2)  # noqa: E731
    scores = split(q) @ split(k).transpose(-2, -1) / math.sqrt(dk)
    if lengths is not None:
        mask = torch.zeros(B, T, T)
        for i, n in enumerate(lengths):
            mask[i, :n, :n] = 1
        scores = scores.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    w = torch.softmax(scores, dim=-1)
    if keep is not None:
        w = w * torch.tensor(keep, dtype=torch.float64) / (1 - p)
    ctx = (w @ split(v)).transpose(1, 2).contiguous().view(B, T, C)
    return ctx, (q, k, v)


def test_training_form_matches_torch_autograd():
    rng = np.random.default_rng(0)
    B, T, H, dk = 3, 21, 2, 8
    q, k, v, g = (rng.standard_normal((B, T, H * dk)) for _ in range(4))
    for lengths, p in ((None, 0.0), ([21, 13, 1], 0.0), ([21, 13, 5], 0.3)):
        keep = ao.dropout_keep_mask(B, H, T, p, 11, 22) if p else None
        ctx, cache = ao.attention_train(q, k, v, lengths, H, keep, p)
        dq, dk_, dv = ao.attention_backward(g, cache)
        tctx, (tq, tk, tv) = _torch_reference(q, k, v, lengths, H, keep, p)
        tctx.backward(torch.tensor(g))
        np.testing.assert_allclose(ctx, tctx.detach().numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(dq, tq.grad.numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(dk_, tk.grad.numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(dv, tv.grad.numpy(), rtol=1e-9, atol=1e-12)
        if p == 0.0:
            np.testing.assert_allclose(ctx, ao.attention_core(q, k, v, lengths, H), rtol=1e-12, atol=1e-14)


def test_keep_mask_rate_and_seeds():
    m = ao.dropout_keep_mask(4, 6, 64, 0.1, 1234, 7)
    assert abs(m.mean() - 0.9) < 3e-3           # 98,304 draws: sigma = 1e-3
    assert abs(m[:, :, :, ::2].mean() - 0.9) < 5e-3 and abs(m[:, :, 1::2].mean() - 0.9) < 5e-3
    assert (ao.dropout_keep_mask(4, 6, 64, 0.1, 1234, 8) != m).mean() > 0.1   # another seed, another mask
    assert ao.dropout_keep_mask(1, 1, 8, 0.0, 1, 2).all()
    # neighbouring elements are uncorrelated
    a, b = m[..., :-1].ravel().astype(float), m[..., 1:].ravel().astype(float)
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.01
