"""CPU restatement of the residual add + LayerNorm of the reference's encoder -- TEST INFRASTRUCTURE,
NOT A PRODUCT PATH (only tests/ and __graft_entry__.smoke() may import it).

Follows `output = self.layer_norm(output + residual)` (src/models/lmu_encoder.py:97) and
`x = layer_norm(x)` (src/models/lmu_encoder.py:274).  The reference calls torch.nn.LayerNorm; the
arithmetic restated here is its documented definition: mean and BIASED variance over the last
dimension, y = (v - mean) / sqrt(var + eps) * weight + bias.  Pinned by tests/test_layernorm.py
against torch.nn.functional.layer_norm on the CPU (the implementation the reference runs) and,
through the whole encoder, by tests/golden/model_e2e.npz.
"""
import numpy as np


def add_layer_norm(x, residual, weight, bias, eps=1e-5):
    """x, residual: [..., C] (residual may be None); statistics in float64, result float32."""
    v = np.asarray(x, dtype=np.float32)
    if residual is not None:
        v = v + np.asarray(residual, dtype=np.float32)  # the reference adds in fp32 (autocast promotes)
    v = v.astype(np.float64)
    mean = v.mean(axis=-1, keepdims=True)
    var = ((v - mean) ** 2).mean(axis=-1, keepdims=True)
    y = (v - mean) / np.sqrt(var + eps) * np.asarray(weight, np.float64) + np.asarray(bias, np.float64)
    return y.astype(np.float32)


def add_layer_norm_backward(x, residual, weight, dy, eps=1e-5):
    """(ds, dgamma, dbeta) of y = LayerNorm(x + residual) * weight + bias in float64: ds is the gradient of the
    sum (of x and of residual alike).  What autograd derives for src/models/lmu_encoder.py:97, 274."""
    v = np.asarray(x, dtype=np.float32)
    if residual is not None:
        v = v + np.asarray(residual, dtype=np.float32)
    v = v.astype(np.float64)
    C = v.shape[-1]
    mean = v.mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(((v - mean) ** 2).mean(axis=-1, keepdims=True) + eps)
    xhat = (v - mean) * rstd
    g = np.asarray(dy, dtype=np.float64)
    gy = g * np.asarray(weight, np.float64)
    ds = rstd * (gy - gy.mean(axis=-1, keepdims=True) - xhat * (gy * xhat).mean(axis=-1, keepdims=True))
    return ds, (g * xhat).reshape(-1, C).sum(axis=0), g.reshape(-1, C).sum(axis=0)


def dropout_keep(shape, p_drop, seed0, seed1):
    """keep[row, col] (bool) of the dropout that follows the layer norm in training (`self.dropout_layer(layer_norm(x))`,
    src/models/lmu_encoder.py:274-277; nn.Dropout multiplies the kept elements by 1 / (1 - p)).  torch's random
    stream cannot be reproduced by another implementation; the build decides an element by a hash of its row, its
    column and the call's two seeds (attention_oracle.hash_keep restates it)."""
    from .attention_oracle import hash_keep

    rows, C = int(np.prod(shape[:-1])), shape[-1]
    keep = hash_keep(np.arange(rows, dtype=np.uint64)[:, None], np.arange(C, dtype=np.uint64)[None, :], C, p_drop, seed0, seed1)
    return keep.reshape(shape)
