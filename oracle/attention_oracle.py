"""CPU restatement (numpy fp64) of the reference's masked multi-head attention core.
TEST INFRASTRUCTURE ONLY.

Reference: MultiHeadAttention.forward, src/models/lmu_encoder.py:68-91, with the mask that
LMUEncoder.forward builds at src/models/lmu_encoder.py:247-252 (mask[b, i, j] = 1 for i, j <
lengths[b]); masked scores are set to -1e9, softmax over the keys, heads concatenated.
Pinned through tests/golden/model_e2e.npz (the reference's full model, whose attention layers
the harness evaluates with the kernel under test).
"""
import numpy as np


def attention_core(q, k, v, lengths, num_heads):
    """q, k, v [B, T, C] -> context [B, T, C]; lengths [B] or None."""
    q, k, v = (np.asarray(a, dtype=np.float64) for a in (q, k, v))
    B, T, C = q.shape
    dk = C // num_heads
    split = lambda a: a.reshape(B, T, num_heads, dk).transpose(0, 2, 1, 3)  # noqa: E731
    Q, K, V = split(q), split(k), split(v)
    scores = Q @ K.transpose(0, 1, 3, 2) / np.sqrt(dk)
    if lengths is not None:
        valid = np.arange(T)[None, :] < np.asarray(lengths)[:, None]
        mask = valid[:, :, None] & valid[:, None, :]
        scores = np.where(mask[:, None], scores, -1e9)
    scores = scores - scores.max(axis=-1, keepdims=True)
    w = np.exp(scores)
    w /= w.sum(axis=-1, keepdims=True)
    return (w @ V).transpose(0, 2, 1, 3).reshape(B, T, C)
