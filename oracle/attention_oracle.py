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


# ---------------------------------------------------------------------------------------------
# Training form: dropout on the attention weights (src/models/lmu_encoder.py:45, 86) and the backward
# that autograd derives from src/models/lmu_encoder.py:72-90.  The dropout decision of element
# (b, h, query, key) is the counter-based hash the HIP kernels evaluate (fmix32 = MurmurHash3's
# 32-bit finaliser); torch's own dropout stream cannot be reproduced bit for bit by any other
# implementation, so parity for p > 0 is: same semantics on the SAME mask (this function), and the
# keep rate.  With p = 0 the functions below are pinned to torch autograd of the reference's
# formula by tests/test_attention_oracle.py.
# ---------------------------------------------------------------------------------------------
def _fmix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x85EBCA6B)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x * np.uint32(0xC2B2AE35)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def hash_keep(rows, cols, width, p_drop, seed0, seed1):
    """keep (bool, broadcast of rows and cols) of the build's mask-free dropout: the elements of a call form rows of
    `width` columns, columns 2c and 2c + 1 share one 32-bit hash and read its low / high 16 bits:
      g = row * ((width + 1) // 2)  (64 bits);  x = fmix32((lo32(g) + col // 2) ^ seed0 ^ ((hi32(g) + seed1) * 0x9E3779B1));
      keep <=> ((x >> 16 (col & 1)) & 0xFFFF) >= round(p * 65536) clamped to [1, 65535]."""
    rows, cols = np.broadcast_arrays(np.asarray(rows, dtype=np.uint64), np.asarray(cols, dtype=np.uint64))
    if p_drop <= 0:
        return np.ones(rows.shape, dtype=bool)
    thr = np.uint32(min(max(int(p_drop * 65536.0 + 0.5), 1), 65535))
    with np.errstate(over="ignore"):
        g = rows * np.uint64((width + 1) // 2)
        lo = (g & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (g >> np.uint64(32)).astype(np.uint32)
        c = np.uint32(seed0) ^ ((hi + np.uint32(seed1)) * np.uint32(0x9E3779B1)).astype(np.uint32)
        x = _fmix32((lo + (cols >> np.uint64(1)).astype(np.uint32)).astype(np.uint32) ^ c)
    field = (x >> ((cols & np.uint64(1)).astype(np.uint32) * np.uint32(16))) & np.uint32(0xFFFF)
    return field >= thr


def dropout_keep_mask(B, H, T, p_drop, seed0, seed1):
    """keep[b, h, i, j] (bool) for dropout probability p_drop and the call's two 32-bit seeds: row (b, h, i), column j."""
    rows = np.arange(B * H * T, dtype=np.uint64).reshape(B, H, T, 1)
    return hash_keep(rows, np.arange(T, dtype=np.uint64).reshape(1, 1, 1, T), T, p_drop, seed0, seed1)


def attention_train(q, k, v, lengths, num_heads, keep=None, p_drop=0.0):
    """Forward with the weights multiplied by keep / (1 - p_drop); returns (context, cache for the backward)."""
    q, k, v = (np.asarray(a, dtype=np.float64) for a in (q, k, v))
    B, T, C = q.shape
    dk = C // num_heads
    split = lambda a: a.reshape(B, T, num_heads, dk).transpose(0, 2, 1, 3)  # noqa: E731
    Q, K, V = split(q), split(k), split(v)
    scores = Q @ K.transpose(0, 1, 3, 2) / np.sqrt(dk)
    inside = np.ones((B, 1, T, T), dtype=bool)
    if lengths is not None:
        valid = np.arange(T)[None, :] < np.asarray(lengths)[:, None]
        inside = (valid[:, :, None] & valid[:, None, :])[:, None]
        scores = np.where(inside, scores, -1e9)
    scores = scores - scores.max(axis=-1, keepdims=True)
    P = np.exp(scores)
    P /= P.sum(axis=-1, keepdims=True)
    scale = 1.0 / (1.0 - p_drop)
    W = P if keep is None else P * keep * scale
    ctx = (W @ V).transpose(0, 2, 1, 3).reshape(B, T, C)
    return ctx, (Q, K, V, P, W, inside, keep, scale, dk)


def attention_backward(dctx, cache):
    """(dq, dk, dv) [B, T, C] from the upstream gradient of the context."""
    Q, K, V, P, W, inside, keep, scale, dk = cache
    B, H, T, _ = Q.shape
    dO = np.asarray(dctx, dtype=np.float64).reshape(B, T, H, dk).transpose(0, 2, 1, 3)
    dV = W.transpose(0, 1, 3, 2) @ dO
    dW = dO @ V.transpose(0, 1, 3, 2)
    dP = dW if keep is None else dW * keep * scale
    dS = P * (dP - (P * dP).sum(axis=-1, keepdims=True))
    dS = np.where(inside, dS, 0.0)  # masked_fill passes no gradient to the overwritten scores
    dQ = dS @ K / np.sqrt(dk)
    dK = dS.transpose(0, 1, 3, 2) @ Q / np.sqrt(dk)
    merge = lambda a: a.transpose(0, 2, 1, 3).reshape(B, T, H * dk)  # noqa: E731
    return merge(dQ), merge(dK), merge(dV)
