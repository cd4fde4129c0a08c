"""Masked self-attention core on the GPU (HIP kernels k_attention_fwd for inference; k_attention_fwd with
dropout + k_attention_bwd_dq / k_attention_bwd_dkv behind an autograd function for training).

Mirrors the score / softmax / context part of the reference's MultiHeadAttention.forward
(src/models/lmu_encoder.py:68-91) with the length mask of LMUEncoder.forward (:247-252), on the
[batch, frames, heads * d_k] tensors that its Linear layers produce and consume.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .dropout import check_capturable

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
SUPPORTED_HEAD_WIDTHS = (32, 64, 128)


def masked_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int,
                     lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T / sqrt(d_k), masked by lengths) v per head; q, k, v, result: [B, T, heads * d_k]."""
    for name, t in (("q", q), ("k", k), ("v", v)):
        if not t.is_cuda:
            raise RuntimeError(f"masked_attention: `{name}` lives on {t.device}; the kernel runs only on a GPU "
                               "(there is no CPU fallback)")
    if q.dim() != 3 or q.shape != k.shape or q.shape != v.shape:
        raise RuntimeError(f"masked_attention: q, k, v must share one [batch, frames, channels] shape, got "
                           f"{tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)}")
    if q.dtype not in _DTYPES or k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError(f"masked_attention: dtypes {q.dtype}, {k.dtype}, {v.dtype}; supported: all float32 or "
                           "all bfloat16")
    B, T, C = q.shape
    if C % num_heads:
        raise RuntimeError(f"masked_attention: {C} channels do not split into {num_heads} heads")
    dk = C // num_heads
    q, k, v = q.detach().contiguous(), k.detach().contiguous(), v.detach().contiguous()
    if lengths is not None:
        lengths = lengths.to(device=q.device, dtype=torch.int32).contiguous()
    ctx = torch.empty_like(q)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(q.device):
        rc = _lib.lib.lmufft_attention_fwd(p(q), p(k), p(v), _DTYPES[q.dtype], p(lengths), B, T, num_heads, dk, p(ctx),
                                           ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
    _lib.check(rc, "lmufft_attention_fwd")
    return ctx


def masked_attention_packed(qkv: torch.Tensor, num_heads: int, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
    """masked_attention on the output [B, T, 3 * heads * d_k] of ONE projection whose weight is the
    concatenation of w_q, w_k, w_v (rows in that order): q, k, v are read in place as column blocks."""
    if not qkv.is_cuda:
        raise RuntimeError(f"masked_attention_packed: `qkv` lives on {qkv.device}; the kernel runs only on a GPU "
                           "(there is no CPU fallback)")
    if qkv.dim() != 3 or qkv.shape[2] % (3 * num_heads):
        raise RuntimeError(f"masked_attention_packed: expected [batch, frames, 3 * heads * d_k], got {tuple(qkv.shape)}")
    if qkv.dtype not in _DTYPES:
        raise RuntimeError(f"masked_attention_packed: dtype {qkv.dtype}; supported: float32, bfloat16")
    B, T, C3 = qkv.shape
    C = C3 // 3
    dk = C // num_heads
    qkv = qkv.detach().contiguous()
    if lengths is not None:
        lengths = lengths.to(device=qkv.device, dtype=torch.int32).contiguous()
    ctx = torch.empty((B, T, C), dtype=qkv.dtype, device=qkv.device)
    es = qkv.element_size()
    base = qkv.data_ptr()
    vp = ctypes.c_void_p
    with torch.cuda.device(qkv.device):
        rc = _lib.lib.lmufft_attention_fwd_ld(
            vp(base), vp(base + C * es), vp(base + 2 * C * es), _DTYPES[qkv.dtype], C3,
            None if lengths is None else vp(lengths.data_ptr()), B, T, num_heads, dk, vp(ctx.data_ptr()),
            vp(torch.cuda.current_stream(qkv.device).cuda_stream))
    _lib.check(rc, "lmufft_attention_fwd_ld")
    return ctx


class _AttentionTrain(torch.autograd.Function):
    """q, k, v [B, T, heads * d_k] -> context, with dropout on the attention weights; the backward recomputes
    the weights tile by tile from the saved row statistics and the dropout decisions from the seeds."""

    @staticmethod
    def forward(ctx, q, k, v, lengths, num_heads, p_drop, seed0, seed1):
        B, T, C = q.shape
        dk = C // num_heads
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = torch.empty_like(q)
        stats = torch.empty((B, num_heads, T, 2), dtype=torch.float32, device=q.device)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(q.device):
            rc = _lib.lib.lmufft_attention_fwd_train(
                p(q), p(k), p(v), _DTYPES[q.dtype], C, p(lengths), B, T, num_heads, dk, float(p_drop), seed0, seed1,
                p(out), p(stats), ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
        _lib.check(rc, "lmufft_attention_fwd_train")
        ctx.save_for_backward(q, k, v, out, stats, lengths)
        ctx.conf = (num_heads, float(p_drop), seed0, seed1)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, stats, lengths = ctx.saved_tensors
        num_heads, p_drop, seed0, seed1 = ctx.conf
        B, T, C = q.shape
        dk = C // num_heads
        dout = dout.contiguous()
        if dout.dtype != q.dtype:
            dout = dout.to(q.dtype)
        dq, dkk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        dsum = torch.empty((B, num_heads, T), dtype=torch.float32, device=q.device)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(q.device):
            rc = _lib.lib.lmufft_attention_bwd(
                p(q), p(k), p(v), p(out), p(dout), p(stats), _DTYPES[q.dtype], C, p(lengths), B, T, num_heads, dk, p_drop,
                seed0, seed1, p(dq), p(dkk), p(dv), C, p(dsum),
                ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
        _lib.check(rc, "lmufft_attention_bwd")
        return dq, dkk, dv, None, None, None, None, None


class _AttentionTrainPacked(torch.autograd.Function):
    """_AttentionTrain on qkv [B, T, 3 * heads * d_k], the output of ONE projection over the concatenated weights
    (q | k | v as column blocks); the backward writes dq | dk | dv as the column blocks of one tensor, the
    projection's output gradient."""

    @staticmethod
    def forward(ctx, qkv, lengths, num_heads, p_drop, seed0, seed1):
        B, T, C3 = qkv.shape
        C = C3 // 3
        dk = C // num_heads
        qkv = qkv.contiguous()
        es = qkv.element_size()
        out = torch.empty((B, T, C), dtype=qkv.dtype, device=qkv.device)
        stats = torch.empty((B, num_heads, T, 2), dtype=torch.float32, device=qkv.device)
        base = qkv.data_ptr()
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(qkv.device):
            rc = _lib.lib.lmufft_attention_fwd_train(
                ctypes.c_void_p(base), ctypes.c_void_p(base + C * es), ctypes.c_void_p(base + 2 * C * es),
                _DTYPES[qkv.dtype], C3, p(lengths), B, T, num_heads, dk, float(p_drop), seed0, seed1, p(out), p(stats),
                ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream))
        _lib.check(rc, "lmufft_attention_fwd_train")
        ctx.save_for_backward(qkv, out, stats, lengths)
        ctx.conf = (num_heads, float(p_drop), seed0, seed1)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, stats, lengths = ctx.saved_tensors
        num_heads, p_drop, seed0, seed1 = ctx.conf
        B, T, C3 = qkv.shape
        C = C3 // 3
        dk = C // num_heads
        es = qkv.element_size()
        dout = dout.contiguous()
        if dout.dtype != qkv.dtype:
            dout = dout.to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        dsum = torch.empty((B, num_heads, T), dtype=torch.float32, device=qkv.device)
        base, gbase = qkv.data_ptr(), dqkv.data_ptr()
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(qkv.device):
            rc = _lib.lib.lmufft_attention_bwd(
                ctypes.c_void_p(base), ctypes.c_void_p(base + C * es), ctypes.c_void_p(base + 2 * C * es), p(out), p(dout),
                p(stats), _DTYPES[qkv.dtype], C3, p(lengths), B, T, num_heads, dk, p_drop, seed0, seed1,
                ctypes.c_void_p(gbase), ctypes.c_void_p(gbase + C * es), ctypes.c_void_p(gbase + 2 * C * es), C3, p(dsum),
                ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream))
        _lib.check(rc, "lmufft_attention_bwd")
        return dqkv, None, None, None, None, None


_dropout_calls = 0


def masked_attention_train(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int,
                           lengths: Optional[torch.Tensor] = None, p_drop: float = 0.0, seeds=None) -> torch.Tensor:
    """masked_attention with autograd and dropout on the attention weights (what MultiHeadAttention.forward does
    in training, src/models/lmu_encoder.py:72-90).  `seeds` = (seed0, seed1), two 32-bit integers naming the
    dropout mask of this call; by default torch's seed and a per-process call counter, so runs that start from
    the same torch.manual_seed and make the same calls draw the same masks.  Under HIP graph capture the seeds are
    frozen into the graph; the library adds the device's dropout epoch counter to them at run time
    (asr_amd.dropout), which the captured step increments -- without a registered counter such a capture raises."""
    global _dropout_calls
    for name, t in (("q", q), ("k", k), ("v", v)):
        if not t.is_cuda:
            raise RuntimeError(f"masked_attention_train: `{name}` lives on {t.device}; the kernels run only on a GPU "
                               "(there is no CPU fallback)")
    if q.dim() != 3 or q.shape != k.shape or q.shape != v.shape:
        raise RuntimeError(f"masked_attention_train: q, k, v must share one [batch, frames, channels] shape, got "
                           f"{tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)}")
    if q.dtype not in _DTYPES or k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError(f"masked_attention_train: dtypes {q.dtype}, {k.dtype}, {v.dtype}; supported: all float32 or "
                           "all bfloat16")
    C = q.shape[2]
    if C % num_heads or C // num_heads not in SUPPORTED_HEAD_WIDTHS:
        raise RuntimeError(f"masked_attention_train: {C} channels over {num_heads} heads; head widths supported: "
                           f"{SUPPORTED_HEAD_WIDTHS}")
    if not 0.0 <= p_drop < 1.0:
        raise ValueError(f"dropout probability has to be in [0, 1), got {p_drop}")
    if lengths is not None:
        lengths = lengths.to(device=q.device, dtype=torch.int32).contiguous()
    check_capturable(q.device, p_drop, seeds is not None)
    if seeds is None:
        _dropout_calls += 1
        seeds = (torch.initial_seed() & 0xFFFFFFFF, _dropout_calls & 0xFFFFFFFF)
    return _AttentionTrain.apply(q, k, v, lengths, num_heads, p_drop, int(seeds[0]) & 0xFFFFFFFF,
                                 int(seeds[1]) & 0xFFFFFFFF)


def masked_attention_train_packed(qkv: torch.Tensor, num_heads: int, lengths: Optional[torch.Tensor] = None,
                                  p_drop: float = 0.0, seeds=None) -> torch.Tensor:
    """masked_attention_train with q, k, v given as the column blocks of one tensor [batch, frames, 3 * channels]
    (w_q, w_k, w_v of src/models/lmu_encoder.py:41-43, 68-70 applied as a single GEMM over the concatenated
    weights); the gradient comes back as one tensor of the same layout.  Seeds as in masked_attention_train."""
    global _dropout_calls
    if not qkv.is_cuda:
        raise RuntimeError(f"masked_attention_train_packed: `qkv` lives on {qkv.device}; the kernels run only on a GPU "
                           "(there is no CPU fallback)")
    if qkv.dim() != 3 or qkv.shape[2] % 3:
        raise RuntimeError(f"masked_attention_train_packed: qkv must be [batch, frames, 3 * channels], got {tuple(qkv.shape)}")
    if qkv.dtype not in _DTYPES:
        raise RuntimeError(f"masked_attention_train_packed: dtype {qkv.dtype}; supported: float32, bfloat16")
    C = qkv.shape[2] // 3
    if C % num_heads or C // num_heads not in SUPPORTED_HEAD_WIDTHS:
        raise RuntimeError(f"masked_attention_train_packed: {C} channels over {num_heads} heads; head widths supported: "
                           f"{SUPPORTED_HEAD_WIDTHS}")
    if not 0.0 <= p_drop < 1.0:
        raise ValueError(f"dropout probability has to be in [0, 1), got {p_drop}")
    if lengths is not None:
        lengths = lengths.to(device=qkv.device, dtype=torch.int32).contiguous()
    check_capturable(qkv.device, p_drop, seeds is not None)
    if seeds is None:
        _dropout_calls += 1
        seeds = (torch.initial_seed() & 0xFFFFFFFF, _dropout_calls & 0xFFFFFFFF)
    return _AttentionTrainPacked.apply(qkv, lengths, num_heads, p_drop, int(seeds[0]) & 0xFFFFFFFF,
                                       int(seeds[1]) & 0xFFFFFFFF)
