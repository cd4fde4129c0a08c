"""torch.autograd binding of the HIP LMUFFT layer.

PyTorch supplies device memory, the current HIP stream and autograd graph
bookkeeping; all arithmetic runs in liblmufft_hip.so through its C ABI.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

_ALGOS = {"auto": _lib.CONV_AUTO, "toeplitz": _lib.CONV_TOEPLITZ, "fft": _lib.CONV_FFT}


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"LMUFFT: `{name}` lives on {t.device}; the layer runs only as HIP kernels on a GPU "
            "(there is no CPU fallback)"
        )


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _shape(B, T, D, M, Hd, ldH) -> _lib.Shape:
    return _lib.Shape(int(B), int(T), int(D), int(M), int(Hd), int(ldH))


class _LMUFFTFunction(torch.autograd.Function):
    """(x, w_u, b_u, W_h, b_h, H) -> (h, h_n); reference: src/lmu.py:290-320."""

    @staticmethod
    def forward(ctx, x, w_u, b_u, W_h, b_h, H, conv_algo):
        for name, t in (("x", x), ("W_u.weight", w_u), ("W_u.bias", b_u), ("W_h.weight", W_h),
                        ("W_h.bias", b_h), ("H", H)):
            _require_gpu(t, name)
            if t.dtype != torch.float32:
                raise RuntimeError(f"LMUFFT: `{name}` has dtype {t.dtype}; this build computes in float32")
            if t.device != x.device:
                raise RuntimeError(f"LMUFFT: `{name}` is on {t.device}, x on {x.device}")
        if x.dim() != 3:
            raise RuntimeError(f"LMUFFT: x must be [batch, seq_len, input_size], got {tuple(x.shape)}")
        B, T, D = x.shape
        M, ldH = H.shape
        Hd = W_h.shape[0]
        if w_u.shape != (1, D):
            raise RuntimeError(f"LMUFFT: input_size mismatch: x has {D} features, W_u expects {w_u.shape[1]}")
        if W_h.shape[1] != M + D:
            raise RuntimeError(f"LMUFFT: W_h has {W_h.shape[1]} columns, expected memory+input = {M + D}")
        if T > ldH:
            raise RuntimeError(
                f"LMUFFT: sequence length {T} exceeds the layer's seq_len {ldH} "
                "(the reference fails likewise when they differ, src/lmu.py:308)"
            )
        x = x.contiguous()
        w_u = w_u.contiguous()
        W_h = W_h.contiguous()
        H = H.contiguous()
        dev = x.device
        u = torch.empty((B, T), dtype=torch.float32, device=dev)
        m = torch.empty((B, T, M), dtype=torch.float32, device=dev)
        h = torch.empty((B, T, Hd), dtype=torch.float32, device=dev)
        h_n = torch.empty((B, Hd), dtype=torch.float32, device=dev)
        shp = _shape(B, T, D, M, Hd, ldH)
        with torch.cuda.device(dev):
            ws = _workspace(_lib.lib.lmufft_forward_workspace_bytes(ctypes.byref(shp), _lib.F32), dev)
            rc = _lib.lib.lmufft_forward_f32(
                ctypes.byref(shp), _ptr(x), _ptr(w_u), _ptr(b_u), _ptr(W_h), _ptr(b_h), _ptr(H),
                _ptr(u), _ptr(m), _ptr(h), _ptr(h_n), conv_algo, _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "lmufft_forward_f32")
        ctx.save_for_backward(x, w_u, W_h, H, u, m, h)
        ctx.conv_algo = conv_algo
        ctx.shape = (B, T, D, M, Hd, ldH)
        ctx.set_materialize_grads(False)
        return h, h_n

    @staticmethod
    def backward(ctx, gh, gh_n):
        x, w_u, W_h, H, u, m, h = ctx.saved_tensors
        B, T, D, M, Hd, ldH = ctx.shape
        dev = x.device
        need_x = ctx.needs_input_grad[0]
        if gh is None and gh_n is None:
            return (None,) * 7
        if gh is not None:
            gh = gh.contiguous()
        if gh_n is not None:
            gh_n = gh_n.contiguous()
        gx = torch.empty_like(x) if need_x else None
        gw_u = torch.empty_like(w_u)
        gb_u = torch.empty((1,), dtype=torch.float32, device=dev)
        gW_h = torch.empty_like(W_h)
        gb_h = torch.empty((Hd,), dtype=torch.float32, device=dev)
        gm = torch.empty_like(m)
        shp = _shape(B, T, D, M, Hd, ldH)
        with torch.cuda.device(dev):
            ws = _workspace(_lib.lib.lmufft_backward_workspace_bytes(ctypes.byref(shp), _lib.F32), dev)
            rc = _lib.lib.lmufft_backward_f32(
                ctypes.byref(shp), _ptr(x), _ptr(w_u), _ptr(W_h), _ptr(H), _ptr(u), _ptr(m), _ptr(h),
                _ptr(gh), _ptr(gh_n), _ptr(gx), _ptr(gw_u), _ptr(gb_u), _ptr(gW_h), _ptr(gb_h), _ptr(gm),
                ctx.conv_algo, _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "lmufft_backward_f32")
        return gx, gw_u, gb_u, gW_h, gb_h, None, None


def lmufft_layer(x, w_u, b_u, W_h, b_h, H, conv_algo="auto"):
    """h, h_n = LMUFFT forward with autograd support (HIP kernels only)."""
    return _LMUFFTFunction.apply(x, w_u, b_u, W_h, b_h, H, _ALGOS[conv_algo])


def causal_conv(u, H, conv_algo="auto"):
    """m[b,t,k] = sum_{s<=t} H[k,t-s] u[b,s]  (no autograd; kernel-level entry point)."""
    _require_gpu(u, "u")
    _require_gpu(H, "H")
    u = u.contiguous()
    H = H.contiguous()
    B, T = u.shape
    M, ldH = H.shape
    m = torch.empty((B, T, M), dtype=torch.float32, device=u.device)
    shp = _shape(B, T, 1, M, 1, ldH)
    with torch.cuda.device(u.device):
        rc = _lib.lib.lmufft_conv_forward_f32(ctypes.byref(shp), _ptr(u), _ptr(H), _ptr(m), _ALGOS[conv_algo],
                                              None, 0, _stream(u.device))
    _lib.check(rc, "lmufft_conv_forward_f32")
    return m


def causal_conv_adjoint(gm, H, conv_algo="auto"):
    """gu[b,s] = sum_{t>=s} sum_k H[k,t-s] gm[b,t,k]  (no autograd)."""
    _require_gpu(gm, "gm")
    _require_gpu(H, "H")
    gm = gm.contiguous()
    H = H.contiguous()
    B, T, M = gm.shape
    ldH = H.shape[1]
    gu = torch.empty((B, T), dtype=torch.float32, device=gm.device)
    shp = _shape(B, T, 1, M, 1, ldH)
    algo = _ALGOS[conv_algo]
    with torch.cuda.device(gm.device):
        ws = _workspace(_lib.lib.lmufft_conv_workspace_bytes(ctypes.byref(shp), _lib.F32, algo), gm.device)
        rc = _lib.lib.lmufft_conv_backward_f32(ctypes.byref(shp), _ptr(gm), _ptr(H), _ptr(gu), algo, _ptr(ws),
                                               ws.numel(), _stream(gm.device))
    _lib.check(rc, "lmufft_conv_backward_f32")
    return gu
