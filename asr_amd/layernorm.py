"""Residual add + LayerNorm on the GPU in one pass (HIP kernel k_add_layernorm; inference, no autograd).

Mirrors `self.layer_norm(output + residual)` of the reference's attention block
(src/models/lmu_encoder.py:97) and the `x = layer_norm(x)` that follows every LMU layer
(src/models/lmu_encoder.py:274); the dropout after each of them is the identity in eval mode.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .dropout import check_capturable
from .parallel import adopt, grad_slot, register_deferred_sums

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
MAX_WIDTH = 1024


def add_layer_norm(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor,
                   eps: float = 1e-5, out_dtype: Optional[torch.dtype] = None, p_drop: float = 0.0,
                   seeds=(0, 0), twin: bool = False):
    """LayerNorm over the last dimension of `x (+ residual)` with fp32 statistics.

    x, residual: [..., C] float32 or bfloat16 (they may differ); weight, bias: [C].  The result has
    `out_dtype`, by default float32 under autocast (where torch runs layer_norm in fp32) and the
    promoted input dtype otherwise.  `p_drop` > 0: dropout on the result in the same pass (training; the mask
    is a function of `seeds` = two 32-bit integers and the element's index, nothing is stored).
    `twin` (float32 result only): returns (result, the same values rounded to bfloat16) -- the copy the next
    block's projections read under bf16 autocast, written by the same pass instead of a cast kernel.
    """
    for name, t in (("x", x), ("residual", residual), ("weight", weight), ("bias", bias)):
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"add_layer_norm: `{name}` lives on {t.device}; the kernel runs only on a GPU "
                               "(there is no CPU fallback)")
    if residual is not None and residual.shape != x.shape:
        raise RuntimeError(f"add_layer_norm: residual {tuple(residual.shape)} does not match x {tuple(x.shape)}")
    C = x.shape[-1]
    if weight.shape != (C,) or bias.shape != (C,):
        raise RuntimeError(f"add_layer_norm: weight / bias must have shape ({C},)")
    if x.dtype not in _DTYPES or (residual is not None and residual.dtype not in _DTYPES):
        raise RuntimeError("add_layer_norm: supported dtypes are float32 and bfloat16")
    if out_dtype is None:
        if torch.is_autocast_enabled():
            out_dtype = torch.float32
        else:
            out_dtype = x.dtype if residual is None else torch.promote_types(x.dtype, residual.dtype)
    if out_dtype not in _DTYPES:
        raise RuntimeError(f"add_layer_norm: output dtype {out_dtype} is not supported")
    x = x.detach().contiguous()
    residual = None if residual is None else residual.detach().contiguous()
    weight = weight.detach().float().contiguous()
    bias = bias.detach().float().contiguous()
    if twin and out_dtype != torch.float32:
        raise RuntimeError("add_layer_norm: the bfloat16 twin goes with a float32 result")
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    out16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if twin else None
    rows = x.numel() // C if C else 0
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(x.device):
        rc = _lib.lib.lmufft_add_layernorm_dropout(
            p(x), _DTYPES[x.dtype], p(residual), _DTYPES[residual.dtype] if residual is not None else 0, p(weight),
            p(bias), float(eps), rows, C, p(out), _DTYPES[out_dtype], float(p_drop), int(seeds[0]) & 0xFFFFFFFF,
            int(seeds[1]) & 0xFFFFFFFF, p(out16), ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(rc, "lmufft_add_layernorm_dropout")
    return (out, out16) if twin else out


# Layer-norm backwards whose parameter sums wait for one common launch: (device, item, tensors kept alive).
_PENDING_PARAM_SUMS = []
# True: a layer-norm backward whose parameter gradients go straight into a gradient bucket leaves the sum of its
# workgroups' partial results to ONE launch for all such layer norms of the step (flush_param_sums; GradBuckets
# calls it before it exchanges or hands out gradients).
DEFER_PARAM_SUMS = True


def flush_param_sums() -> None:
    """Sum the partial parameter gradients of every layer-norm backward that deferred them: one launch per device on
    the current stream, the same numbers as the immediate sums."""
    if not _PENDING_PARAM_SUMS:
        return
    pending = list(_PENDING_PARAM_SUMS)
    _PENDING_PARAM_SUMS.clear()
    for dev in {d for d, _, _ in pending}:
        items = [it for d, it, _ in pending if d == dev]
        arr = (_lib.LnSumItem * len(items))(*items)
        with torch.cuda.device(dev):
            rc = _lib.lib.lmufft_layernorm_param_sum_many(arr, len(items),
                                                          ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "lmufft_layernorm_param_sum_many")


register_deferred_sums(flush_param_sums)


def add_layer_norm_backward(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, dy: torch.Tensor,
                            eps: float = 1e-5, out_weight: Optional[torch.Tensor] = None,
                            out_bias: Optional[torch.Tensor] = None, p_drop: float = 0.0, seeds=(0, 0),
                            dy2: Optional[torch.Tensor] = None, defer_param_sums: bool = False,
                            relu_input: bool = False):
    """(dx, dresidual or None, dweight, dbias) of add_layer_norm: dx / dresidual carry the same values in the dtype
    of x / residual (one tensor when the dtypes agree); dweight, dbias float32, written into `out_weight` /
    `out_bias` (contiguous float32 [C]) when given.  `p_drop`, `seeds`: those of the forward call; dy passes
    through the same dropout decisions first.  `dy2`: a second upstream gradient of the same shape (float32 or
    bfloat16), added to dy on load.  `relu_input`: x is the output of a ReLU whose backward is the next thing to
    happen to dx -- dx is returned already multiplied by (x > 0) (dresidual is not).  HIP kernel k_layernorm_bwd."""
    for name, t in (("x", x), ("residual", residual), ("weight", weight), ("dy", dy)):
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"add_layer_norm_backward: `{name}` lives on {t.device}; the kernel runs only on a GPU "
                               "(there is no CPU fallback)")
    C = x.shape[-1]
    if dy.shape != x.shape or (residual is not None and residual.shape != x.shape) or weight.shape != (C,):
        raise RuntimeError("add_layer_norm_backward: x, residual and dy must share one shape [..., C], weight [C]")
    if x.dtype not in _DTYPES or dy.dtype not in _DTYPES or (residual is not None and residual.dtype not in _DTYPES):
        raise RuntimeError("add_layer_norm_backward: supported dtypes are float32 and bfloat16")
    x, dy = x.detach().contiguous(), dy.detach().contiguous()
    if dy2 is not None:
        if dy2.shape != x.shape or dy2.dtype not in _DTYPES or not dy2.is_cuda:
            raise RuntimeError("add_layer_norm_backward: dy2 must match x's shape, float32 or bfloat16, on the GPU")
        dy2 = dy2.detach().contiguous()
    residual = None if residual is None else residual.detach().contiguous()
    weight = weight.detach().float().contiguous()
    dx = torch.empty_like(x)
    dres = None
    if residual is not None and (residual.dtype != x.dtype or relu_input):
        dres = torch.empty_like(residual)
    for name, t in (("out_weight", out_weight), ("out_bias", out_bias)):
        if t is not None and (t.shape != (C,) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != x.device):
            raise RuntimeError(f"add_layer_norm_backward: `{name}` must be a contiguous float32 tensor [{C}] on {x.device}")
    dgamma = out_weight if out_weight is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = out_bias if out_bias is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib.lib.lmufft_layernorm_bwd_workspace_bytes(C), dtype=torch.uint8, device=x.device)
    rows = x.numel() // C
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(x.device):
        rc = _lib.lib.lmufft_layernorm_dropout_bwd_relu(
            p(x), _DTYPES[x.dtype], p(residual), _DTYPES[residual.dtype] if residual is not None else 0, p(dy),
            _DTYPES[dy.dtype], p(dy2), _DTYPES[dy2.dtype] if dy2 is not None else 0, p(weight), float(eps), rows, C,
            float(p_drop), int(seeds[0]) & 0xFFFFFFFF,
            int(seeds[1]) & 0xFFFFFFFF, p(dx), p(dres), None if defer_param_sums else p(dgamma),
            None if defer_param_sums else p(dbeta), p(ws), ws.numel(),
            ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), 1 if relu_input else 0)
    _lib.check(rc, "lmufft_layernorm_dropout_bwd_relu")
    if defer_param_sums:  # dgamma / dbeta are complete after flush_param_sums()
        _PENDING_PARAM_SUMS.append((x.device, _lib.LnSumItem(rows, C, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr()),
                                    (ws, dgamma, dbeta)))
    if residual is None:
        return dx, None, dgamma, dbeta
    return dx, (dx if dres is None else dres), dgamma, dbeta


class _AddLayerNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps, out_dtype, p_drop, seed0, seed1, twin, relu_input=False):
        ctx.save_for_backward(x, residual, weight)
        ctx.relu_input = bool(relu_input)
        ctx.eps = eps
        ctx.params = (weight, bias)
        ctx.drop = (p_drop, (seed0, seed1))
        ctx.twin = twin
        ctx.set_materialize_grads(False)  # an unused copy hands the backward None, not a tensor of zeros
        return add_layer_norm(x, residual, weight, bias, eps, out_dtype, p_drop, (seed0, seed1), twin)

    @staticmethod
    def backward(ctx, dy, dy16=None):
        x, residual, weight = ctx.saved_tensors
        if dy is None:  # only the bfloat16 copy was used
            dy, dy16 = dy16, None
        # parameters of a GradBuckets: the kernel writes into the bucket and autograd adopts the slices as .grad
        slot_w, slot_b = grad_slot(ctx.params[0]), grad_slot(ctx.params[1])
        defer = DEFER_PARAM_SUMS and slot_w is not None and slot_b is not None
        if not defer and _PENDING_PARAM_SUMS:
            flush_param_sums()  # a gradient handed to autograd may be added into a slice that a deferred sum still owes
        dx, dres, dgamma, dbeta = add_layer_norm_backward(x, residual, weight, dy, ctx.eps, slot_w, slot_b, *ctx.drop, dy16,
                                                          defer_param_sums=defer, relu_input=ctx.relu_input)
        dgamma = adopt(dgamma) if slot_w is not None else dgamma.to(weight.dtype)
        dbeta = adopt(dbeta) if slot_b is not None else dbeta.to(weight.dtype)
        return (dx if ctx.needs_input_grad[0] else None, dres if ctx.needs_input_grad[1] else None,
                dgamma, dbeta, None, None, None, None, None, None, None)


_dropout_calls = 0


def add_layer_norm_train(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor,
                         eps: float = 1e-5, out_dtype: Optional[torch.dtype] = None, p_drop: float = 0.0,
                         seeds=None, twin: bool = False, relu_input: bool = False):
    """add_layer_norm with autograd (forward k_add_layernorm, backward k_layernorm_bwd), optionally followed by
    dropout in the same passes (`self.dropout_layer(layer_norm(x))`, src/models/lmu_encoder.py:274-277).
    `twin`: returns (float32 result, its bfloat16 rounding) as two outputs; the gradients that arrive through the two
    are added inside the backward kernel (no cast, no accumulation kernel).
    `seeds` = (seed0, seed1) name the dropout mask; by default torch's seed and a per-process call counter, so runs
    that start from the same torch.manual_seed and make the same calls draw the same masks.
    `relu_input`: x is the output of a ReLU (an LMUFFT layer's h) with no other consumer; the gradient returned for x
    is already multiplied by (x > 0), so that layer can skip its own masking pass (LMUFFT.grad_is_masked)."""
    global _dropout_calls
    if not 0.0 <= p_drop < 1.0:
        raise ValueError(f"dropout probability has to be in [0, 1), got {p_drop}")
    if out_dtype is None:
        out_dtype = torch.float32 if torch.is_autocast_enabled() else (
            x.dtype if residual is None else torch.promote_types(x.dtype, residual.dtype))
    check_capturable(x.device, p_drop, seeds is not None)
    if seeds is None:
        if p_drop > 0:
            _dropout_calls += 1
        seeds = (torch.initial_seed() & 0xFFFFFFFF, 0x4C4E0000 ^ (_dropout_calls & 0xFFFFFFFF))
    return _AddLayerNormTrain.apply(x, residual, weight, bias, eps, out_dtype, float(p_drop), int(seeds[0]) & 0xFFFFFFFF,
                                    int(seeds[1]) & 0xFFFFFFFF, bool(twin), bool(relu_input))
