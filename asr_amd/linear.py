"""Linear layers of the encoder on the GPU for bf16 inference (HIP kernel k_linear_glds_bf16; no autograd).

Mirrors torch.nn.Linear under bf16 autocast as the reference runs it -- input_projection /
output_projection (src/models/lmu_encoder.py:184, 229) and the attention block's w_q, w_k, w_v,
w_o (src/models/lmu_encoder.py:40-43) -- and, for w_o, the `self.layer_norm(output + residual)`
that follows it (src/models/lmu_encoder.py:94-97) in the same kernel.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .parallel import adopt, fused_slot, grad_slot, register_deferred_sums

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
FUSED_LAYER_NORM_MAX_WIDTH = 384


def supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """True when linear_bf16 can take this call: bf16 operands on a GPU, K and N multiples of 8."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.dim() == 2
            and x.shape[-1] == weight.shape[1] and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0)


def _check(x, weight, bias):
    for name, t in (("x", x), ("weight", weight), ("bias", bias)):
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"linear_bf16: `{name}` lives on {t.device}; the kernel runs only on a GPU "
                               "(there is no CPU fallback)")
    if x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
        raise RuntimeError("linear_bf16: x and weight must be bfloat16")
    if weight.dim() != 2 or x.shape[-1] != weight.shape[1]:
        raise RuntimeError(f"linear_bf16: weight {tuple(weight.shape)} does not match x {tuple(x.shape)}")
    if bias is not None and (bias.dtype != torch.float32 or bias.shape != (weight.shape[0],)):
        raise RuntimeError(f"linear_bf16: bias must be float32 of shape ({weight.shape[0]},)")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def linear_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [..., K] bf16, weight [N, K] bf16 (nn.Linear layout), bias [N] fp32 or None -> [..., N] bf16.

    The bias is added in fp32 before the single rounding; pass the bf16-rounded bias widened to
    fp32 to reproduce autocast's arithmetic exactly.
    """
    _check(x, weight, bias)
    N, K = weight.shape
    x = x.detach().contiguous()
    weight = weight.detach().contiguous()
    bias = None if bias is None else bias.detach().contiguous()
    out = torch.empty(x.shape[:-1] + (N,), dtype=torch.bfloat16, device=x.device)
    rows = x.numel() // K
    with torch.cuda.device(x.device):
        rc = _lib.lib.lmufft_linear_bf16(_ptr(x), _ptr(weight), _ptr(bias), rows, K, N, _ptr(out),
                                         ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(rc, "lmufft_linear_bf16")
    return out


def linear_add_layer_norm_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                               residual: Optional[torch.Tensor], ln_weight: torch.Tensor, ln_bias: torch.Tensor,
                               eps: float = 1e-5, out_dtype: torch.dtype = torch.float32,
                               return_projection: bool = False):
    """LayerNorm(bf16(x weight^T + bias) + residual) in one kernel; N <= 384.

    residual: [..., N] float32 or bfloat16, or None; the result has `out_dtype` (float32 or bfloat16).
    With `return_projection` the bf16 projection is written too and (y, projection) is returned.
    """
    _check(x, weight, bias)
    N, K = weight.shape
    if N > FUSED_LAYER_NORM_MAX_WIDTH:
        raise RuntimeError(f"linear_add_layer_norm_bf16: width {N} exceeds {FUSED_LAYER_NORM_MAX_WIDTH}")
    shape = x.shape[:-1] + (N,)
    if residual is not None:
        if not residual.is_cuda or residual.dtype not in _DTYPES or residual.shape != shape:
            raise RuntimeError(f"linear_add_layer_norm_bf16: residual must be a float32 / bfloat16 GPU tensor of shape "
                               f"{tuple(shape)}")
        residual = residual.detach().contiguous()
    if out_dtype not in _DTYPES:
        raise RuntimeError(f"linear_add_layer_norm_bf16: output dtype {out_dtype} is not supported")
    if ln_weight.shape != (N,) or ln_bias.shape != (N,) or not ln_weight.is_cuda or not ln_bias.is_cuda:
        raise RuntimeError(f"linear_add_layer_norm_bf16: layer-norm weight / bias must be GPU tensors of shape ({N},)")
    x = x.detach().contiguous()
    weight = weight.detach().contiguous()
    bias = None if bias is None else bias.detach().contiguous()
    g = ln_weight.detach().float().contiguous()
    b = ln_bias.detach().float().contiguous()
    y = torch.empty(shape, dtype=out_dtype, device=x.device)
    proj = torch.empty(shape, dtype=torch.bfloat16, device=x.device) if return_projection else None
    rows = x.numel() // K
    with torch.cuda.device(x.device):
        rc = _lib.lib.lmufft_linear_ln_bf16(
            _ptr(x), _ptr(weight), _ptr(bias), rows, K, N, _ptr(proj), _ptr(residual),
            _DTYPES[residual.dtype] if residual is not None else 0, _ptr(g), _ptr(b), float(eps), _ptr(y),
            _DTYPES[out_dtype], ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(rc, "lmufft_linear_ln_bf16")
    return (y, proj) if return_projection else y


def linear_weight_grad_bf16(x: torch.Tensor, dy: torch.Tensor, want_bias: bool = True,
                            out_weight: Optional[torch.Tensor] = None, out_bias: Optional[torch.Tensor] = None,
                            sum_stream: Optional["torch.cuda.Stream"] = None, defer_sum: bool = False):
    """(dW [N, K] fp32, db [N] fp32 or None) of y = x W^T + b from x [..., K] and dy [..., N], both bf16:
    dW = sum over rows of dy^T x, db = sum over rows of dy (HIP kernel k_wgrad_bf16, fp32 accumulation in a
    fixed order).  K and N must be multiples of 8.  `out_weight` / `out_bias`: contiguous float32 tensors to write
    into (e.g. slices of a gradient bucket) instead of new ones."""
    for name, t in (("x", x), ("dy", dy)):
        if not t.is_cuda:
            raise RuntimeError(f"linear_weight_grad_bf16: `{name}` lives on {t.device}; the kernel runs only on a GPU "
                               "(there is no CPU fallback)")
        if t.dtype != torch.bfloat16:
            raise RuntimeError("linear_weight_grad_bf16: x and dy must be bfloat16")
    K, N = x.shape[-1], dy.shape[-1]
    rows = x.numel() // K
    if dy.numel() // N != rows or K % 8 or N % 8:
        raise RuntimeError(f"linear_weight_grad_bf16: x {tuple(x.shape)} and dy {tuple(dy.shape)} must share their rows "
                           "and have widths that are multiples of 8")
    x, dy = x.detach().contiguous(), dy.detach().contiguous()
    for name, t, shape in (("out_weight", out_weight, (N, K)), ("out_bias", out_bias, (N,))):
        if t is not None and (t.shape != shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != x.device):
            raise RuntimeError(f"linear_weight_grad_bf16: `{name}` must be a contiguous float32 tensor {shape} on {x.device}")
    dW = out_weight if out_weight is not None else torch.empty((N, K), dtype=torch.float32, device=x.device)
    db = None
    if want_bias:
        db = out_bias if out_bias is not None else torch.empty((N,), dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib.lib.lmufft_linear_wgrad_workspace_bytes(rows, K, N), dtype=torch.uint8, device=x.device)
    cur = torch.cuda.current_stream(x.device)
    with torch.cuda.device(x.device):
        if defer_sum:
            # the GEMM now; the sum of its row ranges' partial results with every other deferred one, in ONE launch,
            # when flush_weight_grad_sums() is called (GradBuckets does, before it exchanges or hands out gradients)
            rc = _lib.lib.lmufft_linear_wgrad_slabs_bf16(_ptr(x), _ptr(dy), rows, K, N, _ptr(ws), ws.numel(),
                                                         ctypes.c_void_p(cur.cuda_stream))
            _lib.check(rc, "lmufft_linear_wgrad_slabs_bf16")
            _PENDING_SUMS.append((x.device, _lib.WgradSumItem(rows, K, N, dW.data_ptr(), None if db is None else db.data_ptr(),
                                                              ws.data_ptr(), ws.numel()), (ws, dW, db)))
            return dW, db
        if sum_stream is None:
            rc = _lib.lib.lmufft_linear_wgrad_bf16(_ptr(x), _ptr(dy), rows, K, N, _ptr(dW), _ptr(db), _ptr(ws), ws.numel(),
                                                   ctypes.c_void_p(cur.cuda_stream))
            _lib.check(rc, "lmufft_linear_wgrad_bf16")
            return dW, db
        # the GEMM here, the sum of its row ranges' partial results on `sum_stream` (the caller joins that stream
        # before the gradient is read and keeps `ws` alive until then: it is returned as a third value)
        rc = _lib.lib.lmufft_linear_wgrad_slabs_bf16(_ptr(x), _ptr(dy), rows, K, N, _ptr(ws), ws.numel(),
                                                     ctypes.c_void_p(cur.cuda_stream))
        _lib.check(rc, "lmufft_linear_wgrad_slabs_bf16")
        sum_stream.wait_stream(cur)
        rc = _lib.lib.lmufft_linear_wgrad_sum_bf16(rows, K, N, _ptr(dW), _ptr(db), _ptr(ws), ws.numel(),
                                                   ctypes.c_void_p(sum_stream.cuda_stream))
        _lib.check(rc, "lmufft_linear_wgrad_sum_bf16")
    return dW, db, ws


# Weight gradients whose slab sums wait for one common launch: (device, item, tensors kept alive).
_PENDING_SUMS = []
# True: a weight gradient that is written straight into a gradient bucket leaves the sum of its partial slabs to ONE
# launch for all such gradients of the step, made when the bucket's owner asks for the gradients
# (flush_weight_grad_sums; asr_amd.parallel.GradBuckets calls it before it exchanges or hands out gradients).
DEFER_SLAB_SUMS = True


def flush_weight_grad_sums() -> None:
    """Sum the partial slabs of every weight gradient whose sum was deferred, one launch per device, on the current
    stream.  The same numbers as the immediate sums."""
    if not _PENDING_SUMS:
        return
    pending = list(_PENDING_SUMS)
    _PENDING_SUMS.clear()
    for dev in {d for d, _, _ in pending}:
        items = [it for d, it, _ in pending if d == dev]
        arr = (_lib.WgradSumItem * len(items))(*items)
        with torch.cuda.device(dev):
            rc = _lib.lib.lmufft_linear_wgrad_sum_many_bf16(arr, len(items),
                                                            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "lmufft_linear_wgrad_sum_many_bf16")


register_deferred_sums(flush_weight_grad_sums)


# True: weight gradients that are written straight into a gradient bucket run on a side stream, beside the rest of
# the backward (nothing downstream depends on them); asr_amd.parallel.GradBuckets joins the stream before it
# exchanges or hands out gradients.  Off by default; the results are the same numbers either way.
ASYNC_WEIGHT_GRAD = False
# True: only the SUM of a weight gradient's partial slabs runs on that side stream (a short memory-bound kernel with
# no LDS).  Off by default as well: inside the one-graph training step the nine forks and their join cost more than
# the sums they take off the main chain (2.26 -> 2.33 ms per step).
ASYNC_SLAB_SUMS = False


def _weight_grad_into_slots(xb, dyb, has_bias, slot_w, slot_b):
    """linear_weight_grad_bf16 into bucket slices -- on the side stream when ASYNC_WEIGHT_GRAD is on and every
    output is a bucket slice (a gradient handed back to autograd could be accumulated on this stream at once)."""
    from . import functional as F

    into_slots = slot_w is not None and (slot_b is not None or not has_bias)
    if DEFER_SLAB_SUMS and into_slots and not ASYNC_WEIGHT_GRAD and not ASYNC_SLAB_SUMS:
        return linear_weight_grad_bf16(xb, dyb, has_bias, slot_w, slot_b, defer_sum=True)
    if _PENDING_SUMS and not into_slots:
        # this gradient goes back to autograd, which may add it into `.grad` -- a bucket slice whose deferred sum
        # would overwrite the result afterwards: sum first
        flush_weight_grad_sums()
    if not (ASYNC_WEIGHT_GRAD and into_slots):
        if ASYNC_SLAB_SUMS and into_slots:
            dev = xb.device
            side = F._wgrad_stream(dev)
            dW, db, ws = linear_weight_grad_bf16(xb, dyb, has_bias, slot_w, slot_b, sum_stream=side)
            done = torch.cuda.Event()
            done.record(side)
            key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
            F._ASYNC_WGRAD.setdefault(key, []).append((done, (ws, dW, db)))
            return dW, db
        if ASYNC_WEIGHT_GRAD or ASYNC_SLAB_SUMS:
            # this gradient goes back to autograd, which may add it into `.grad` -- a bucket slice that a gradient
            # of the same step may still be writing on the side stream -- on this stream at once
            F.join_async_weight_grads(xb.device)
        return linear_weight_grad_bf16(xb, dyb, has_bias, slot_w, slot_b)

    dev = xb.device
    side = F._wgrad_stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        dW, db = linear_weight_grad_bf16(xb, dyb, has_bias, slot_w, slot_b)
        done = torch.cuda.Event()
        done.record(side)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    F._ASYNC_WGRAD.setdefault(key, []).append((done, (xb, dyb, dW, db)))
    return dW, db


def _prep_items(weights, biases):
    """Output tensors and the kernel's items for the concatenation of the given fp32 weights [N_i, K] (biases: all or
    none): ((w16, w16t, b32), [WprepItem, ...], tensors to keep alive until the launch)."""
    K = weights[0].shape[1]
    N = sum(w.shape[0] for w in weights)
    Np = -(-N // 8) * 8
    dev = weights[0].device
    has_bias = biases is not None and all(b is not None for b in biases)
    w16 = torch.empty((Np, K), dtype=torch.bfloat16, device=dev)
    w16t = torch.empty((K, Np), dtype=torch.bfloat16, device=dev)
    b32 = torch.empty((Np,), dtype=torch.float32, device=dev) if has_bias else None
    items, keep, n0 = [], [], 0
    for i, w in enumerate(weights):
        wd = w.detach().contiguous()
        n = wd.shape[0]
        last = i == len(weights) - 1
        if (n % 8 and not last) or wd.shape[1] != K or wd.dtype != torch.float32:
            raise RuntimeError("linear_weight_prep: float32 weights of one input width; only the last output width "
                               "may be other than a multiple of 8")
        npad = Np - n0 if last else n
        bd = biases[i].detach().contiguous() if has_bias else None
        keep += [wd, bd]
        items.append(_lib.WprepItem(wd.data_ptr(), None if bd is None else bd.data_ptr(), n, K, npad,
                                    w16.data_ptr() + n0 * K * 2, w16t.data_ptr() + n0 * 2, Np,
                                    None if b32 is None else b32.data_ptr() + n0 * 4))
        n0 += n
    return (w16, w16t, b32), items, keep


def _launch_prep(items, dev):
    arr = (_lib.WprepItem * len(items))(*items)
    with torch.cuda.device(dev):
        rc = _lib.lib.lmufft_linear_wprep_many_bf16(arr, len(items),
                                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "lmufft_linear_wprep_many_bf16")


# Operands prepared ahead of the layers' calls by prepare_weights: id(first weight) -> (weights, biases, operands).
# An entry is handed out once, to the call with the very same parameter objects, and prepare_weights starts from an
# empty table, so nothing prepared for one forward pass is seen by another.
_PREPARED = {}


def prepare_weights(groups) -> None:
    """The training operands of several Linear layers in ONE kernel launch (linear_weight_prep launches once per
    weight: fifteen launches in a step of the three-block encoder).  `groups`: [(weights, biases), ...], a group
    being what ONE later linear_train / fused_linear_train call takes -- a single layer, or the layers of a fused
    projection in order.  Call it at the start of a forward pass, before the layers; a layer whose parameters are
    not in the table prepares its own operands as before."""
    _PREPARED.clear()
    items, keep, dev = [], [], None
    for weights, biases in groups:
        weights = list(weights)
        biases = None if biases is None else list(biases)
        out, its, k = _prep_items(weights, biases)
        dev = weights[0].device
        items += its
        keep += k
        _PREPARED[id(weights[0])] = (weights, biases, out)
    if items:
        _launch_prep(items, dev)


def _take_prepared(weights, biases):
    hit = _PREPARED.pop(id(weights[0]), None)
    if hit is None:
        return None
    pw, pb, out = hit
    same = len(pw) == len(weights) and all(a is b for a, b in zip(pw, weights))
    if biases is None or pb is None:
        same = same and (biases is None or all(b is None for b in biases)) and (pb is None or all(b is None for b in pb))
    else:
        same = same and len(pb) == len(biases) and all(a is b for a, b in zip(pb, biases))
    return out if same else None


def linear_weight_prep(weights, biases):
    """(W16 [N_pad, K], W16^T [K, N_pad], bias as autocast sees it [N_pad] fp32 or None) for the concatenation of the
    given fp32 weights [N_i, K] (and biases, all or none): one kernel launch instead of cast, transpose and two bias
    casts per weight -- or none, when prepare_weights has made them already.  N_pad rounds the total output width up
    to a multiple of 8 with zero rows."""
    weights = list(weights)
    ready = _take_prepared(weights, biases)
    if ready is not None:
        return ready
    out, items, keep = _prep_items(weights, biases)
    _launch_prep(items, weights[0].device)
    return out


class _LinearTrain(torch.autograd.Function):
    """nn.Linear under bf16 autocast with this package's kernels in both directions: y = bf16(x) bf16(W)^T + b
    (fp32 accumulation, one rounding); dx = dy bf16(W) in bf16; dW, db in fp32 straight from bf16 x and dy.
    An output width that is not a multiple of 8 (the vocabulary projection) runs on zero rows of padding."""

    @staticmethod
    def forward(ctx, x, weight, bias, keep_padding=False):
        N = weight.shape[0]
        pad = -N % 8
        ctx.keep_padding = bool(keep_padding) and pad > 0
        xb = x.to(torch.bfloat16)
        w16, w16t, b32 = linear_weight_prep([weight], None if bias is None else [bias])
        ctx.save_for_backward(xb, w16t)
        ctx.has_bias = bias is not None
        ctx.x_dtype = x.dtype
        ctx.params = (weight, bias)
        ctx.pad = pad
        y = linear_bf16(xb, w16, b32)
        return y[..., :N] if pad and not ctx.keep_padding else y

    @staticmethod
    def backward(ctx, dy):
        xb, w16t = ctx.saved_tensors
        dyb = dy.to(torch.bfloat16)
        if ctx.pad and not ctx.keep_padding:  # (with keep_padding the gradient arrives with its padding columns)
            dyb = torch.nn.functional.pad(dyb, (0, ctx.pad))
        dyb = dyb.contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = linear_bf16(dyb, w16t).to(ctx.x_dtype)   # dy W: the same kernel on W^T
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            # parameters of a GradBuckets: the kernel writes into the bucket and autograd adopts the slice as .grad
            weight, bias = ctx.params
            slot_w = None if ctx.pad else grad_slot(weight)
            slot_b = grad_slot(bias) if ctx.has_bias and not ctx.pad else None
            dW, db = _weight_grad_into_slots(xb, dyb, ctx.has_bias, slot_w, slot_b)
            if slot_w is not None:
                dW = adopt(dW)
            if slot_b is not None:
                db = adopt(db)
            if ctx.pad:
                n = weight.shape[0]
                dW = dW[:n]
                db = None if db is None else db[:n]
        return dx, dW, db, None


class _FusedLinearTrain(torch.autograd.Function):
    """Several nn.Linear layers on one input as ONE product over their concatenated weights (w_q, w_k, w_v of
    src/models/lmu_encoder.py:41-43, 68-70): y = x [W_1; W_2; ...]^T, one input-gradient GEMM, one weight-gradient
    launch.  When the layers' slices of a gradient bucket lie back to back (weights in this order, then biases in
    this order), the weight gradient is written straight into them."""

    @staticmethod
    def forward(ctx, x, *params):
        weights, biases = params[0::2], params[1::2]
        xb = x.to(torch.bfloat16)
        w16, w16t, b32 = linear_weight_prep(list(weights), list(biases))
        ctx.save_for_backward(xb, w16t)
        ctx.x_dtype = x.dtype
        ctx.params = (weights, biases)
        return linear_bf16(xb, w16, b32)

    @staticmethod
    def backward(ctx, dy):
        xb, w16t = ctx.saved_tensors
        weights, biases = ctx.params
        dyb = dy.to(torch.bfloat16).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_bf16(dyb, w16t).to(ctx.x_dtype)
        fw = fused_slot(weights)
        fb = fused_slot(biases)
        dW, db = _weight_grad_into_slots(xb, dyb, True, None if fw is None else fw[0], None if fb is None else fb[0])
        rows = [w.shape[0] for w in weights]
        gw = [adopt(s) for s in fw[1]] if fw is not None else list(torch.split(dW, rows))
        gb = [adopt(s) for s in fb[1]] if fb is not None else list(torch.split(db, rows))
        out = [dx]
        for a, b in zip(gw, gb):
            out += [a, b]
        return tuple(out)


def fused_linear_train(x: torch.Tensor, layers) -> torch.Tensor:
    """[layer(x) for layer in layers] concatenated along the last dimension, as bf16 autocast computes them, with
    autograd, on HIP kernels.  Every layer needs a bias and the same input width."""
    layers = list(layers)
    if not layers or any(l.bias is None for l in layers) or not all(train_supported(x, l.weight) for l in layers) \
            or any(l.weight.shape[0] % 8 for l in layers):
        raise RuntimeError("fused_linear_train: needs GPU tensors and float32 Linear layers with biases, one input width "
                           "(a multiple of 8) and output widths that are multiples of 8")
    params = []
    for l in layers:
        params += [l.weight, l.bias]
    return _FusedLinearTrain.apply(x, *params)


def train_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """True when linear_train can take this call (GPU, fp32 master weight, input width a multiple of 8)."""
    return (x.is_cuda and weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2
            and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] == weight.shape[1]
            and weight.shape[1] % 8 == 0)


def linear_train(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                 keep_padding: bool = False) -> torch.Tensor:
    """torch.nn.functional.linear(x, weight, bias) as bf16 autocast computes it, with autograd, on HIP kernels.
    `keep_padding`: an output width that is not a multiple of 8 comes back with the kernel's zero-weight padding
    columns (width rounded up to 8) instead of a slice, and its gradient is expected with them (zeros there): the
    vocabulary projection hands both to log_softmax_rows without a slice forward and a pad backward."""
    if not train_supported(x, weight):
        raise RuntimeError("linear_train: needs GPU tensors, a float32 weight [N, K] with K a multiple of 8 and a "
                           "float32 or bfloat16 input [..., K]")
    return _LinearTrain.apply(x, weight, bias, keep_padding)
