"""torch.autograd binding of the HIP LMUFFT layer.

PyTorch supplies device memory, the current HIP stream and autograd graph
bookkeeping; all arithmetic runs in liblmufft_hip.so through its C ABI.
"""
from __future__ import annotations

import ctypes
import weakref

import torch

from . import _lib
from .parallel import register_deferred_sums

# True: the bf16 backward always forms the memory gradient gm = gz W_h[:, :M] and runs the adjoint convolution
# on it (the reference's own order of operations); False: where the library can, it takes the encoder gradient
# straight from gz (include/lmufft.h, lmufft_backward_needs_gm_bf16).  Tests compare the two.
FORCE_MEMORY_GRADIENT = False
# bf16: the forward also writes h's relu pattern as one bit per element and the backward masks the upstream
# gradient with it instead of reading h back (tests switch it off to compare the two)
USE_RELU_MASK = True

_ALGOS = {"auto": _lib.CONV_AUTO, "toeplitz": _lib.CONV_TOEPLITZ, "fft": _lib.CONV_FFT, "scan": _lib.CONV_SCAN}
_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


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


def prepare_impulse_bf16(H: torch.Tensor) -> torch.Tensor:
    """bf16 copies of H in both orientations, as the bf16 kernels read them.

    H is a constant buffer of the module, so callers build this once per H and
    pass it to `lmufft_layer(..., hprep=...)`.
    """
    _require_gpu(H, "H")
    H = H.contiguous().float()
    M, ldH = H.shape
    shp = _shape(1, 1, 1, M, 1, ldH)
    buf = _workspace(_lib.lib.lmufft_hprep_bytes(ctypes.byref(shp)), H.device)
    with torch.cuda.device(H.device):
        rc = _lib.lib.lmufft_hprep_bf16(ctypes.byref(shp), _ptr(H), _ptr(buf), _stream(H.device))
    _lib.check(rc, "lmufft_hprep_bf16")
    return buf


def scan_supported(M: int) -> bool:
    """True when the state-space ("scan") convolution route of the bf16 path takes memory size M."""
    return _lib.lib.lmufft_scanprep_bytes(int(M)) != 0


def prepare_scan_bf16(A: torch.Tensor, B: torch.Tensor, device=None) -> torch.Tensor:
    """Constants of the state-space convolution route (powers of A, the chunk's Toeplitz block, the
    chunk-to-chunk transition), built once per (A, B) by the library on the host and copied to `device`.

    A [M, M] and B [M, 1] are the module's buffers (src/lmu.py:240-241); H = [A^t B] is what the route
    reproduces."""
    import numpy as np

    M = A.shape[0]
    nbytes = _lib.lib.lmufft_scanprep_bytes(int(M))
    if nbytes == 0:
        raise RuntimeError(f"LMUFFT: the scan convolution route does not support memory_size {M} "
                           "(multiples of 8 up to 256)")
    device = A.device if device is None else device
    a = np.ascontiguousarray(A.detach().float().cpu().numpy())
    b = np.ascontiguousarray(B.detach().float().cpu().numpy().reshape(-1))
    host = np.empty(nbytes, dtype=np.uint8)
    rc = _lib.lib.lmufft_scanprep_build_host(a.ctypes.data, b.ctypes.data, int(M), host.ctypes.data)
    _lib.check(rc, "lmufft_scanprep_build_host")
    return torch.from_numpy(host).to(device)


def fft_supported(T: int, M: int = 1) -> bool:
    """True when the LDS-resident FFT convolution kernels can take sequence length T."""
    shp = _shape(1, T, 1, M, 1, T)
    return _lib.lib.lmufft_fftprep_bytes(ctypes.byref(shp)) != 0


def prepare_impulse_fft(H: torch.Tensor, T: int) -> torch.Tensor:
    """Spectra of the first T taps of H for the FFT route (counterpart of the
    reference's `fft_H` buffer, src/lmu.py:285).  Built once per (H, T)."""
    _require_gpu(H, "H")
    H = H.contiguous().float()
    M, ldH = H.shape
    shp = _shape(1, T, 1, M, 1, ldH)
    nbytes = _lib.lib.lmufft_fftprep_bytes(ctypes.byref(shp))
    if nbytes == 0:
        raise RuntimeError(f"LMUFFT: the FFT convolution kernels do not support seq_len {T} (limit 16384)")
    buf = _workspace(nbytes, H.device)
    with torch.cuda.device(H.device):
        rc = _lib.lib.lmufft_fftprep_f32(ctypes.byref(shp), _ptr(H), _ptr(buf), _stream(H.device))
    _lib.check(rc, "lmufft_fftprep_f32")
    return buf


# Weight gradients running on the side stream: per device, (event recorded after them, tensors they read or
# write -- kept alive until the join).  See _LMUFFTFunction.backward and join_async_weight_grads.
_ASYNC_WGRAD = {}
_WGRAD_STREAMS = {}


def _wgrad_stream(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _WGRAD_STREAMS.get(key)
    if st is None:
        st = _WGRAD_STREAMS[key] = torch.cuda.Stream(device)
    return st


# LMUFFT layers whose weight-gradient slabs wait for one common sum: (device, item, tensors kept alive).
_PENDING_WEIGHT_SUMS = []
# True: a bf16 backward that writes its parameter gradients straight into a gradient bucket leaves the sum of the
# weight-gradient kernel's partial slabs to ONE launch for all such layers of the step, made when the bucket's owner
# asks for the gradients (flush_weight_sums; asr_amd.parallel.GradBuckets calls it before it exchanges or hands out
# gradients).  The same numbers as the immediate sums.
DEFER_WEIGHT_SUMS = True


def flush_weight_sums() -> None:
    """Sum the deferred weight-gradient slabs, one launch per device, on the current stream."""
    if not _PENDING_WEIGHT_SUMS:
        return
    pending = list(_PENDING_WEIGHT_SUMS)
    _PENDING_WEIGHT_SUMS.clear()
    for dev in {d for d, _, _ in pending}:
        items = [it for d, it, _ in pending if d == dev]
        arr = (_lib.LmuSumItem * len(items))(*items)
        with torch.cuda.device(dev):
            rc = _lib.lib.lmufft_backward_weights_sum_many_bf16(arr, len(items), _stream(dev))
        _lib.check(rc, "lmufft_backward_weights_sum_many_bf16")


register_deferred_sums(flush_weight_sums)


def join_async_weight_grads(device=None) -> None:
    """Make the current stream of `device` (default: every device with work outstanding) wait for the weight
    gradients that layers with `async_weight_grad` launched on the side stream.  GradBuckets calls this before it
    exchanges or hands out gradients; call it yourself before reading `.grad` of such a layer otherwise."""
    for key in list(_ASYNC_WGRAD):
        if device is not None:
            dkey = (device.type, device.index if device.index is not None else torch.cuda.current_device())
            if key != dkey:
                continue
        pending = _ASYNC_WGRAD.pop(key)
        cur = torch.cuda.current_stream(torch.device(*key))
        for done, _keep in pending:
            cur.wait_event(done)


class GradSink:
    """Destination of one layer's parameter gradients inside a flat exchange buffer.

    The first backward of a step writes gw_u, gb_u, gW_h, gb_h straight into
    these tensors (no autograd accumulation kernels) and calls `on_ready`;
    further backwards of the same step return their gradients to autograd,
    which adds them into `.grad` -- by then the same tensors.
    """

    def __init__(self, params, views, on_ready, on_late=None):
        self.params = list(params)  # W_u.weight, W_u.bias, W_h.weight, W_h.bias
        self.views = list(views)    # fp32, contiguous, 16-byte aligned, same shapes
        self.on_ready = on_ready
        self.on_late = on_late      # called when a further backward of the same step reaches the layer
        self.filled = False

    def usable(self, needs) -> bool:
        return not self.filled and all(needs) and all(p.grad is None for p in self.params)

    def commit(self) -> None:
        self.filled = True
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.on_ready(len(self.params))


class _QEntry:
    """The weights-only matrix Q^T of one layer's backward (sequences up to 512 frames), waiting to be made: by the
    masking pass of another layer's backward, where its workgroups are free, or by one launch for all that wait."""

    __slots__ = ("item", "keep", "stream", "done", "__weakref__")

    def __init__(self, shp, hprep, wprep, qt, stream):
        self.item = _lib.QItem(shp, hprep.data_ptr(), wprep.data_ptr(), qt.data_ptr())
        self.keep = (hprep, wprep, qt)
        self.stream = stream
        self.done = False


# weak references to the _QEntry objects of forwards whose backward has not run yet, per device; an entry lives as
# long as its autograd node does
_PENDING_Q = {}
# True: layers whose gradient arrives masked (grad_masked) get their Q^T made off their own backward's critical path
SHARE_QPREP = True


def _waiting_q(dev, stream, but=None):
    alive = [r for r in _PENDING_Q.get(dev, []) if r() is not None and not r().done]
    _PENDING_Q[dev] = alive
    return [r() for r in alive if r() is not but and r().stream == stream]


class ReluMaskLink:
    """Hands the bit mask of (h > 0) that a layer's forward makes from that layer to the one stacked on it.

    offer(h, bits) is called by the lower layer's forward, take(x) by the upper layer's: it returns the bits only if x
    is that h (same memory, same shape -- a detached alias counts), so a stale or foreign mask is never used; without
    bits the upper layer compares x itself."""

    def __init__(self):
        self._key = None
        self._bits = None

    @staticmethod
    def _key_of(t):
        return (t.device, t.data_ptr(), tuple(t.shape), t.dtype)

    def offer(self, h, bits) -> None:
        self._key, self._bits = (self._key_of(h), bits) if bits is not None else (None, None)

    def take(self, x):
        if self._bits is None or self._key != self._key_of(x) or not x.is_contiguous():
            return None
        return self._bits


class _LMUFFTFunction(torch.autograd.Function):
    """(x, w_u, b_u, W_h, b_h, H) -> (h, h_n); reference: src/lmu.py:290-320."""

    @staticmethod
    def forward(ctx, x, w_u, b_u, W_h, b_h, H, conv_algo, hprep, fftprep, grad_sink, scanprep=None,
                async_wgrad=False, relu_input=False, grad_masked=False, mask_link=None):
        _require_gpu(x, "x")
        if x.dtype not in _DTYPES:
            raise RuntimeError(f"LMUFFT: x has dtype {x.dtype}; supported: float32, bfloat16")
        for name, t in (("W_u.weight", w_u), ("W_u.bias", b_u), ("W_h.weight", W_h), ("W_h.bias", b_h), ("H", H)):
            _require_gpu(t, name)
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
        dt = x.dtype
        ctx.param_dtypes = (w_u.dtype, b_u.dtype, W_h.dtype, b_h.dtype)
        # parameters and H are consumed in fp32 (master weights); no-ops for fp32 modules
        x = x.contiguous()
        w_u = w_u.contiguous().float()
        b_u = b_u.contiguous().float()
        W_h = W_h.contiguous().float()
        b_h = b_h.contiguous().float()
        H = H.contiguous().float()
        dev = x.device
        u = torch.empty((B, T), dtype=torch.float32, device=dev)
        m = torch.empty((B, T, M), dtype=dt, device=dev)
        h = torch.empty((B, T, Hd), dtype=dt, device=dev)
        h_n = torch.empty((B, Hd), dtype=dt, device=dev)
        shp = _shape(B, T, D, M, Hd, ldH)
        hmask = None
        with torch.cuda.device(dev):
            ws = _workspace(_lib.lib.lmufft_forward_workspace_bytes(ctypes.byref(shp), _DTYPES[dt]), dev)
            if dt == torch.float32:
                if fftprep is None and conv_algo == _lib.CONV_FFT:
                    fftprep = prepare_impulse_fft(H, T)
                rc = _lib.lib.lmufft_forward_f32(
                    ctypes.byref(shp), _ptr(x), _ptr(w_u), _ptr(b_u), _ptr(W_h), _ptr(b_h), _ptr(H), _ptr(fftprep),
                    _ptr(u), _ptr(m), _ptr(h), _ptr(h_n), conv_algo, _ptr(ws), ws.numel(), _stream(dev))
                what = "lmufft_forward_f32"
            else:
                if hprep is None:
                    hprep = prepare_impulse_bf16(H)
                # bf16 copies of W_h (both orientations), made by the forward's first launch and reused
                # by the backward
                wprep = _workspace(_lib.lib.lmufft_wprep_bytes(ctypes.byref(shp)), dev)
                # the relu pattern of h as a bit mask: the backward reads it instead of h (1/16 of the bytes)
                if USE_RELU_MASK and Hd % 8 == 0 and any(ctx.needs_input_grad[:5]):
                    hmask = torch.empty((B * T, Hd // 8), dtype=torch.uint8, device=dev)
                rc = _lib.lib.lmufft_forward_bf16(
                    ctypes.byref(shp), _ptr(x), _ptr(w_u), _ptr(b_u), _ptr(wprep), _ptr(W_h), _ptr(b_h), _ptr(hprep),
                    _ptr(scanprep), _ptr(u), _ptr(m), _ptr(h), _ptr(h_n), _ptr(hmask), conv_algo, _ptr(ws), ws.numel(),
                    _stream(dev))
                what = "lmufft_forward_bf16"
        _lib.check(rc, what)
        ctx.save_for_backward(x, w_u, W_h, H, u, m, h, hprep, wprep if dt != torch.float32 else None,
                              fftprep if dt == torch.float32 else None, scanprep if dt != torch.float32 else None, hmask)
        ctx.conv_algo = conv_algo
        ctx.async_wgrad = bool(async_wgrad)
        ctx.relu_input = bool(relu_input)
        ctx.grad_masked = bool(grad_masked)
        # (x > 0) as bits, if the layer that produced x left them for this very tensor (ReluMaskLink)
        ctx.x_mask = mask_link.take(x) if (mask_link is not None and relu_input) else None
        if mask_link is not None and grad_masked:
            mask_link.offer(h, hmask)
        ctx.q_entry = None
        if (SHARE_QPREP and grad_masked and dt == torch.bfloat16 and hmask is not None and scanprep is None
                and not FORCE_MEMORY_GRADIENT):
            qt_bytes = _lib.lib.lmufft_qt_bytes(ctypes.byref(shp))
            if qt_bytes:
                ctx.q_entry = _QEntry(shp, hprep, wprep, torch.empty(qt_bytes, dtype=torch.uint8, device=dev),
                                      torch.cuda.current_stream(dev).cuda_stream)
                _PENDING_Q.setdefault(dev, []).append(weakref.ref(ctx.q_entry))
        ctx.grad_sink = grad_sink
        ctx.shape = (B, T, D, M, Hd, ldH)
        ctx.set_materialize_grads(False)
        return h, h_n

    @staticmethod
    def backward(ctx, gh, gh_n):
        x, w_u, W_h, H, u, m, h, hprep, wprep, fftprep, scanprep, hmask = ctx.saved_tensors
        B, T, D, M, Hd, ldH = ctx.shape
        dev = x.device
        dt = x.dtype
        need_x = ctx.needs_input_grad[0]
        if gh is None and gh_n is None:
            return (None,) * 15
        if gh is not None:
            gh = gh.to(dt).contiguous()
        if gh_n is not None:
            gh_n = gh_n.to(dt).contiguous()
        gx = torch.empty_like(x) if need_x else None
        sink = ctx.grad_sink
        if sink is not None and sink.filled:
            if sink.on_late is not None:
                sink.on_late()  # raises when the bucket has already been handed to the all-reduce
            # A further backward of the same step: autograd will ADD the gradients returned below into
            # `.grad` -- the sink's views -- on this stream.  If the first backward's weight gradient is
            # still running on the side stream it is writing those same views: wait for it first.
            join_async_weight_grads(dev)
            flush_weight_sums()  # (a deferred sum would overwrite what autograd is about to add)
        direct = sink is not None and sink.usable(ctx.needs_input_grad[1:5])
        if direct:
            gw_u, gb_u, gW_h, gb_h = sink.views
        else:
            gw_u = torch.empty_like(w_u)
            gb_u = torch.empty((1,), dtype=torch.float32, device=dev)
            gW_h = torch.empty_like(W_h)
            gb_h = torch.empty((Hd,), dtype=torch.float32, device=dev)
        shp = _shape(B, T, D, M, Hd, ldH)
        pending = None
        # the bf16 backward forms the memory gradient only on the routes that need it (long sequences, odd sizes)
        gm = None
        if dt == torch.float32 or FORCE_MEMORY_GRADIENT or _lib.lib.lmufft_backward_needs_gm_bf16(
                ctypes.byref(shp), ctx.conv_algo, 0 if scanprep is None else 1):
            gm = torch.empty_like(m)
        with torch.cuda.device(dev):
            ws = _workspace(_lib.lib.lmufft_backward_workspace_bytes(ctypes.byref(shp), _DTYPES[dt]), dev)
            if dt == torch.float32:
                rc = _lib.lib.lmufft_backward_f32(
                    ctypes.byref(shp), _ptr(x), _ptr(w_u), _ptr(W_h), _ptr(H), _ptr(fftprep), _ptr(u), _ptr(m),
                    _ptr(h), _ptr(gh), _ptr(gh_n), _ptr(gx), _ptr(gw_u), _ptr(gb_u), _ptr(gW_h), _ptr(gb_h),
                    _ptr(gm), ctx.conv_algo, _ptr(ws), ws.numel(), _stream(dev))
                what = "lmufft_backward_f32"
            elif ctx.async_wgrad and direct and not (ctx.relu_input and need_x):
                # The parameter gradients depend on nothing after gx and nothing below this layer depends on
                # them: they run on a side stream, beside the backward of the layers below, and are joined by
                # whoever consumes the gradient bucket (join_async_weight_grads).  Only with a gradient sink:
                # gradients handed back to autograd are accumulated on this stream at once.
                rc = _lib.lib.lmufft_backward_data_bf16(
                    ctypes.byref(shp), _ptr(x), _ptr(w_u), _ptr(wprep), _ptr(hprep), _ptr(scanprep), _ptr(u), _ptr(m),
                    _ptr(h), _ptr(hmask), _ptr(gh), _ptr(gh_n), _ptr(gx), _ptr(gm), ctx.conv_algo, _ptr(ws), ws.numel(),
                    _stream(dev))
                _lib.check(rc, "lmufft_backward_data_bf16")
                side = _wgrad_stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    rc = _lib.lib.lmufft_backward_weights_bf16(
                        ctypes.byref(shp), _ptr(x), _ptr(m), _ptr(gw_u), _ptr(gb_u), _ptr(gW_h), _ptr(gb_h), _ptr(ws),
                        ws.numel(), _stream(dev))
                    done = torch.cuda.Event()
                    done.record(side)
                key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
                _ASYNC_WGRAD.setdefault(key, []).append((done, (x, m, ws, gw_u, gb_u, gW_h, gb_h)))
                what = "lmufft_backward_weights_bf16"
            else:
                flags = 0
                if ctx.relu_input and need_x:
                    # x is a ReLU's output and that ReLU's backward comes next: gx leaves already multiplied by (x > 0).
                    # The layer below may rely on it (grad_masked), so this is never skipped silently: shapes the
                    # kernel does not take are an error of the C call.
                    flags |= _lib.BWD_MASK_GX
                if ctx.grad_masked and gh is not None and gh_n is None and gh.data_ptr() % 16 == 0:
                    # gh is zero wherever h <= 0 already (the layer above ran with relu_input on this h): it is
                    # used as the masked gradient as it is.  Otherwise the masking pass runs, which is always right.
                    flags |= _lib.BWD_GH_MASKED
                if DEFER_WEIGHT_SUMS and direct:
                    flags |= _lib.BWD_PARTIAL_SUMS
                    pending = (dev, _lib.LmuSumItem(shp, gw_u.data_ptr(), gb_u.data_ptr(), gW_h.data_ptr(),
                                                    gb_h.data_ptr(), ws.data_ptr(), ws.numel()),
                               (ws, gw_u, gb_u, gW_h, gb_h))
                extras = _lib.BwdExtras(None if ctx.x_mask is None else ctx.x_mask.data_ptr(), None, None, 0)
                own = ctx.q_entry
                cur = torch.cuda.current_stream(dev).cuda_stream
                riders = []
                if own is not None and (flags & _lib.BWD_GH_MASKED) and gm is None and own.stream == cur:
                    if not own.done:
                        # nobody has made this layer's Q^T yet: one launch now, for every layer that waits for one
                        batch = [own] + _waiting_q(dev, cur, but=own)
                        arr = (_lib.QItem * len(batch))(*[e.item for e in batch])
                        _lib.check(_lib.lib.lmufft_qprep_many_bf16(arr, len(batch), _stream(dev)),
                                   "lmufft_qprep_many_bf16")
                        for e in batch:
                            e.done = True
                    extras.qt = own.item.qt
                else:
                    if own is not None:
                        own.done = True  # (this backward makes its own, in its workspace)
                    if not (flags & _lib.BWD_GH_MASKED) and gm is None:
                        # this backward runs a masking pass: the waiting layers' Q^T ride on its spare workgroups
                        riders = _waiting_q(dev, cur)
                        if riders:
                            arr = (_lib.QItem * len(riders))(*[e.item for e in riders])
                            extras.q_items, extras.q_count = arr, len(riders)
                rc = _lib.lib.lmufft_backward_flags_bf16(
                    ctypes.byref(shp), _ptr(x), _ptr(w_u), _ptr(wprep), _ptr(hprep), _ptr(scanprep), _ptr(u), _ptr(m),
                    _ptr(h), _ptr(hmask), _ptr(gh), _ptr(gh_n), _ptr(gx), _ptr(gw_u), _ptr(gb_u), _ptr(gW_h), _ptr(gb_h),
                    _ptr(gm), ctx.conv_algo, _ptr(ws), ws.numel(), _stream(dev), flags, ctypes.byref(extras))
                what = "lmufft_backward_flags_bf16"
                if rc == 0:
                    for e in riders:
                        e.done = True
        _lib.check(rc, what)
        if pending is not None:
            _PENDING_WEIGHT_SUMS.append(pending)  # (before commit(): the bucket's owner may ask for the sums there)
        if direct:
            sink.commit()
            return (gx,) + (None,) * 14
        pd = ctx.param_dtypes
        return (gx, gw_u.to(pd[0]), gb_u.to(pd[1]), gW_h.to(pd[2]), gb_h.to(pd[3])) + (None,) * 10


def lmufft_layer(x, w_u, b_u, W_h, b_h, H, conv_algo="auto", hprep=None, fftprep=None, grad_sink=None,
                 scanprep=None, async_weight_grad=False, relu_input=False, grad_masked=False, mask_link=None):
    """h, h_n = LMUFFT forward with autograd support (HIP kernels only).

    x float32 -> exact-fp32 kernels; x bfloat16 -> bf16 kernels with fp32
    accumulation (`hprep` = prepare_impulse_bf16(H), built on the fly if None).
    `fftprep` = prepare_impulse_fft(H, T) enables the FFT convolution of the
    fp32 path (required for conv_algo="fft", built on the fly if None).
    `grad_sink` (a GradSink) receives the parameter gradients in place.
    `scanprep` = prepare_scan_bf16(A, B) makes the bf16 path take the state-space convolution route
    (conv_algo "auto" or "scan"; required for "scan").
    `async_weight_grad` (bf16 path, with a grad_sink only): the parameter gradients are launched on a side stream
    and joined by the sink's owner (join_async_weight_grads).
    `relu_input`, `grad_masked` (bf16 path; set in pairs, see asr_amd.lmu.fuse_relu_backward): `relu_input` -- x is
    the output of a ReLU whose backward is the next thing to happen to x's gradient, so that gradient is returned
    already multiplied by (x > 0); `grad_masked` -- the gradient of h arrives already zero wherever h <= 0, the
    layer's own masking pass is skipped.  `mask_link` (a ReluMaskLink shared by the two layers of a pair): the
    lower layer leaves the bit mask of (h > 0) that its forward makes anyway, the upper layer uses it in place of
    reading x when its x is that very h.
    """
    return _LMUFFTFunction.apply(x, w_u, b_u, W_h, b_h, H, _ALGOS[conv_algo], hprep, fftprep, grad_sink, scanprep,
                                 async_weight_grad, relu_input, grad_masked, mask_link)


def lmufft_layer_norm_inference(x, w_u, b_u, W_h, b_h, H, ln_weight, ln_bias, eps=1e-5, out_dtype=torch.float32,
                                hprep=None):
    """(layer_norm(h), h_n) of the bf16 LMUFFT forward, the normalisation done in the projection
    kernel's epilogue (no autograd; h itself is not written).  x bfloat16 [B, T, D]; needs
    hidden_size <= 384 and memory / input / hidden sizes that are multiples of 8.
    Reference: src/lmu.py:297-320 followed by src/models/lmu_encoder.py:274."""
    _require_gpu(x, "x")
    if x.dtype != torch.bfloat16 or x.dim() != 3:
        raise RuntimeError("lmufft_layer_norm_inference: x must be a bfloat16 [batch, seq_len, input_size] tensor")
    if out_dtype not in _DTYPES:
        raise RuntimeError(f"lmufft_layer_norm_inference: output dtype {out_dtype} is not supported")
    B, T, D = x.shape
    M, ldH = H.shape
    Hd = W_h.shape[0]
    if w_u.shape != (1, D) or W_h.shape[1] != M + D or ln_weight.shape != (Hd,) or ln_bias.shape != (Hd,):
        raise RuntimeError("lmufft_layer_norm_inference: parameter shapes do not match x")
    if T > ldH:
        raise RuntimeError(f"LMUFFT: sequence length {T} exceeds the layer's seq_len {ldH}")
    dev = x.device
    x = x.detach().contiguous()
    f32 = lambda t: t.detach().contiguous().float()  # noqa: E731
    w_u, b_u, W_h, b_h, H, g, b = f32(w_u), f32(b_u), f32(W_h), f32(b_h), f32(H), f32(ln_weight), f32(ln_bias)
    if hprep is None:
        hprep = prepare_impulse_bf16(H)
    u = torch.empty((B, T), dtype=torch.float32, device=dev)
    m = torch.empty((B, T, M), dtype=torch.bfloat16, device=dev)
    h_n = torch.empty((B, Hd), dtype=torch.bfloat16, device=dev)
    y = torch.empty((B, T, Hd), dtype=out_dtype, device=dev)
    shp = _shape(B, T, D, M, Hd, ldH)
    with torch.cuda.device(dev):
        wprep = _workspace(_lib.lib.lmufft_wprep_bytes(ctypes.byref(shp)), dev)
        rc = _lib.lib.lmufft_forward_ln_bf16(
            ctypes.byref(shp), _ptr(x), _ptr(w_u), _ptr(b_u), _ptr(wprep), _ptr(W_h), _ptr(b_h), _ptr(hprep), None,
            _ptr(u), _ptr(m), None, _ptr(h_n), _ptr(g), _ptr(b), float(eps), _ptr(y), _DTYPES[out_dtype],
            _lib.CONV_AUTO, None, 0, _stream(dev))
    _lib.check(rc, "lmufft_forward_ln_bf16")
    return y, h_n


def _conv_fftprep(H, T, conv_algo, fftprep):
    if fftprep is None and conv_algo == "fft":
        fftprep = prepare_impulse_fft(H, T)
    return fftprep


def causal_conv(u, H, conv_algo="auto", fftprep=None):
    """m[b,t,k] = sum_{s<=t} H[k,t-s] u[b,s]  (no autograd; kernel-level entry point)."""
    _require_gpu(u, "u")
    _require_gpu(H, "H")
    u = u.contiguous()
    H = H.contiguous()
    B, T = u.shape
    M, ldH = H.shape
    m = torch.empty((B, T, M), dtype=torch.float32, device=u.device)
    shp = _shape(B, T, 1, M, 1, ldH)
    fftprep = _conv_fftprep(H, T, conv_algo, fftprep)
    algo = _ALGOS[conv_algo]
    with torch.cuda.device(u.device):
        ws = _workspace(_lib.lib.lmufft_conv_workspace_bytes(ctypes.byref(shp), _lib.F32, algo), u.device)
        rc = _lib.lib.lmufft_conv_forward_f32(ctypes.byref(shp), _ptr(u), _ptr(H), _ptr(fftprep), _ptr(m), algo,
                                              _ptr(ws), ws.numel(), _stream(u.device))
    _lib.check(rc, "lmufft_conv_forward_f32")
    return m


def causal_conv_adjoint(gm, H, conv_algo="auto", fftprep=None):
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
    fftprep = _conv_fftprep(H, T, conv_algo, fftprep)
    with torch.cuda.device(gm.device):
        ws = _workspace(_lib.lib.lmufft_conv_workspace_bytes(ctypes.byref(shp), _lib.F32, algo), gm.device)
        rc = _lib.lib.lmufft_conv_backward_f32(ctypes.byref(shp), _ptr(gm), _ptr(H), _ptr(fftprep), _ptr(gu), algo,
                                               _ptr(ws), ws.numel(), _stream(gm.device))
    _lib.check(rc, "lmufft_conv_backward_f32")
    return gu


def causal_conv_bf16(u, H, hprep=None):
    """m = causal convolution of u [B, T] (fp32) with H, written as bf16 [B, T, M] by the bf16
    kernels (no autograd; kernel-level entry point, src/lmu.py:303-312)."""
    _require_gpu(u, "u")
    _require_gpu(H, "H")
    u = u.contiguous().float()
    B, T = u.shape
    M, ldH = H.shape
    if hprep is None:
        hprep = prepare_impulse_bf16(H)
    m = torch.empty((B, T, M), dtype=torch.bfloat16, device=u.device)
    shp = _shape(B, T, 1, M, 1, ldH)
    with torch.cuda.device(u.device):
        rc = _lib.lib.lmufft_conv_forward_bf16(ctypes.byref(shp), _ptr(u), _ptr(hprep), _ptr(m), _stream(u.device))
    _lib.check(rc, "lmufft_conv_forward_bf16")
    return m


def causal_conv_adjoint_bf16(gm, H, hprep=None, workspace=None, out=None):
    """gu[b,s] = sum_{t>=s} sum_k H[k,t-s] gm[b,t,k] for bf16 gm, fp32 result (no autograd)."""
    _require_gpu(gm, "gm")
    _require_gpu(H, "H")
    if gm.dtype != torch.bfloat16:
        raise RuntimeError("causal_conv_adjoint_bf16: gm must be bfloat16")
    gm = gm.contiguous()
    B, T, M = gm.shape
    ldH = H.shape[1]
    if hprep is None:
        hprep = prepare_impulse_bf16(H)
    gu = out if out is not None else torch.empty((B, T), dtype=torch.float32, device=gm.device)
    shp = _shape(B, T, 1, M, 1, ldH)
    with torch.cuda.device(gm.device):
        if workspace is None:
            workspace = _workspace(_lib.lib.lmufft_conv_workspace_bytes(ctypes.byref(shp), _lib.BF16, _lib.CONV_AUTO),
                                   gm.device)
        rc = _lib.lib.lmufft_conv_backward_bf16(ctypes.byref(shp), _ptr(gm), _ptr(hprep), _ptr(gu), _ptr(workspace),
                                                workspace.numel(), _stream(gm.device))
    _lib.check(rc, "lmufft_conv_backward_bf16")
    return gu


def causal_conv_scan_bf16(u, scanprep, M, workspace=None):
    """m = causal convolution of u [B, T] (fp32) with H = [A^t B], written as bf16 [B, T, M] by the
    state-space route (no autograd; kernel-level entry point, src/lmu.py:303-312)."""
    _require_gpu(u, "u")
    u = u.contiguous().float()
    B, T = u.shape
    m = torch.empty((B, T, M), dtype=torch.bfloat16, device=u.device)
    shp = _shape(B, T, 1, M, 1, T)
    with torch.cuda.device(u.device):
        if workspace is None:
            workspace = _workspace(_lib.lib.lmufft_conv_scan_workspace_bytes(ctypes.byref(shp)), u.device)
        rc = _lib.lib.lmufft_conv_forward_scan_bf16(ctypes.byref(shp), _ptr(u), _ptr(scanprep), _ptr(m),
                                                   _ptr(workspace), workspace.numel(), _stream(u.device))
    _lib.check(rc, "lmufft_conv_forward_scan_bf16")
    return m


def causal_conv_adjoint_scan_bf16(gm, scanprep, workspace=None):
    """gu[b,s] = sum_{t>=s} sum_k H[k,t-s] gm[b,t,k] for bf16 gm by the state-space route (fp32 result)."""
    _require_gpu(gm, "gm")
    if gm.dtype != torch.bfloat16:
        raise RuntimeError("causal_conv_adjoint_scan_bf16: gm must be bfloat16")
    gm = gm.contiguous()
    B, T, M = gm.shape
    gu = torch.empty((B, T), dtype=torch.float32, device=gm.device)
    shp = _shape(B, T, 1, M, 1, T)
    with torch.cuda.device(gm.device):
        if workspace is None:
            workspace = _workspace(_lib.lib.lmufft_conv_scan_workspace_bytes(ctypes.byref(shp)), gm.device)
        rc = _lib.lib.lmufft_conv_backward_scan_bf16(ctypes.byref(shp), _ptr(gm), _ptr(scanprep), _ptr(gu),
                                                    _ptr(workspace), workspace.numel(), _stream(gm.device))
    _lib.check(rc, "lmufft_conv_backward_scan_bf16")
    return gu


def conv_pair_runner(u, H, conv_algo, dtype, scanprep=None, split=False):
    """A callable that runs the causal convolution of u and the adjoint of its result through the
    C ABI's conv-only entry points with every buffer preallocated (benchmarks).  `split` (bf16 Toeplitz and fp32
    routes): returns (forward, adjoint) callables instead -- the adjoint then reads a gradient tensor of its own and
    has a workspace of its own, so the two may run on different streams."""
    dev = u.device
    B, T = u.shape
    M, ldH = H.shape
    shp = _shape(B, T, 1, M, 1, ldH)
    if dtype == "bf16" and conv_algo == "scan":
        m = torch.empty((B, T, M), dtype=torch.bfloat16, device=dev)
        gu = torch.empty((B, T), dtype=torch.float32, device=dev)
        ws = _workspace(_lib.lib.lmufft_conv_scan_workspace_bytes(ctypes.byref(shp)), dev)

        def run():
            st = _stream(dev)
            _lib.check(_lib.lib.lmufft_conv_forward_scan_bf16(ctypes.byref(shp), _ptr(u), _ptr(scanprep), _ptr(m),
                                                             _ptr(ws), ws.numel(), st), "lmufft_conv_forward_scan_bf16")
            _lib.check(_lib.lib.lmufft_conv_backward_scan_bf16(ctypes.byref(shp), _ptr(m), _ptr(scanprep), _ptr(gu),
                                                              _ptr(ws), ws.numel(), st), "lmufft_conv_backward_scan_bf16")
            return m, gu

        return run
    if dtype == "bf16":
        hprep = prepare_impulse_bf16(H)
        m = torch.empty((B, T, M), dtype=torch.bfloat16, device=dev)
        gu = torch.empty((B, T), dtype=torch.float32, device=dev)
        ws = _workspace(_lib.lib.lmufft_conv_workspace_bytes(ctypes.byref(shp), _lib.BF16, _lib.CONV_AUTO), dev)

        gm = torch.randn((B, T, M), device=dev).to(torch.bfloat16) if split else m

        def fwd():
            st = _stream(dev)  # the stream current at the call (a capturing stream under graph capture)
            _lib.check(_lib.lib.lmufft_conv_forward_bf16(ctypes.byref(shp), _ptr(u), _ptr(hprep), _ptr(m), st),
                       "lmufft_conv_forward_bf16")

        def adj():
            _lib.check(_lib.lib.lmufft_conv_backward_bf16(ctypes.byref(shp), _ptr(gm), _ptr(hprep), _ptr(gu), _ptr(ws),
                                                         ws.numel(), _stream(dev)), "lmufft_conv_backward_bf16")

        def run():
            fwd()
            adj()
            return m, gu

        return (fwd, adj) if split else run
    algo = _ALGOS[conv_algo]
    fftprep = prepare_impulse_fft(H, T) if conv_algo == "fft" else None
    m = torch.empty((B, T, M), dtype=torch.float32, device=dev)
    gu = torch.empty((B, T), dtype=torch.float32, device=dev)
    ws = _workspace(_lib.lib.lmufft_conv_workspace_bytes(ctypes.byref(shp), _lib.F32, algo), dev)

    gm = torch.randn((B, T, M), device=dev) if split else m
    ws2 = _workspace(ws.numel(), dev) if split else ws

    def fwd():
        _lib.check(_lib.lib.lmufft_conv_forward_f32(ctypes.byref(shp), _ptr(u), _ptr(H), _ptr(fftprep), _ptr(m), algo,
                                                   _ptr(ws), ws.numel(), _stream(dev)), "lmufft_conv_forward_f32")

    def adj():
        _lib.check(_lib.lib.lmufft_conv_backward_f32(ctypes.byref(shp), _ptr(gm), _ptr(H), _ptr(fftprep), _ptr(gu), algo,
                                                    _ptr(ws2), ws2.numel(), _stream(dev)), "lmufft_conv_backward_f32")

    def run():
        fwd()
        adj()
        return m, gu

    return (fwd, adj) if split else run
