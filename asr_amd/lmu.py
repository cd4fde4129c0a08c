"""Drop-in module surface of the reference's `src/lmu.py` for the FFT-LMU path.

`LMUFFT` keeps the reference's constructor signature, attribute / child /
buffer names and `forward(x) -> (h, h_n)` contract (src/lmu.py:208-320), so a
caller such as the reference's `LMUEncoder` (src/models/lmu_encoder.py:204-212,
:266) and its checkpoints (`model_state_dict` keys `A, B, H, fft_H,
W_u.{weight,bias}, W_h.{weight,bias}`) work unchanged.  The arithmetic runs in
hand-written gfx950 kernels (asr_amd/csrc) -- there is no eager or CPU path.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .functional import (ReluMaskLink, fft_supported, lmufft_layer, prepare_impulse_bf16, prepare_impulse_fft,
                         prepare_scan_bf16, scan_supported)

__all__ = ["LMUFFT", "LMU", "legendre_state_space", "impulse_response", "fuse_relu_backward"]


def legendre_state_space(memory_size: int, theta: float):
    """Discretised (zero-order hold, dt = 1) Legendre delay system A [M,M], B [M,1].

    Same construction as the reference (src/lmu.py:248-272): continuous
    A_c[i,j] = (2i+1)/theta * (-1 if i<j else (-1)^(i-j+1)),
    B_c[i] = (2i+1)/theta * (-1)^i, discretised by the same third-party routine
    (scipy.signal.cont2discrete, method "zoh"), then cast to float32.
    """
    from scipy.signal import cont2discrete

    q = np.arange(memory_size, dtype=np.float64)
    scale = ((2.0 * q + 1.0) / float(theta))[:, None]
    row, col = np.meshgrid(q, q, indexing="ij")
    a_c = scale * np.where(row < col, -1.0, (-1.0) ** (row - col + 1.0))
    b_c = scale * ((-1.0) ** q)[:, None]
    a_d, b_d, _, _, _ = cont2discrete(
        (a_c, b_c, np.ones((1, memory_size)), np.zeros((1,))), dt=1.0, method="zoh")
    return torch.from_numpy(a_d).float(), torch.from_numpy(b_d).float()


def impulse_response(A: torch.Tensor, B: torch.Tensor, seq_len: int):
    """H [M, seq_len] with H[:, t] = A^t B and its spectrum rfft(H, 2*seq_len).

    The powers are accumulated in float32 on the CPU in the reference's order
    (`A_t <- A @ A_t`, src/lmu.py:278-285): an fp64-exact A^t B differs from
    that by ~8e-6 relative, which would eat into the 1e-4 parity budget.
    """
    M = A.shape[0]
    H = torch.empty((M, seq_len), dtype=torch.float32)
    a_t = torch.eye(M, dtype=torch.float32)
    A = A.float().cpu()
    B = B.float().cpu()
    for t in range(seq_len):
        H[:, t : t + 1] = a_t @ B
        a_t = A @ a_t
    fft_H = torch.fft.rfft(H, n=2 * seq_len, dim=-1)
    return H, fft_H


def _AUTO_PREFERS_FFT(T, M):
    """Shape policy of conv_algo="auto" on the fp32 path, from kernel timings on MI355X (tools/fft_bench.py: batch 64,
    192 memory channels, forward + adjoint, event-bracketed kernels of a route, us): T=100 Toeplitz 36 / FFT 38,
    T=128 37 / 39, T=160 46 / 46, T=200 52 / 48, T=256 55 / 52, T=300 77 / 64, T=400 99 / 70, T=512 115 / 83;
    batch 32, T=3000 (partitioned FFT) 1970 / 620.  The Toeplitz kernels' work grows with T^2, the FFT's with the
    transform length (256 / 512 / 1024 points) and, beyond one transform, with the number of partition products."""
    return T >= 160


# bf16 path: sequence length from which conv_algo="auto" takes the state-space ("scan") route.  Its work per
# frame is constant (M * (M + 128) multiply-adds) where the Toeplitz kernels' grows with T / 2; measured on
# MI355X at batch 32, 192 memory channels, forward + adjoint, Toeplitz / scan: T=1024 51 / 72 us,
# T=1536 76 / 77, T=2048 111 / 83, T=3000 215 / 104.
_SCAN_MIN_T = 1600

# (device, memory size, bytes of A and B) -> constants of the scan route; layers with the same (A, B) share them
_SCANPREP_CACHE = {}


class LMUFFT(nn.Module):
    """Parallel (convolutional) Legendre Memory Unit layer on MI355X.

    Parameters follow the reference (src/lmu.py:225): input_size, hidden_size,
    memory_size, seq_len, theta.  `forward` accepts any T <= seq_len (the
    reference requires T == seq_len; where it runs, results are identical).

    `fft_H` is kept as a buffer only so that state_dicts stay interchangeable;
    the kernels read `H`.  `conv_algo` ("auto" | "toeplitz" | "fft" | "scan") selects the
    causal-convolution kernel and is not part of the reference signature ("fft": fp32 only;
    "scan": bf16 only, evaluates the same convolution from the buffers A and B in chunks).
    """

    def __init__(self, input_size, hidden_size, memory_size, seq_len, theta, conv_algo="auto"):
        super().__init__()
        self.hidden_size = hidden_size
        self.memory_size = memory_size
        self.seq_len = seq_len
        self.theta = theta
        self.conv_algo = conv_algo

        self.W_u = nn.Linear(in_features=input_size, out_features=1)
        self.f_u = nn.ReLU()
        self.W_h = nn.Linear(in_features=memory_size + input_size, out_features=hidden_size)
        self.f_h = nn.ReLU()

        A, B = legendre_state_space(memory_size, theta)
        self.register_buffer("A", A)  # [memory_size, memory_size]
        self.register_buffer("B", B)  # [memory_size, 1]
        H, fft_H = impulse_response(A, B, seq_len)
        self.register_buffer("H", H)  # [memory_size, seq_len]
        self.register_buffer("fft_H", fft_H)  # [memory_size, seq_len + 1] complex64
        self._hprep = None  # (key, bf16 copies of H), rebuilt when H changes
        self._fftprep = {}  # T -> (key, spectra of H[:, :T]) for the FFT route
        self.grad_sink = None  # set by asr_amd.parallel.GradBuckets: gradients land in its flat buffer
        # bf16 path, with a grad_sink: run the parameter gradients on a side stream, beside the backward of the
        # layers below (GradBuckets joins them before it exchanges or hands out gradients)
        self.async_weight_grad = False
        # set in pairs by fuse_relu_backward(lower, upper) for two directly stacked layers (bf16 path)
        self.input_is_relu_output = False  # the input gradient is returned already multiplied by (x > 0)
        self.grad_is_masked = False        # the gradient of h arrives already zero wherever h <= 0
        self._mask_out = None              # ReluMaskLink to the layer stacked on this one
        self._mask_in = None               # ReluMaskLink from the layer below

    _FFTPREP_MAX = 8  # cached FFT-route spectra (one per distinct sequence length)

    def _impulse_bf16(self):
        H = self.H
        key = (H.data_ptr(), H._version, H.device, H.dtype)
        if self._hprep is None or self._hprep[0] != key:
            self._hprep = (key, prepare_impulse_bf16(H))
        return self._hprep[1]

    def _scan_operands(self):
        A, B = self.A, self.B
        key = (A.device, A.shape[0], A._version, B._version, A.data_ptr(), B.data_ptr())
        if getattr(self, "_scanprep", None) is None or self._scanprep[0] != key:
            if A.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("LMUFFT: the state-space convolution route reads A and B on the host at its first "
                                   "call; run the layer once at this sequence length before capturing a HIP graph")
            content = (str(A.device), A.shape[0], A.detach().float().cpu().numpy().tobytes(),
                       B.detach().float().cpu().numpy().tobytes())
            hit = _SCANPREP_CACHE.get(content)
            if hit is None:
                if len(_SCANPREP_CACHE) >= 4:
                    _SCANPREP_CACHE.pop(next(iter(_SCANPREP_CACHE)))
                hit = prepare_scan_bf16(A, B)
                _SCANPREP_CACHE[content] = hit
            self._scanprep = (key, hit)
        return self._scanprep[1]

    def _impulse_fft(self, T):
        H = self.H
        key = (H.data_ptr(), H._version, H.device, H.dtype)
        hit = self._fftprep.get(T)
        if hit is None or hit[0] != key:
            # spectra of other lengths made from an older H are stale too; and a layer fed batches of
            # many different lengths keeps the most recent few only (each entry is ~M * N * 8 bytes)
            for t in [t for t, v in self._fftprep.items() if v[0] != key]:
                del self._fftprep[t]
            while len(self._fftprep) >= self._FFTPREP_MAX:
                del self._fftprep[next(iter(self._fftprep))]
            hit = (key, prepare_impulse_fft(H, T))
        else:
            del self._fftprep[T]  # re-insert: most recently used last
        self._fftprep[T] = hit
        return hit[1]

    def forward(self, x):
        """x [batch, seq_len, input_size] -> (h [batch, seq_len, hidden], h_n [batch, hidden]).

        float32 input runs the exact-fp32 kernels.  bfloat16 input, or any input
        under `torch.autocast("cuda", dtype=torch.bfloat16)`, runs the bf16
        kernels (fp32 accumulation, fp32 master parameters) and returns bf16.
        """
        if torch.is_autocast_enabled() and x.is_cuda and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            x = x.to(torch.bfloat16)
        elif x.dtype == torch.float16:
            # fp16 autocast is what the reference's trainer runs under (src/training/distributed.py:66,143);
            # there torch.fft takes the layer to fp32 anyway.  Same here: fp16 activations are widened
            # and the exact-fp32 kernels run (the result is fp32, as an autocast-to-fp32 op returns).
            x = x.float()
        hprep = self._impulse_bf16() if (x.is_cuda and x.dtype == torch.bfloat16) else None
        scanprep = None
        if hprep is not None and x.dim() == 3 and x.shape[1] <= self.H.shape[1]:
            if self.conv_algo == "scan" or (self.conv_algo == "auto" and x.shape[1] >= _SCAN_MIN_T
                                            and scan_supported(self.memory_size)):
                scanprep = self._scan_operands()
        fftprep = None
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 3:
            T = x.shape[1]
            if T > self.H.shape[1]:
                pass  # rejected with the layer's own message below
            elif self.conv_algo == "fft" or (self.conv_algo == "auto" and _AUTO_PREFERS_FFT(T, self.memory_size)
                                           and fft_supported(T)):
                fftprep = self._impulse_fft(T)
        return lmufft_layer(x, self.W_u.weight, self.W_u.bias, self.W_h.weight, self.W_h.bias, self.H,
                            conv_algo=self.conv_algo, hprep=hprep, fftprep=fftprep,
                            grad_sink=self.grad_sink, scanprep=scanprep,
                            async_weight_grad=self.async_weight_grad,
                            relu_input=self.input_is_relu_output, grad_masked=self.grad_is_masked,
                            mask_link=self._mask_link())


    def _mask_link(self):
        # one argument carries both directions: a layer in the middle of a stack takes the mask of its x from the
        # link below and offers the mask of its h to the link above
        lo, up = self._mask_in if self.input_is_relu_output else None, self._mask_out if self.grad_is_masked else None
        if lo is None and up is None:
            return None
        return _BothLinks(lo, up)

    # hidden sizes whose rows fit one projection tile: the layer norm can ride on the projection kernel
    FUSED_LAYER_NORM_MAX_HIDDEN = 384

    def supports_fused_layer_norm(self, x) -> bool:
        """True when forward_layer_norm(x, ...) can run: inference on the bf16 path, shapes the
        LDS-DMA projection kernel takes, the whole hidden row inside one tile."""
        bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled()
                                             and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        return (x.is_cuda and x.dim() == 3 and bf16 and not torch.is_grad_enabled()
                and self.hidden_size <= self.FUSED_LAYER_NORM_MAX_HIDDEN and self.hidden_size % 8 == 0
                and self.memory_size % 8 == 0 and x.shape[-1] % 8 == 0)

    def forward_layer_norm(self, x, layer_norm: nn.LayerNorm, out_dtype=torch.float32):
        """(layer_norm(h), h_n): the layer followed by the encoder's LayerNorm
        (src/models/lmu_encoder.py:266-274) in one pass; see supports_fused_layer_norm."""
        from .functional import lmufft_layer_norm_inference

        return lmufft_layer_norm_inference(x.to(torch.bfloat16), self.W_u.weight, self.W_u.bias, self.W_h.weight,
                                           self.W_h.bias, self.H, layer_norm.weight, layer_norm.bias, layer_norm.eps,
                                           out_dtype, hprep=self._impulse_bf16())


class _BothLinks:
    """take() from the link below a layer, offer() to the link above it."""

    def __init__(self, below, above):
        self.below, self.above = below, above

    def take(self, x):
        return self.below.take(x) if self.below is not None else None

    def offer(self, h, bits) -> None:
        if self.above is not None:
            self.above.offer(h, bits)


def fuse_relu_backward(lower: "LMUFFT", upper: "LMUFFT", enable: bool = True) -> None:
    """For two DIRECTLY stacked layers -- `upper(lower(x)[0])` with nothing in between and no other consumer of
    `lower`'s h -- fold the ReLU backward of `lower`'s output (autograd of src/lmu.py:318) into `upper`'s
    input-gradient kernel.

    `upper` then returns its input gradient already multiplied by (x > 0), and `lower` takes the gradient of h as the
    masked gradient without a pass of its own (one kernel and ~20 MB of HBM traffic less per fused pair at batch
    64 x 400 frames).  bf16 path only; on the fp32 path both layers ignore the setting.  The parameter gradients and
    the gradient that leaves `lower` are the same numbers either way; the tensor that travels between the two layers
    is no longer the plain gradient of h, so do not read it (`retain_grad`, hooks) while the fusion is on.  `lower`
    falls back to its own masking pass whenever the gradient of its last-frame output h_n is in use."""
    if not isinstance(lower, LMUFFT) or not isinstance(upper, LMUFFT):
        raise TypeError("fuse_relu_backward: both layers must be asr_amd.LMUFFT modules")
    d_in = upper.W_u.in_features
    if enable and (d_in != lower.hidden_size or d_in % 8 or upper.hidden_size % 8):
        raise ValueError(f"fuse_relu_backward: the upper layer's input size ({d_in}) must equal the lower layer's hidden "
                         f"size ({lower.hidden_size}), and it and the upper hidden size must be multiples of 8")
    upper.input_is_relu_output = bool(enable)
    lower.grad_is_masked = bool(enable)
    link = ReluMaskLink() if enable else None
    lower._mask_out = link
    upper._mask_in = link


class LMU(nn.Module):
    """Name kept because the reference encoder imports it (src/models/lmu_encoder.py:9,15).

    The sequential (recurrent) LMU is outside the scope of this library.
    """

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(
            "asr_amd implements the parallel LMUFFT layer only; construct the model with use_fft_lmu=True")
