"""CTC head for inference on the GPU: projection to the vocabulary, log-softmax and greedy decode.

Host-side mirror of the reference's CTCDecoder (src/models/asr_model.py:13-50: a module holding
`ctc_projection = nn.Linear(input_size, vocab_size)`, forward -> log-probabilities) and of
LMUASRModel.decode with beam_width=1 (src/models/asr_model.py:158-213), which there is a Python
loop with one device-to-host `.item()` per frame.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


class CTCDecoder(nn.Module):
    def __init__(self, input_size: int, vocab_size: int):
        super().__init__()
        self.input_size = input_size
        self.vocab_size = vocab_size
        self.ctc_projection = nn.Linear(input_size, vocab_size)

    def _run(self, enc: torch.Tensor, lengths: Optional[torch.Tensor], want_log_probs: bool):
        if not enc.is_cuda:
            raise RuntimeError(f"CTCDecoder: encoder output lives on {enc.device}; the head runs only as HIP "
                               "kernels on a GPU (there is no CPU fallback)")
        if enc.dim() != 3 or enc.shape[-1] != self.input_size:
            raise RuntimeError(f"CTCDecoder: expected [batch, frames, {self.input_size}], got {tuple(enc.shape)}")
        if enc.dtype not in _DTYPES:
            raise RuntimeError(f"CTCDecoder: dtype {enc.dtype}; supported: float32, bfloat16")
        enc = enc.detach().contiguous()
        B, T, Hd = enc.shape
        dev = enc.device
        V = self.vocab_size
        W = self.ctc_projection.weight.detach().float().contiguous()
        b = self.ctc_projection.bias.detach().float().contiguous()
        if lengths is not None:
            lengths = lengths.to(device=dev, dtype=torch.int32).contiguous()
        lp = torch.empty((B, T, V), dtype=torch.float32, device=dev) if want_log_probs else None
        tokens = torch.empty((B, T), dtype=torch.int32, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        ws = torch.empty(max(_lib.lib.lmufft_ctc_workspace_bytes(B, T), 256), dtype=torch.uint8, device=dev)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(dev):
            rc = _lib.lib.lmufft_ctc_greedy(p(enc), _DTYPES[enc.dtype], p(W), p(b), p(lengths), B, T, Hd, V, V - 1,
                                            p(lp), p(tokens), p(counts), p(ws), ws.numel(),
                                            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "lmufft_ctc_greedy")
        return lp, tokens, counts

    def forward(self, encoder_output: torch.Tensor) -> torch.Tensor:
        """[batch, frames, input_size] -> log-probabilities [batch, frames, vocab_size] (no autograd)."""
        return self._run(encoder_output, None, True)[0]

    def greedy(self, encoder_output: torch.Tensor, input_lengths: Optional[torch.Tensor] = None
               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Device-side result: tokens [batch, frames] int32 (row b: counts[b] tokens, then -1), counts [batch]."""
        _, tokens, counts = self._run(encoder_output, input_lengths, False)
        return tokens, counts

    def decode(self, encoder_output: torch.Tensor, input_lengths: Optional[torch.Tensor] = None) -> List[List[int]]:
        """Greedy CTC decode to Python lists (one host transfer for the whole batch)."""
        tokens, counts = self.greedy(encoder_output, input_lengths)
        tokens, counts = tokens.cpu(), counts.cpu()
        return [tokens[i, : int(counts[i])].tolist() for i in range(tokens.shape[0])]


def pack_targets(targets: torch.Tensor, target_lengths: torch.Tensor, total: Optional[int] = None):
    """Padded labels [batch, max_len] -> (all labels back to back, offsets [batch + 1]), on the device.

    What LMUASRModel.compute_loss builds before torch's CTC loss (src/models/asr_model.py:145-151) with one
    `.tolist()` per utterance.  `total` = sum of the lengths when the caller knows it (e.g. from the data loader);
    otherwise it is read back from the device once, because the packed tensor's size must equal it.
    """
    if not targets.is_cuda:
        raise RuntimeError(f"pack_targets: targets live on {targets.device}; this runs only as a HIP kernel on a GPU "
                           "(there is no CPU fallback)")
    if targets.dim() != 2:
        raise RuntimeError(f"pack_targets: expected [batch, max_len], got {tuple(targets.shape)}")
    dev = targets.device
    B, L = targets.shape
    if target_lengths.numel() != B:
        raise RuntimeError(f"pack_targets: {target_lengths.numel()} lengths for {B} utterances")
    if total is not None:
        total = int(total)
        if total < 0 or total > B * L:
            raise RuntimeError(f"pack_targets: total = {total} labels cannot come from {B} utterances of at most {L}")
        if not target_lengths.is_cuda:
            have = int(target_lengths.clamp(min=0, max=L).sum())
            if have != total:
                raise RuntimeError(f"pack_targets: total = {total}, but the lengths (clamped to max_len = {L}) sum to {have}")
    t = targets.to(torch.int64)
    if t.stride(1) != 1:
        t = t.contiguous()
    lengths = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
    packed = torch.empty(B * L, dtype=torch.int64, device=dev)
    offsets = torch.empty(B + 1, dtype=torch.int32, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
    with torch.cuda.device(dev):
        rc = _lib.lib.lmufft_ctc_pack_targets(p(t), p(lengths), B, L, t.stride(0), p(packed), p(offsets),
                                              ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "lmufft_ctc_pack_targets")
    if total is None:
        total = int(offsets[B].item())
    return packed[:total], offsets


class _CTCLoss(torch.autograd.Function):
    """Loss and gradient come out of one kernel launch; the backward only scales the stored gradient."""

    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank, zero_infinity):
        B, T, V = log_probs.shape
        dev = log_probs.device
        lp = log_probs.detach().float().contiguous()
        Lmax = targets.shape[1]
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        grad = torch.empty((B, T, V), dtype=torch.float32, device=dev)
        ws = torch.empty(max(_lib.lib.lmufft_ctc_loss_workspace_bytes(B, T, Lmax), 16), dtype=torch.uint8, device=dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(dev):
            rc = _lib.lib.lmufft_ctc_loss(p(lp), p(targets), targets.stride(0), p(input_lengths), p(target_lengths), B, T, V,
                                          Lmax, blank, 1, 1 if zero_infinity else 0, p(nll), p(grad), p(ws), ws.numel(),
                                          ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(rc, "lmufft_ctc_loss")
            # mean over the batch of nll / max(length, 1), infinite terms zeroed: one launch
            loss = torch.empty((), dtype=torch.float32, device=dev)
            rc = _lib.lib.lmufft_ctc_mean_loss(p(nll), p(target_lengths), B, 1 if zero_infinity else 0, p(loss),
                                               ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "lmufft_ctc_mean_loss")
        ctx.save_for_backward(grad)
        ctx.in_dtype = log_probs.dtype
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (grad * gout).to(ctx.in_dtype), None, None, None, None, None


class _LogSoftmaxRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, V):
        x = logits.detach()
        if x.stride(-1) != 1 or not x.is_contiguous():
            x = x.contiguous()
        ld = x.shape[-1]
        rows = x.numel() // ld
        lp = torch.empty((*x.shape[:-1], V), dtype=torch.float32, device=x.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(x.device):
            rc = _lib.lib.lmufft_log_softmax_rows(p(x), 1 if x.dtype == torch.bfloat16 else 0, ld, rows, V, p(lp),
                                                  ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        _lib.check(rc, "lmufft_log_softmax_rows")
        ctx.save_for_backward(lp)
        ctx.ld, ctx.in_dtype = ld, logits.dtype
        return lp

    @staticmethod
    def backward(ctx, dlp):
        (lp,) = ctx.saved_tensors
        V = lp.shape[-1]
        g = dlp.to(torch.float32).contiguous()
        rows = lp.numel() // V
        dx = torch.empty((*lp.shape[:-1], ctx.ld), dtype=ctx.in_dtype, device=lp.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(lp.device):
            rc = _lib.lib.lmufft_log_softmax_rows_bwd(p(g), p(lp), rows, V, p(dx), 1 if dx.dtype == torch.bfloat16 else 0,
                                                      ctx.ld, ctypes.c_void_p(torch.cuda.current_stream(lp.device).cuda_stream))
        _lib.check(rc, "lmufft_log_softmax_rows_bwd")
        return dx, None


def log_softmax_rows(logits: torch.Tensor, vocab_size: int) -> torch.Tensor:
    """torch.log_softmax(logits[..., :vocab_size].float(), dim=-1) with autograd, as one HIP kernel each way: the
    logits may carry padding columns behind the vocabulary (asr_amd.linear.linear_train(..., keep_padding=True)),
    which the forward ignores and the backward fills with zeros.  float32 or bfloat16 logits of at most 64 columns;
    the result is float32 [..., vocab_size]."""
    if not logits.is_cuda:
        raise RuntimeError(f"log_softmax_rows: logits live on {logits.device}; this runs only as a HIP kernel on a GPU "
                           "(there is no CPU fallback)")
    if logits.dtype not in (torch.float32, torch.bfloat16) or not (0 < vocab_size <= logits.shape[-1] <= 64):
        raise RuntimeError(f"log_softmax_rows: float32 or bfloat16 logits with vocab_size <= width <= 64 "
                           f"(got {logits.dtype}, width {logits.shape[-1]}, vocab_size {vocab_size})")
    return _LogSoftmaxRows.apply(logits, int(vocab_size))


def ctc_loss(log_probs: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor, target_lengths: torch.Tensor,
             blank: int, zero_infinity: bool = True) -> torch.Tensor:
    """Mean-reduced CTC loss of log_probs [batch, frames, vocab] (batch first, as the model produces them) against
    the padded labels targets [batch, max_len]: what nn.CTCLoss(blank, reduction='mean', zero_infinity) returns for
    log_probs.transpose(0, 1) in the reference (src/models/asr_model.py:105, 144-154), with autograd.  HIP kernel
    k_ctc_loss (loss and gradient in one launch, no target packing, lengths stay on the device).

    Range: the recursions run in the linear domain on doubles, rescaled every fourth frame by the frame's largest
    state.  Log-probabilities down to about -170 per frame are exact; states more than e^-700 below the largest state
    of their frame are flushed to zero.  An utterance all of whose alignments run through such states -- e.g. L labels
    whose log-probabilities average below -700 / L on the way while the blank stays near 1 -- gets an infinite term,
    which zero_infinity zeroes, where torch's log-domain kernel returns a large finite loss."""
    if not log_probs.is_cuda:
        raise RuntimeError(f"ctc_loss: log_probs live on {log_probs.device}; this runs only as a HIP kernel on a GPU "
                           "(there is no CPU fallback)")
    if log_probs.dim() != 3 or targets.dim() != 2 or targets.shape[0] != log_probs.shape[0]:
        raise RuntimeError(f"ctc_loss: expected log_probs [batch, frames, vocab] and targets [batch, max_len], got "
                           f"{tuple(log_probs.shape)} and {tuple(targets.shape)}")
    B, T, V = log_probs.shape
    if V > 64:
        raise RuntimeError(f"ctc_loss: vocabulary of {V} entries; the kernel holds at most 64")
    dev = log_probs.device
    t = targets.to(device=dev, dtype=torch.int64)
    if t.shape[1] == 0:
        t = torch.zeros((B, 1), dtype=torch.int64, device=dev)
    if t.stride(1) != 1:
        t = t.contiguous()
    il = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
    return _CTCLoss.apply(log_probs, t, il, tl, int(blank), bool(zero_infinity))
