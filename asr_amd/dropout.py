"""Dropout masks for training steps replayed from HIP graphs.

The dropout kernels (attention weights, layer norm + dropout) decide every element by a hash of its
position and the call's two seeds.  The seeds are by-value kernel arguments: a captured graph freezes
them, and a replayed step would draw the same masks for ever, where the reference draws fresh ones at
every call (nn.Dropout, src/models/lmu_encoder.py:45, 86, 232).  The library therefore adds a 32-bit
counter read from DEVICE memory to the seeds when the kernels run (C ABI lmufft_dropout_epoch); the
captured step increments the counter as its first node, so every replay sees a new value, and the
forward and backward kernels of one step see the same one.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

_EPOCHS = {}  # device index -> int32 tensor [1], registered with the library for that device


def _index(device) -> int:
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def dropout_epoch(device) -> torch.Tensor:
    """The device's epoch counter, created and registered with the library on first use.  Must first be
    called outside a graph capture (its allocation and zero-fill must not become graph nodes)."""
    idx = _index(device)
    t = _EPOCHS.get(idx)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("asr_amd: a dropout step is being captured into a HIP graph before the device's dropout "
                               "epoch counter exists; call asr_amd.dropout.dropout_epoch(device) before the capture "
                               "(GradBuckets.capture_* do) and increment it inside the captured step (bump_dropout_epoch)")
        dev = torch.device("cuda", idx)
        t = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.lmufft_dropout_epoch(ctypes.c_void_p(t.data_ptr())), "lmufft_dropout_epoch")
        _EPOCHS[idx] = t
    return t


def epoch_registered(device) -> bool:
    return _index(device) in _EPOCHS


def bump_dropout_epoch(device) -> None:
    """Increment the device's counter on the current stream (inside a capture: a node of the graph)."""
    dropout_epoch(device).add_(1)


def check_capturable(device, p_drop: float, seeds_given: bool) -> None:
    """Raises when a dropout call with default seeds is being captured and no epoch counter is registered:
    the replayed step would reuse one mask for ever."""
    if p_drop > 0 and not seeds_given and torch.cuda.is_current_stream_capturing() and not epoch_registered(device):
        dropout_epoch(device)  # raises with the explanation
