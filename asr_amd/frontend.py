"""Log-mel frontend on the GPU (HIP kernels of liblmufft_hip.so).

Host-side mirror of the reference's AudioPreprocessor feature extraction
(src/data/preprocessing.py:10-51, 76-117): same constructor arguments, same
`extract_mel_features(waveform) -> [T, n_mels]`, plus a batched form that takes
raw 16 kHz audio already resident in HBM and returns the zero-padded
[B, T, 80] batch the reference's collate_fn would build on the host.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class MelFrontend:
    def __init__(self, sample_rate: int = 16000, n_mels: int = 80, n_fft: int = 400, hop_length: int = 160,
                 win_length: int = 400, normalize: bool = True):
        if (sample_rate, n_mels, n_fft, hop_length, win_length) != (16000, 80, 400, 160, 400):
            raise ValueError(
                "MelFrontend implements the reference's configuration only: sample_rate=16000, n_mels=80, "
                "n_fft=400, hop_length=160, win_length=400")
        self.sample_rate, self.n_mels, self.n_fft = sample_rate, n_mels, n_fft
        self.hop_length, self.win_length, self.normalize = hop_length, win_length, normalize
        self._prep = {}  # device -> constants

    def _constants(self, device):
        buf = self._prep.get(device)
        if buf is None:
            buf = torch.empty(_lib.lib.lmufft_melprep_bytes(), dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                rc = _lib.lib.lmufft_melprep(ctypes.c_void_p(buf.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
                _lib.check(rc, "lmufft_melprep")
            self._prep[device] = buf
        return buf

    def num_frames(self, n_samples: int) -> int:
        return int(_lib.lib.lmufft_mel_frames(int(n_samples)))

    def __call__(self, wave: torch.Tensor, lengths: torch.Tensor | None = None):
        """wave [B, L] float32 on the GPU, lengths [B] (samples per utterance, optional)
        -> (mel [B, 1 + L // 160, 80] float32, frames [B] int32)."""
        if not wave.is_cuda:
            raise RuntimeError(f"MelFrontend: `wave` lives on {wave.device}; the frontend runs only as HIP kernels "
                               "on a GPU (there is no CPU fallback)")
        if wave.dim() != 2 or wave.dtype != torch.float32:
            raise RuntimeError(f"MelFrontend: wave must be float32 [batch, samples], got {wave.dtype} "
                               f"{tuple(wave.shape)}")
        wave = wave.contiguous()
        B, L = wave.shape
        dev = wave.device
        if lengths is not None:
            given_on_gpu = lengths.is_cuda
            host_lengths = lengths
            lengths = lengths.to(device=dev, dtype=torch.int32).contiguous()
            if lengths.shape != (B,):
                raise RuntimeError(f"MelFrontend: lengths must have shape ({B},), got {tuple(lengths.shape)}")
            if not given_on_gpu:
                # lengths that came from the host are checked here; lengths already on the GPU are the
                # caller's to keep in range (no device-to-host synchronisation on this path)
                lo, hi = int(host_lengths.min()), int(host_lengths.max())
                if lo <= self.n_fft // 2 or hi > L:
                    raise RuntimeError(
                        f"MelFrontend: every utterance needs more than {self.n_fft // 2} samples (the reflect padding "
                        f"of the STFT, as torchaudio requires) and at most {L}; got lengths in [{lo}, {hi}]")
        T = self.num_frames(L)
        mel = torch.empty((B, T, self.n_mels), dtype=torch.float32, device=dev)
        prep = self._constants(dev)
        with torch.cuda.device(dev):
            rc = _lib.lib.lmufft_mel_forward_f32(
                ctypes.c_void_p(wave.data_ptr()), None if lengths is None else ctypes.c_void_p(lengths.data_ptr()),
                B, L, L, ctypes.c_void_p(mel.data_ptr()), T, int(self.normalize), ctypes.c_void_p(prep.data_ptr()),
                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "lmufft_mel_forward_f32")
        if lengths is None:
            frames = torch.full((B,), T, dtype=torch.int32, device=dev)
        else:
            # (the kernel clamps a length to [1, L]; the frame counts follow the same clamp)
            frames = 1 + torch.div(lengths.clamp(1, L), self.hop_length, rounding_mode="floor")
        return mel, frames

    def extract_mel_features(self, waveform: torch.Tensor) -> torch.Tensor:
        """waveform [L] -> [T, n_mels] (reference: src/data/preprocessing.py:76-99)."""
        mel, _ = self(waveform.reshape(1, -1))
        return mel[0]
