"""CPU port of the reference's LMUFFT arithmetic with torch ops -- TEST / CPU-BASELINE
INFRASTRUCTURE, NOT A PRODUCT PATH.

It exists for one purpose: to be *timed* on the host cores as the
`cpu_baseline` ("port") next to the GPU number in bench.py, and to cross-check
the numpy oracle.  It follows the reference's route exactly -- Linear+ReLU,
rfft(n=2T) x rfft(H) -> irfft, slice, cat, Linear+ReLU (src/lmu.py:297-320) --
so its cost is the cost of the reference's CPU path; gradients come from
torch.autograd like the reference's.
"""
from __future__ import annotations

import torch


class PortLMUFFT(torch.nn.Module):
    def __init__(self, input_size, hidden_size, memory_size, H: torch.Tensor):
        super().__init__()
        self.W_u = torch.nn.Linear(input_size, 1)
        self.W_h = torch.nn.Linear(memory_size + input_size, hidden_size)
        T = H.shape[1]
        self.register_buffer("H", H.clone())
        self.register_buffer("fft_H", torch.fft.rfft(H, n=2 * T, dim=-1))

    def forward(self, x):
        T = x.shape[1]
        u = torch.relu(self.W_u(x))  # [B,T,1]                      src/lmu.py:300
        fu = torch.fft.rfft(u.permute(0, 2, 1), n=2 * T, dim=-1)  # src/lmu.py:303-304
        m = torch.fft.irfft(fu * self.fft_H.unsqueeze(0), n=2 * T, dim=-1)[:, :, :T]  # :308-311
        m = m.permute(0, 2, 1)  # [B,T,M]
        h = torch.relu(self.W_h(torch.cat((m, x), dim=-1)))  # src/lmu.py:315-316
        return h, h[:, -1, :]
