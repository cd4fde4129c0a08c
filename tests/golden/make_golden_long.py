"""Golden fixture for a LONG sequence, generated from the reference itself (build container only, where the
reference checkout is mounted at /root/reference):

    python tests/golden/make_golden_long.py

The reference's `src/lmu.py::LMUFFT` at seq_len 2048, memory 192, theta 1000 (small input / hidden sizes so that
the file stays small): inputs, parameters, H, outputs and autograd gradients, fp32.  It pins the routes the layer
takes for long sequences (fp32: partitioned FFT; bf16: state-space chunks) to the reference's own numbers.  Only
numbers are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
from lmu import LMUFFT  # noqa: E402  (the reference implementation)


def main():
    D, Hd, M, T, theta, B, seed = 16, 32, 192, 2048, 1000, 2, 31
    torch.manual_seed(seed)
    layer = LMUFFT(D, Hd, M, T, theta)
    for lin in (layer.W_u, layer.W_h):
        torch.nn.init.xavier_uniform_(lin.weight)
        torch.nn.init.uniform_(lin.bias, -0.1, 0.1)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, D, generator=g, requires_grad=True)
    gh = torch.randn(B, T, Hd, generator=g)
    h, h_n = layer(x)
    (h * gh).sum().backward()
    out = {
        "x": x.detach().numpy(), "gh": gh.numpy(),
        "w_u": layer.W_u.weight.detach().numpy(), "b_u": layer.W_u.bias.detach().numpy(),
        "W_h": layer.W_h.weight.detach().numpy(), "b_h": layer.W_h.bias.detach().numpy(),
        "H": layer.H.numpy(), "h": h.detach().numpy(), "h_n": h_n.detach().numpy(),
        "gx": x.grad.numpy(), "gw_u": layer.W_u.weight.grad.numpy(), "gb_u": layer.W_u.bias.grad.numpy(),
        "gW_h": layer.W_h.weight.grad.numpy(), "gb_h": layer.W_h.bias.grad.numpy(),
        "meta": np.array([D, Hd, M, T, theta, B], dtype=np.int64),
    }
    np.savez_compressed(os.path.join(HERE, "long2048_f32.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
