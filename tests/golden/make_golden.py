"""Generate the golden fixtures in this directory from the reference itself.

Run ONLY in the build container, where the reference checkout is mounted at
/root/reference:

    python tests/golden/make_golden.py

It imports the reference's `src/lmu.py::LMUFFT` (torch + numpy + scipy only),
drives it with seeded inputs and stores inputs, parameters, buffers, outputs
and autograd gradients as .npz files.  Nothing of the reference's source is
stored -- only numbers.  The tests and the GPU box read the .npz files; they
never read /root/reference.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
from lmu import LMUFFT  # noqa: E402  (the reference implementation)


def run_case(name, D, Hd, M, T, theta, B, seed, dtype, bias_scale, use_hn=True):
    torch.manual_seed(seed)
    layer = LMUFFT(D, Hd, M, T, theta)
    # same re-initialisation the reference model applies to every nn.Linear
    # (xavier-uniform weights), plus optional non-zero biases so that the bias
    # paths are exercised
    for lin in (layer.W_u, layer.W_h):
        torch.nn.init.xavier_uniform_(lin.weight)
        if bias_scale:
            torch.nn.init.uniform_(lin.bias, -bias_scale, bias_scale)
        else:
            torch.nn.init.zeros_(lin.bias)
    if dtype == torch.float64:
        layer = layer.double()
        # .double() leaves the complex64 spectrum untouched; rebuild it so the
        # fp64 case really is fp64 end to end
        layer.fft_H = torch.fft.rfft(layer.H, n=2 * T, dim=-1)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, D, generator=g, dtype=dtype, requires_grad=True)
    gh = torch.randn(B, T, Hd, generator=g, dtype=dtype)
    gh_n = torch.randn(B, Hd, generator=g, dtype=dtype)
    h, h_n = layer(x)
    loss = (h * gh).sum()
    if use_hn:
        loss = loss + (h_n * gh_n).sum()
    loss.backward()
    out = {
        "x": x.detach().numpy(),
        "gh": gh.numpy(),
        "gh_n": gh_n.numpy() if use_hn else np.zeros((0,), dtype=gh.numpy().dtype),
        "w_u": layer.W_u.weight.detach().numpy(),
        "b_u": layer.W_u.bias.detach().numpy(),
        "W_h": layer.W_h.weight.detach().numpy(),
        "b_h": layer.W_h.bias.detach().numpy(),
        "H": layer.H.numpy(),
        "h": h.detach().numpy(),
        "h_n": h_n.detach().numpy(),
        "gx": x.grad.numpy(),
        "gw_u": layer.W_u.weight.grad.numpy(),
        "gb_u": layer.W_u.bias.grad.numpy(),
        "gW_h": layer.W_h.weight.grad.numpy(),
        "gb_h": layer.W_h.bias.grad.numpy(),
        "meta": np.array([D, Hd, M, T, theta, B], dtype=np.int64),
    }
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if k in ("x", "h", "gW_h")})
    return layer


def main():
    run_case("tiny_f64", 12, 10, 6, 16, 8, 2, 11, torch.float64, 0.3)
    run_case("tiny_f32", 12, 10, 6, 16, 8, 2, 11, torch.float32, 0.3)
    # the shape the reference's own smoke script drives (pytorch-test-setup.py:55-64)
    run_case("smoke100_f32", 80, 256, 128, 100, 1000, 4, 21, torch.float32, 0.1)
    # CPU-runnable reference case: 80-mel input, 400 frames / in-encoder 384 input, 200 frames
    layer = run_case("cfg1a_f32", 80, 384, 192, 400, 1000, 2, 31, torch.float32, 0.0)
    run_case("cfg1c_f32", 384, 384, 192, 200, 1000, 2, 41, torch.float32, 0.0, use_hn=False)

    # constructor buffers for theta=1000, M=192
    sd = layer.state_dict()
    manifest = {k: {"shape": list(v.shape), "dtype": str(v.dtype)} for k, v in sd.items()}
    with open(os.path.join(HERE, "state_dict_manifest_cfg1a.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    long_layer = LMUFFT(8, 8, 192, 3000, 1000)
    H3k = long_layer.H.numpy()
    np.savez(
        os.path.join(HERE, "buffers_theta1000_M192.npz"),
        A=layer.A.numpy(),
        B=layer.B.numpy(),
        H400=layer.H.numpy(),
        fft_H400=layer.fft_H.numpy(),
        H3000_first64=H3k[:, :64],
        H3000_last64=H3k[:, -64:],
        H3000_col1500=H3k[:, 1500],
    )
    small = LMUFFT(4, 4, 6, 16, 8)
    np.savez(
        os.path.join(HERE, "buffers_theta8_M6.npz"),
        A=small.A.numpy(),
        B=small.B.numpy(),
        H16=small.H.numpy(),
        fft_H16=small.fft_H.numpy(),
    )
    print("manifest", manifest)


if __name__ == "__main__":
    main()
