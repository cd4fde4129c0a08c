"""GPU parity tests of the LDS-resident FFT route of the causal convolution
(fp32): same bar as the Toeplitz route, rtol 1e-4 with an absolute floor of
1e-5 x the tensor's largest magnitude (an fp32 FFT's rounding error is relative
to the signal's norm, not to each element; the reference's own torch.fft route,
src/lmu.py:308-313, has the same property)."""
import os

import numpy as np
import pytest
import torch

from oracle import lmufft_oracle as orc
from test_gpu_parity import _close, _dev, _load, _rand_H, _t

pytestmark = pytest.mark.gpu

# (B, T, M, ldH): transform lengths 256 / 512 / 1024, odd channel counts (the
# kernels pack two channels per complex transform), batch sizes that leave
# waves idle, T < ldH (the filter is truncated to the call's T)
FFT_SHAPES = [
    (2, 16, 6, 16),
    (3, 37, 5, 41),
    (1, 128, 1, 128),
    (2, 129, 7, 129),
    (2, 100, 128, 100),
    (5, 200, 192, 200),
    (2, 256, 33, 300),
    (3, 257, 66, 257),
    (2, 400, 192, 400),
    (1, 512, 65, 512),
    # longer than one transform: partitioned convolution in blocks of 256 samples
    (2, 513, 6, 513),     # three blocks, the last one holding a single frame
    (1, 600, 33, 640),    # odd channel count, T < seq_len
    (2, 1536, 64, 1536),  # exactly six blocks = one adjoint block group
    (1, 1793, 8, 1793),   # eight blocks: two adjoint block groups, ragged end
    (1, 3000, 64, 3000),  # the long-utterance length (12 blocks)
]


@pytest.mark.parametrize("B,T,M,ldH", FFT_SHAPES)
def test_fft_conv_forward_matches_oracle(B, T, M, ldH):
    from asr_amd import causal_conv

    dev = _dev()
    rng = np.random.default_rng(B * 1000 + T)
    u = np.maximum(rng.standard_normal((B, T)), 0).astype(np.float32)
    H = _rand_H(rng, M, ldH)
    got = causal_conv(_t(u, dev), _t(H, dev), "fft").cpu().numpy()
    _close(got, orc.causal_conv_direct(u, H[:, :T]), "m")


@pytest.mark.parametrize("B,T,M,ldH", FFT_SHAPES)
def test_fft_conv_backward_matches_oracle(B, T, M, ldH):
    from asr_amd import causal_conv_adjoint

    dev = _dev()
    rng = np.random.default_rng(B * 2000 + T)
    gm = rng.standard_normal((B, T, M)).astype(np.float32)
    H = _rand_H(rng, M, ldH)
    got = causal_conv_adjoint(_t(gm, dev), _t(H, dev), "fft").cpu().numpy()
    _close(got, orc.anticausal_corr_direct(gm, H[:, :T]), "gu")


def test_fft_and_toeplitz_agree_at_full_size(golden_dir):
    """B=64, T=400, M=192 with the Legendre impulse response: the two routes
    agree to fp32 rounding, and the FFT route is causal to rounding."""
    from asr_amd import causal_conv, causal_conv_adjoint

    dev = _dev()
    H = _t(np.load(os.path.join(golden_dir, "buffers_theta1000_M192.npz"))["H400"], dev)
    B, T, M = 64, 400, 192
    g = torch.Generator(device="cpu").manual_seed(11)
    u = torch.relu(torch.randn(B, T, generator=g)).to(dev)
    gm = torch.randn(B, T, M, generator=g).to(dev)
    a = causal_conv(u, H, "toeplitz")
    b = causal_conv(u, H, "fft")
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    ga = causal_conv_adjoint(gm, H, "toeplitz")
    gb = causal_conv_adjoint(gm, H, "fft")
    assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max())
    u2 = u.clone()
    u2[:, 200:] = 0
    b2 = causal_conv(u2, H, "fft")
    assert float((b2[:, :200] - b[:, :200]).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("name", ["tiny_f32", "smoke100_f32", "cfg1a_f32", "cfg1c_f32"])
def test_layer_fft_route_matches_reference_golden(golden_dir, name):
    """Full layer forward + backward with conv_algo="fft" against the reference's outputs."""
    from asr_amd import lmufft_layer

    dev = _dev()
    g = _load(golden_dir, name)
    x = _t(g["x"], dev).requires_grad_(True)
    w_u = _t(g["w_u"], dev).requires_grad_(True)
    b_u = _t(g["b_u"], dev).requires_grad_(True)
    W_h = _t(g["W_h"], dev).requires_grad_(True)
    b_h = _t(g["b_h"], dev).requires_grad_(True)
    H = _t(g["H"], dev)
    h, h_n = lmufft_layer(x, w_u, b_u, W_h, b_h, H, conv_algo="fft")
    _close(h.detach().cpu().numpy(), g["h"], "h")
    _close(h_n.detach().cpu().numpy(), g["h_n"], "h_n")
    loss = (h * _t(g["gh"], dev)).sum()
    if g["gh_n"].size:
        loss = loss + (h_n * _t(g["gh_n"], dev)).sum()
    loss.backward()
    o = orc.lmufft_backward(g["x"], g["w_u"], g["b_u"], g["W_h"], g["b_h"], g["H"], g["gh"],
                            g["gh_n"] if g["gh_n"].size else None)
    ga_abs = float(np.abs(o["gu"] * (o["u"] > 0)).sum())
    _close(x.grad.cpu().numpy(), g["gx"], "gx")
    _close(w_u.grad.cpu().numpy(), g["gw_u"], "gw_u", cond=ga_abs * float(np.abs(g["x"]).max()))
    _close(b_u.grad.cpu().numpy(), g["gb_u"], "gb_u", cond=ga_abs)
    _close(W_h.grad.cpu().numpy(), g["gW_h"], "gW_h")
    _close(b_h.grad.cpu().numpy(), g["gb_h"], "gb_h")


def test_fft_route_rejects_sequences_beyond_its_range():
    from asr_amd import causal_conv
    from asr_amd.functional import fft_supported

    dev = _dev()
    assert fft_supported(512) and fft_supported(3000) and fft_supported(16384) and not fft_supported(16385)
    u = torch.zeros(1, 16500, device=dev)
    H = torch.zeros(2, 16500, device=dev)
    with pytest.raises(RuntimeError, match="16500"):
        causal_conv(u, H, "fft")


def test_long_sequence_layer_routes_agree():
    """T = 1200 through the whole fp32 layer, forward and backward: the partitioned FFT route
    (the module's choice for long sequences) against the Toeplitz route."""
    from asr_amd import LMUFFT

    dev = _dev()
    torch.manual_seed(21)
    B, T, D, M, Hd = 3, 1200, 48, 64, 96
    layer = LMUFFT(D, Hd, M, T, 300.0).to(dev)
    x = torch.randn(B, T, D, device=dev)
    gy = torch.randn(B, T, Hd, device=dev)
    out = {}
    for algo in ("fft", "toeplitz", "auto"):
        layer.conv_algo = algo
        xi = x.clone().requires_grad_(True)
        layer.zero_grad(set_to_none=True)
        y, h_n = layer(xi)
        (y * gy).sum().backward()
        out[algo] = [y.detach(), h_n.detach(), xi.grad] + [p.grad.clone() for p in layer.parameters()]
    for a, b2 in zip(out["fft"], out["toeplitz"]):
        scale = float(b2.abs().max()) + 1e-30
        assert float((a - b2).abs().max()) <= 2e-4 * scale, float((a - b2).abs().max()) / scale
    for a, b2 in zip(out["auto"], out["fft"]):
        assert torch.equal(a, b2)  # auto takes the FFT route at this length
