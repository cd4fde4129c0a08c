"""The long-utterance configuration at its full shape: 30 s of audio = 3000 frames, batch 32,
input = hidden = 384, memory 192 (the reference's layer, src/lmu.py:297-320, at seq_len 3000).

One batch item of the full-shape run is checked against the fp64 oracle (its FFT form, which the CPU
tests tie to the direct causal sums; the O(T^2 M) direct sum takes minutes at this length), the
parameter gradients -- sums over the whole batch -- on a two-utterance batch of the same length.
fp32 takes `conv_algo="auto"` (the partitioned FFT route at this length), bf16 both of its routes:
the state-space chunks that "auto" selects for long sequences, and the Toeplitz-MFMA kernels."""
import numpy as np
import pytest
import torch

from oracle import lmufft_oracle as orc

pytestmark = pytest.mark.gpu

B, T, D, M, HD, THETA = 32, 3000, 384, 192, 384, 1000.0


def _layer(dev, conv_algo="auto"):
    from asr_amd import LMUFFT

    torch.manual_seed(4)
    layer = LMUFFT(D, HD, M, T, THETA, conv_algo=conv_algo)
    torch.nn.init.xavier_uniform_(layer.W_u.weight)
    torch.nn.init.xavier_uniform_(layer.W_h.weight)
    torch.nn.init.zeros_(layer.W_u.bias)
    torch.nn.init.normal_(layer.W_h.bias, std=0.05)
    return layer.to(dev)


def _np(t):
    return t.detach().float().cpu().numpy()


def _params(layer, rounded):
    sd = {k: _np(v) for k, v in layer.state_dict().items() if k != "fft_H"}
    # what the bf16 kernels round: W_h and H (their MFMA operands).  w_u, the biases and u stay fp32.
    r = orc.round_bf16 if rounded else (lambda a: a)
    return sd["W_u.weight"], sd["W_u.bias"], r(sd["W_h.weight"]), sd["W_h.bias"], r(sd["H"])


def _close32(got, ref, name, cancel=None):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    tol = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max() + (0.0 if cancel is None else 1e-6 * cancel)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), (name, int(bad.sum()), float(np.abs(got - ref).max()), float(np.abs(ref).max()))


def test_c4_fp32_full_shape():
    dev = torch.device("cuda:0")
    layer = _layer(dev)
    g = torch.Generator(device=dev).manual_seed(1234 + 4)
    x = torch.randn(B, T, D, device=dev, generator=g).requires_grad_(True)
    gh = torch.randn(B, T, HD, device=dev, generator=g)
    h, h_n = layer(x)
    h.backward(gh)
    assert torch.equal(h_n, h[:, -1, :])
    b = 17
    w_u, b_u, W_h, b_h, H = _params(layer, rounded=False)
    # ReLU pattern: at 1.15 M activations per utterance a handful of pre-activations lie within fp32
    # round-off of zero (|z| < 1e-5) and land on the other branch than in fp64; one such flip moves gm
    # at its frame and, through the adjoint convolution, ga at the ~1000 frames before it by ~1e-2.
    # The backward is therefore checked on the kernel's own pattern, and the pattern itself is
    # checked: it may differ from the fp64 one only where |z| is round-off.
    fwd = orc.lmufft_forward(_np(x[b:b + 1]), w_u, b_u, W_h, b_h, H, conv="fft")
    z = fwd["m"] @ W_h[:, :M].T.astype(np.float64) + _np(x[b:b + 1]).astype(np.float64) @ W_h[:, M:].T + b_h
    h_mask = _np(h[b:b + 1]) > 0
    flips = h_mask != (z > 0)
    assert flips.sum() <= 1e-5 * flips.size and (np.abs(z[flips]) < 1e-5).all(), (int(flips.sum()), np.abs(z[flips]).max())
    ref = orc.lmufft_backward(_np(x[b:b + 1]), w_u, b_u, W_h, b_h, H, _np(gh[b:b + 1]), conv="fft", h_mask=h_mask)
    _close32(_np(h[b:b + 1]), ref["h"], "h")
    _close32(_np(x.grad[b:b + 1]), ref["gx"], "gx")
    # parameter gradients on a two-utterance batch of the same length
    layer.zero_grad(set_to_none=True)
    x2 = x.detach()[:2].clone().requires_grad_(True)
    h2, _ = layer(x2)
    h2.backward(gh[:2])
    ref2 = orc.lmufft_backward(_np(x2), w_u, b_u, W_h, b_h, H, _np(gh[:2]), conv="fft", h_mask=_np(h2) > 0)
    absx = np.abs(_np(x2))
    _close32(_np(layer.W_h.weight.grad), ref2["gW_h"], "gW_h")
    _close32(_np(layer.W_h.bias.grad), ref2["gb_h"], "gb_h")
    # gw_u / gb_u are sums of ~6000 signed terms far larger than the result: allow the fp32 rounding of the terms
    _close32(_np(layer.W_u.weight.grad), ref2["gw_u"], "gw_u", cancel=np.einsum("bt,btd->d", np.abs(ref2["ga"]), absx)[None, :])
    _close32(_np(layer.W_u.bias.grad), ref2["gb_u"], "gb_u", cancel=np.abs(ref2["ga"]).sum())


@pytest.mark.parametrize("conv_algo", ["auto", "toeplitz"])
def test_c4_bf16_full_shape(conv_algo):
    """bf16 at T = 3000, on the route conv_algo="auto" takes at this length (the state-space chunks) and
    on the Toeplitz-MFMA kernels.  Against the oracle on the bf16-rounded operands and the kernel's own ReLU
    pattern every output is inside the bf16 bar.  (A comparison with the oracle on the UN-rounded
    operands reads very differently -- max error 0.20 of max|gx| and 0.07 of max|gW_h| at this shape:
    rounding x to bf16 moves the encoder's pre-activation x.w_u, a cancelling sum of 384 terms, by
    ~4 % of its spread, so a few per cent of the frames land on the other side of u = relu(.) and
    their whole rank-one term ga w_u appears in or vanishes from gx.  That is a property of the
    rounding of the INPUT, the same for any bf16 implementation, not of the accumulation: on the
    frames whose ReLU pattern agrees the error is bf16 rounding noise, which this test checks.)"""
    from tests.test_gpu_bf16 import _close, _termnorms

    dev = torch.device("cuda:0")
    layer = _layer(dev, conv_algo)
    g = torch.Generator(device=dev).manual_seed(1234 + 4)
    x = torch.randn(B, T, D, device=dev, generator=g).bfloat16().requires_grad_(True)
    gh = torch.randn(B, T, HD, device=dev, generator=g).bfloat16()
    h, h_n = layer(x)
    h.backward(gh)
    assert h.dtype == torch.bfloat16 and torch.equal(h_n, h[:, -1, :])
    b = 9
    w_u, b_u, W_h, b_h, H = _params(layer, rounded=True)
    xb, ghb = _np(x[b:b + 1]), _np(gh[b:b + 1])
    ref = orc.lmufft_backward(xb, w_u, b_u, W_h, b_h, H, ghb, h_mask=_np(h[b:b + 1]) > 0, conv="fft")
    tn = _termnorms(xb, w_u, W_h, H, ref, conv="fft")
    _close(_np(h[b:b + 1]), ref["h"], "h C4", tn["h"])
    _close(_np(x.grad[b:b + 1]), ref["gx"], "gx C4", tn["gx"])
    # the un-rounded comparison, restricted to frames whose encoder ReLU pattern is the same
    w0, _, W0, _, H0 = _params(layer, rounded=False)
    fwd0 = orc.lmufft_forward(xb, w0, b_u, W0, b_h, H0, conv="fft")
    same = (fwd0["u"] > 0) == (ref["u"] > 0)
    flipped = 1.0 - float(same.mean())
    assert flipped < 0.05, flipped
    ref0 = orc.lmufft_backward(xb, w0, b_u, W0, b_h, H0, ghb, h_mask=_np(h[b:b + 1]) > 0, u_mask=ref["u"] > 0, conv="fft")
    err = np.abs(_np(x.grad[b:b + 1]) - ref0["gx"])
    assert err.max() <= 4e-2 * (np.abs(ref0["gx"]) + 2 * tn["gx"]).max(), err.max()
    # parameter gradients on a two-utterance batch
    layer.zero_grad(set_to_none=True)
    x2 = x.detach()[:2].clone().requires_grad_(True)
    h2, _ = layer(x2)
    h2.backward(gh[:2])
    ref2 = orc.lmufft_backward(_np(x2), w_u, b_u, W_h, b_h, H, _np(gh[:2]), h_mask=_np(h2) > 0, conv="fft")
    tn2 = _termnorms(_np(x2), w_u, W_h, H, ref2, conv="fft")
    for p, k in ((layer.W_h.weight, "gW_h"), (layer.W_h.bias, "gb_h"), (layer.W_u.weight, "gw_u"), (layer.W_u.bias, "gb_u")):
        _close(_np(p.grad), ref2[k], k + " C4", tn2[k])
