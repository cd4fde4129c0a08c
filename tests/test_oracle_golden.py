"""Pins the CPU oracle against vectors produced by the reference itself.

The .npz files under tests/golden/ were written by tests/golden/make_golden.py,
which imports the reference's LMUFFT in the build container.  These tests run
without a GPU and without the reference checkout.
"""
import json
import os

import numpy as np
import pytest

from oracle import lmufft_oracle as orc

CASES = ["tiny_f64", "tiny_f32", "smoke100_f32", "cfg1a_f32", "cfg1c_f32"]


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def _tol(name):
    # fp64 case: the oracle (time-domain sum) and the reference (FFT) agree to
    # fp64 round-off.  fp32 cases: the reference ran in fp32, the oracle in
    # fp64, so the gap is the reference's own fp32 round-off.
    return (1e-10, 1e-12) if name.endswith("f64") else (1e-4, 2e-5)


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    rtol, atol = _tol(name)
    out = orc.lmufft_forward(g["x"], g["w_u"], g["b_u"], g["W_h"], g["b_h"], g["H"])
    np.testing.assert_allclose(out["h"], g["h"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(out["h_n"], g["h_n"], rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", CASES)
def test_backward_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    rtol, atol = _tol(name)
    gh_n = g["gh_n"] if g["gh_n"].size else None
    out = orc.lmufft_backward(g["x"], g["w_u"], g["b_u"], g["W_h"], g["b_h"], g["H"], g["gh"], gh_n)
    for k in ("gx", "gw_u", "gb_u", "gW_h", "gb_h"):
        ref = g[k]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(out[k], ref, rtol=rtol, atol=atol * scale, err_msg=k)


def test_fft_route_equals_direct_route(golden_dir):
    g = _load(golden_dir, "smoke100_f32")
    rng = np.random.default_rng(0)
    u = np.maximum(rng.standard_normal((3, 100)), 0.0)
    a = orc.causal_conv_direct(u, g["H"])
    b = orc.causal_conv_fft(u.astype(np.float64), g["H"].astype(np.float64))
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)


def test_conv_known_answers(golden_dir):
    H = _load(golden_dir, "buffers_theta1000_M192")["H400"]
    T = 64
    # impulse in -> impulse response out
    u = np.zeros((1, T))
    u[0, 0] = 1.0
    m = orc.causal_conv_direct(u, H)
    np.testing.assert_allclose(m[0], H[:, :T].T.astype(np.float64), rtol=0, atol=0)
    # constant in -> running sum of the impulse response
    m1 = orc.causal_conv_direct(np.ones((1, T)), H)
    np.testing.assert_allclose(m1[0], np.cumsum(H[:, :T].astype(np.float64), axis=1).T, rtol=1e-12)
    # adjoint identity <conv(u), g> == <u, corr(g)>
    rng = np.random.default_rng(1)
    u = rng.standard_normal((2, T))
    gm = rng.standard_normal((2, T, H.shape[0]))
    lhs = (orc.causal_conv_direct(u, H) * gm).sum()
    rhs = (u * orc.anticausal_corr_direct(gm, H)).sum()
    assert abs(lhs - rhs) <= 1e-10 * abs(lhs)


def test_recurrence_equals_conv(golden_dir):
    """m_t = A m_{t-1} + B u_t (src/lmu.py:130, sequential LMU) gives the conv."""
    b = _load(golden_dir, "buffers_theta8_M6")
    A = b["A"].astype(np.float64)
    Bv = b["B"].astype(np.float64)[:, 0]
    T = 16
    H = np.stack([np.linalg.matrix_power(A, t) @ Bv for t in range(T)], axis=1)
    rng = np.random.default_rng(2)
    u = rng.standard_normal((1, T))
    m = np.zeros(6)
    rec = []
    for t in range(T):
        m = A @ m + Bv * u[0, t]
        rec.append(m.copy())
    np.testing.assert_allclose(orc.causal_conv_direct(u, H)[0], np.stack(rec), rtol=1e-10, atol=1e-13)


def test_state_space_and_impulse_match_reference(golden_dir):
    for fname, M, theta, key, T in (
        ("buffers_theta8_M6", 6, 8, "H16", 16),
        ("buffers_theta1000_M192", 192, 1000, "H400", 400),
    ):
        b = _load(golden_dir, fname)
        A, Bm = orc.state_space_matrices(M, theta)
        np.testing.assert_allclose(A, b["A"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(Bm, b["B"], rtol=1e-5, atol=1e-8)
        # from the reference's own A, B the fp32 iteration reproduces H
        H = orc.impulse_response(b["A"], b["B"], T)
        np.testing.assert_allclose(H, b[key], rtol=2e-4, atol=2e-6)


def test_state_dict_manifest(golden_dir):
    with open(os.path.join(golden_dir, "state_dict_manifest_cfg1a.json")) as f:
        man = json.load(f)
    assert man["fft_H"] == {"shape": [192, 401], "dtype": "torch.complex64"}
    assert man["W_h.weight"]["shape"] == [384, 192 + 80]


def test_round_bf16():
    a = np.array([1.0, 1.00390625, 1.005859375, -3.1415926], dtype=np.float32)
    r = orc.round_bf16(a)
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == np.float32(1.0078125)
    assert abs(r[3] + 3.140625) < 1e-6


@pytest.mark.parametrize("name", ["tiny_f32", "smoke100_f32", "cfg1a_f32", "cfg1c_f32"])
def test_torch_port_matches_reference(golden_dir, name):
    """oracle/lmufft_torch_port.py -- what bench.py times as the CPU baseline -- reproduces the
    reference's own outputs and gradients (it follows the same torch.fft route in fp32, so the gap is
    fp32 round-off of two runs of the same arithmetic)."""
    import torch

    from oracle.lmufft_torch_port import PortLMUFFT

    g = _load(golden_dir, name)
    D, Hd, M = g["x"].shape[2], g["W_h"].shape[0], g["H"].shape[0]
    layer = PortLMUFFT(D, Hd, M, torch.from_numpy(g["H"]))
    with torch.no_grad():
        layer.W_u.weight.copy_(torch.from_numpy(g["w_u"]))
        layer.W_u.bias.copy_(torch.from_numpy(g["b_u"]))
        layer.W_h.weight.copy_(torch.from_numpy(g["W_h"]))
        layer.W_h.bias.copy_(torch.from_numpy(g["b_h"]))
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    h, h_n = layer(x)
    np.testing.assert_allclose(h.detach().numpy(), g["h"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(h_n.detach().numpy(), g["h_n"], rtol=1e-4, atol=2e-5)
    loss = (h * torch.from_numpy(g["gh"])).sum()
    if g["gh_n"].size:
        loss = loss + (h_n * torch.from_numpy(g["gh_n"])).sum()
    loss.backward()
    got = {"gx": x.grad, "gw_u": layer.W_u.weight.grad, "gb_u": layer.W_u.bias.grad,
           "gW_h": layer.W_h.weight.grad, "gb_h": layer.W_h.bias.grad}
    for k, v in got.items():
        ref = g[k]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(v.numpy(), ref, rtol=1e-4, atol=2e-5 * scale, err_msg=k)


def test_fft_adjoint_equals_direct_adjoint():
    rng = np.random.default_rng(3)
    H = rng.standard_normal((7, 90)) * np.exp(-np.arange(90) / 25.0)[None, :]
    gm = rng.standard_normal((3, 90, 7))
    a = orc.anticausal_corr_direct(gm, H)
    b = orc.anticausal_corr_fft(gm, H)
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11)
    x = rng.standard_normal((2, 90, 5))
    args = (x, rng.standard_normal((1, 5)), np.array([0.1]), rng.standard_normal((4, 12)), rng.standard_normal(4), H,
            rng.standard_normal((2, 90, 4)))
    d, f = orc.lmufft_backward(*args), orc.lmufft_backward(*args, conv="fft")
    for k in ("gx", "gw_u", "gb_u", "gW_h", "gb_h", "h"):
        np.testing.assert_allclose(d[k], f[k], rtol=1e-8, atol=1e-10, err_msg=k)


def test_long_sequence_matches_reference(golden_dir):
    """The reference's LMUFFT at 2048 frames (tests/golden/make_golden_long.py): the oracle's FFT form -- the one
    the long-sequence GPU tests use -- against the reference's own outputs and autograd gradients."""
    g = _load(golden_dir, "long2048_f32")
    out = orc.lmufft_backward(g["x"], g["w_u"], g["b_u"], g["W_h"], g["b_h"], g["H"], g["gh"], None, conv="fft")
    np.testing.assert_allclose(out["h"], g["h"], rtol=1e-4, atol=2e-5 * float(np.abs(g["h"]).max()))
    np.testing.assert_allclose(out["h"][:, -1, :], g["h_n"], rtol=1e-4, atol=2e-5 * float(np.abs(g["h"]).max()))
    for k in ("gx", "gw_u", "gb_u", "gW_h", "gb_h"):
        ref = g[k]
        # the reference ran in fp32: its own round-off on sums of 4096 terms is what separates the two
        np.testing.assert_allclose(out[k], ref, rtol=2e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())), err_msg=k)
