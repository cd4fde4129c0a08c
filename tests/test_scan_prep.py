"""CPU checks of the state-space ("scan") convolution route's constants: the block that
lmufft_scanprep_build_host writes from the module's buffers A and B reproduces the reference's causal
convolution with H[:, t] = A^t B (src/lmu.py:275-283, 303-312) and its adjoint when the route is
evaluated in numpy from that block -- layout, powers, Toeplitz block and both recurrences."""
import ctypes

import numpy as np
import pytest

from oracle import lmufft_oracle as orc
from tests.scan_common import round_bf16, scan_adjoint_numpy, scan_forward_numpy, scan_layout, scan_sections


def _block(M, theta):
    from asr_amd import _lib
    from asr_amd.lmu import impulse_response, legendre_state_space

    A, B = legendre_state_space(M, theta)
    n = _lib.lib.lmufft_scanprep_bytes(M)
    L = _lib.lib.lmufft_scan_chunk()
    assert n == scan_layout(M, L)["total"]
    a = np.ascontiguousarray(A.numpy(), np.float32)
    b = np.ascontiguousarray(B.numpy().reshape(-1), np.float32)
    host = np.empty(n, np.uint8)
    assert _lib.lib.lmufft_scanprep_build_host(a.ctypes.data, b.ctypes.data, M, host.ctypes.data) == 0
    return host, L, A, B, impulse_response


def test_scanprep_sizes_and_errors():
    from asr_amd import _lib

    assert _lib.lib.lmufft_scanprep_bytes(192) > 0 and _lib.lib.lmufft_scanprep_bytes(256) > 0
    assert _lib.lib.lmufft_scanprep_bytes(100) == 0 and _lib.lib.lmufft_scanprep_bytes(264) == 0
    assert _lib.lib.lmufft_scanprep_build_host(None, None, 192, None) != 0
    assert b"NULL" in _lib.lib.lmufft_last_error()


@pytest.mark.parametrize("M,theta,T", [(16, 40.0, 300), (192, 1000.0, 600), (64, 200.0, 129)])
def test_scanprep_reproduces_the_convolution(M, theta, T):
    host, L, A, B, impulse_response = _block(M, theta)
    sec = scan_sections(host, M, L)
    H, _ = impulse_response(A, B, T)  # the reference's fp32 iteration
    H = H.numpy().astype(np.float64)
    # sections agree with each other and with H
    assert np.array_equal(sec["WcT"], sec["Wc"].T) and np.array_equal(sec["ALT"], sec["AL"].T)
    assert np.array_equal(sec["HsT"], sec["Hs"].T)
    Lh = min(L, T - 1)
    assert np.allclose(sec["Hs"][L - Lh:, :], H[:, Lh:0:-1].T, rtol=1e-4, atol=1e-6)  # Hs[i] = A^(L-i) B = H[:, L-i]
    assert np.allclose(sec["Wc"][M, :M], H[:, 0], rtol=2 ** -8)                       # Toeplitz block, lag 0
    assert np.all(sec["Wc"][M + 1:, :M] == 0)                                          # nothing above the diagonal
    rng = np.random.default_rng(M + T)
    u = round_bf16(np.maximum(rng.standard_normal((2, T)), 0.0))
    gm = round_bf16(rng.standard_normal((2, T, M)))
    ref = orc.causal_conv_direct(u, H)
    term = np.sqrt(orc.causal_conv_direct(u ** 2, H ** 2))
    got = scan_forward_numpy(u, sec, M, L)
    assert np.all(np.abs(got - ref) <= 2e-2 * (np.abs(ref) + 2 * term) + 1e-30)
    assert np.sqrt(((got - ref) ** 2).mean()) <= 5e-3 * np.sqrt((ref ** 2).mean())
    gref = orc.anticausal_corr_direct(gm, H)
    gterm = np.sqrt(orc.anticausal_corr_direct(gm ** 2, H ** 2))
    ggot = scan_adjoint_numpy(gm, sec, M, L)
    assert np.all(np.abs(ggot - gref) <= 2e-2 * (np.abs(gref) + 2 * gterm) + 1e-30)
    assert np.sqrt(((ggot - gref) ** 2).mean()) <= 5e-3 * np.sqrt((gref ** 2).mean())
