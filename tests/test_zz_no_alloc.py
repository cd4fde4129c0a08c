"""Allocation-free adjoint convolution (lmufft_conv_backward_noalloc_f32) against the workspace
form it is a variant of (lmufft_conv_backward_f32), through the C ABI.

Two calls in a row, the second with a smaller input: the built-in scratch array keeps the first
call's partial results, so the second call must not pick any of them up.  Bar: the fp32 bar of
the parity tests (|got - ref| <= 1e-4 |ref| + 1e-5 max|ref|); the FFT route sums in a fixed order,
so there the two entry points must agree bit for bit.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    from asr_amd import _lib
    from asr_amd.lmu import impulse_response, legendre_state_space

    dev = torch.device("cuda:0")
    A, B = legendre_state_space(192, 1000.0)
    H, _ = impulse_response(A, B, 400)
    return _lib, dev, H.to(dev).contiguous()


def _adjoint_pair(_lib, dev, H, Bn, T, algo_name, seed):
    from asr_amd.functional import _ALGOS, _ptr, _shape, _stream, _workspace, prepare_impulse_fft

    M, ldH = H.shape
    g = torch.Generator(device=dev).manual_seed(seed)
    gm = torch.randn(Bn, T, M, device=dev, generator=g)
    algo = _ALGOS[algo_name]
    shp = _shape(Bn, T, 1, M, 1, ldH)
    fftprep = prepare_impulse_fft(H, T) if algo_name == "fft" else None
    need = _lib.lib.lmufft_conv_workspace_bytes(ctypes.byref(shp), _lib.F32, algo)
    assert need <= _lib.NOALLOC_SCRATCH_BYTES
    ws = _workspace(need, dev)
    ref = torch.full((Bn, T), float("nan"), device=dev)
    got = torch.full((Bn, T), float("nan"), device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib.lmufft_conv_backward_f32(ctypes.byref(shp), _ptr(gm), _ptr(H), _ptr(fftprep), _ptr(ref),
                                                    algo, _ptr(ws), ws.numel(), _stream(dev)),
                   "lmufft_conv_backward_f32")
        _lib.check(_lib.lib.lmufft_conv_backward_noalloc_f32(ctypes.byref(shp), _ptr(gm), _ptr(H), _ptr(fftprep),
                                                            _ptr(got), algo, _stream(dev)),
                   "lmufft_conv_backward_noalloc_f32")
    torch.cuda.synchronize(dev)
    return got.cpu().numpy().astype(np.float64), ref.cpu().numpy().astype(np.float64), gm, fftprep


@pytest.mark.parametrize("algo", ["fft", "toeplitz"])
def test_noalloc_adjoint_matches_workspace_form_twice(algo):
    _lib, dev, H = _setup()
    # the first call leaves its partial results in the scratch array; the second is smaller in both
    # batch and length, so its slabs cover only a part of what the first one wrote
    for Bn, T, seed in ((48, 400, 11), (5, 136, 12)):
        got, ref, gm, fftprep = _adjoint_pair(_lib, dev, H, Bn, T, algo, seed)
        assert np.isfinite(got).all() and np.isfinite(ref).all()
        if algo == "fft":
            assert np.array_equal(got, ref), f"FFT route differs at B={Bn}, T={T}"
        tol = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max()
        assert (np.abs(got - ref) <= tol).all(), (algo, Bn, T, np.abs(got - ref).max())
        # and both against the direct sum on a few entries (the oracle's definition of the adjoint)
        Hn = H.cpu().numpy().astype(np.float64)
        gmn = gm.cpu().numpy().astype(np.float64)
        for (b, s) in ((0, 0), (Bn - 1, T - 1), (Bn // 2, T // 3)):
            want = sum(float(gmn[b, t] @ Hn[:, t - s]) for t in range(s, T))
            assert abs(got[b, s] - want) <= 1e-4 * abs(want) + 1e-5 * np.abs(ref).max()


def test_noalloc_rejects_oversized_shape():
    _lib, dev, H = _setup()
    from asr_amd.functional import _ptr, _shape, _stream

    M, ldH = H.shape
    Bn, T = 4096, 400  # 4096 * 7 * 400 * 4 B of slabs: beyond the built-in buffer
    shp = _shape(Bn, T, 1, M, 1, ldH)
    assert _lib.lib.lmufft_conv_workspace_bytes(ctypes.byref(shp), _lib.F32, _lib.CONV_TOEPLITZ) > _lib.NOALLOC_SCRATCH_BYTES
    one = torch.zeros(1, device=dev)  # never dereferenced: the size check comes first
    with torch.cuda.device(dev):
        rc = _lib.lib.lmufft_conv_backward_noalloc_f32(ctypes.byref(shp), _ptr(one), _ptr(H), None, _ptr(one),
                                                      _lib.CONV_TOEPLITZ, _stream(dev))
    assert rc != 0
    assert b"no-alloc" in _lib.lib.lmufft_last_error()
