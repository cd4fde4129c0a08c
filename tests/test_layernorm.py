"""Residual add + LayerNorm: the oracle against torch's layer_norm on the CPU, and the HIP kernel
(through the C ABI) against the oracle."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import layernorm_oracle as lo  # noqa: E402


def _case(rows, C, seed, with_res=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, C, generator=g) * 3.0 + 0.5
    res = torch.randn(rows, C, generator=g) if with_res else None
    w = torch.randn(C, generator=g)
    b = torch.randn(C, generator=g)
    return x, res, w, b


@pytest.mark.parametrize("rows,C,with_res", [(7, 384, True), (1, 4, False), (33, 1024, True), (5, 100, False)])
def test_oracle_matches_torch_layer_norm(rows, C, with_res):
    x, res, w, b = _case(rows, C, rows * 31 + C, with_res)
    v = x if res is None else x + res
    want = torch.nn.functional.layer_norm(v.double(), (C,), w.double(), b.double(), 1e-5).float().numpy()
    got = lo.add_layer_norm(x.numpy(), None if res is None else res.numpy(), w.numpy(), b.numpy(), 1e-5)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    # and the fp32 op the reference actually runs, at fp32 accuracy
    want32 = torch.nn.functional.layer_norm(v, (C,), w, b, 1e-5).numpy()
    np.testing.assert_allclose(got, want32, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("res_dtype", [None, torch.float32, torch.bfloat16])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(1, 4), (67, 384), (9, 100), (130, 1024), (1000, 512)])
def test_kernel_matches_oracle(rows, C, x_dtype, res_dtype, out_dtype):
    from asr_amd.layernorm import add_layer_norm

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    x, res, w, b = _case(rows, C, rows + C, res_dtype is not None)
    x = x.to(x_dtype)
    res = None if res is None else res.to(res_dtype)
    got = add_layer_norm(x.to(dev), None if res is None else res.to(dev), w.to(dev), b.to(dev), 1e-5,
                         out_dtype=out_dtype).float().cpu().numpy()
    want = lo.add_layer_norm(x.float().numpy(), None if res is None else res.float().numpy(), w.numpy(), b.numpy())
    if out_dtype == torch.float32:
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5)  # fp32 bar
    else:
        np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-2)  # one bf16 rounding of the result
        exact = torch.from_numpy(want).to(torch.bfloat16).float().numpy()
        assert np.mean(got == exact) > 0.99  # the rounding itself is round-to-nearest-even of the fp32 value


@pytest.mark.gpu
def test_kernel_in_place_and_errors():
    """out may alias x (direct C-ABI call); bad widths and CPU tensors are refused."""
    from asr_amd import _lib
    from asr_amd.layernorm import add_layer_norm

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    x, res, w, b = _case(257, 384, 11)
    want = lo.add_layer_norm(x.numpy(), res.numpy(), w.numpy(), b.numpy())
    xd, rd, wd, bd = x.to(dev), res.to(dev), w.to(dev), b.to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    rc = _lib.lib.lmufft_add_layernorm(p(xd), _lib.F32, p(rd), _lib.F32, p(wd), p(bd), 1e-5, 257, 384, p(xd), _lib.F32,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "lmufft_add_layernorm")
    np.testing.assert_allclose(xd.cpu().numpy(), want, rtol=1e-4, atol=2e-5)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        add_layer_norm(torch.zeros(3, 6, device=dev), None, torch.ones(6, device=dev), torch.zeros(6, device=dev))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        add_layer_norm(torch.zeros(3, 8), None, torch.ones(8), torch.zeros(8))
    assert add_layer_norm(torch.zeros(0, 8, device=dev), None, torch.ones(8, device=dev),
                          torch.zeros(8, device=dev)).shape == (0, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,D,M,Hd", [(3, 50, 80, 64, 128), (2, 400, 384, 192, 384), (5, 37, 24, 8, 40)])
def test_layer_norm_in_projection_epilogue(B, T, D, M, Hd, out_dtype):
    """LMUFFT layer + LayerNorm in one pass (lmufft_forward_ln_bf16) against the layer followed by the
    oracle's layer norm of its bf16 output; h_n identical to the plain forward's."""
    from asr_amd import LMUFFT

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 100 + Hd)
    layer = LMUFFT(D, Hd, M, T, 100.0).to(dev)
    ln = torch.nn.LayerNorm(Hd).to(dev)
    torch.nn.init.normal_(ln.weight, 1.0, 0.3)
    torch.nn.init.normal_(ln.bias, 0.0, 0.3)
    x = torch.randn(B, T, D, device=dev).bfloat16()
    with torch.no_grad():
        assert layer.supports_fused_layer_norm(x)
        h, h_n = layer(x)
        y, h_n2 = layer.forward_layer_norm(x, ln, out_dtype)
    assert y.dtype == out_dtype and torch.equal(h_n, h_n2)
    want = lo.add_layer_norm(h.float().cpu().numpy(), None, ln.weight.detach().cpu().numpy(),
                             ln.bias.detach().cpu().numpy(), ln.eps)
    got = y.float().cpu().numpy()
    if out_dtype == torch.float32:
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5)
    else:
        np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-2)
    # shapes the epilogue cannot take are refused by the C ABI, and the module says so beforehand
    wide = LMUFFT(D, 512, M, T, 100.0).to(dev)
    with torch.no_grad():
        assert not wide.supports_fused_layer_norm(x)
        with pytest.raises(RuntimeError, match="lmufft_add_layernorm"):
            wide.forward_layer_norm(x, torch.nn.LayerNorm(512).to(dev))
