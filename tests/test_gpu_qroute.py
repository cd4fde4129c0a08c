"""The bf16 backward's two ways to the encoder gradient: through the memory gradient gm = gz W_h[:, :M] and the
adjoint convolution (autograd's order for src/lmu.py:303-316), and straight from gz against Q^T = H^T W_h[:, :M]^T
(the two linear steps composed; gm is never formed).  Both are held to the oracle elsewhere (tests/test_gpu_bf16.py
runs whichever the library picks); here they are compared with each other on the same inputs, the library's choice
is checked, and a NULL gm on a shape the composed route does not take is refused."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,D,M,Hd", [(64, 400, 384, 192, 384), (3, 129, 80, 64, 128), (2, 512, 96, 40, 256),
                                        (5, 31, 64, 8, 32), (96, 400, 384, 192, 384)])
def test_composed_route_equals_memory_gradient_route(B, T, D, M, Hd):
    from asr_amd import LMUFFT, _lib, functional

    dev = torch.device("cuda:0")
    torch.manual_seed(B + T)
    layer = LMUFFT(D, Hd, M, T, 200.0)
    torch.nn.init.xavier_uniform_(layer.W_u.weight)
    torch.nn.init.xavier_uniform_(layer.W_h.weight)
    layer = layer.to(dev)
    shp = _lib.Shape(B, T, D, M, Hd, T)
    assert _lib.lib.lmufft_backward_needs_gm_bf16(ctypes.byref(shp), _lib.CONV_AUTO, 0) == 0
    assert _lib.lib.lmufft_backward_needs_gm_bf16(ctypes.byref(shp), _lib.CONV_AUTO, 1) == 1  # state-space route
    g = torch.Generator(device=dev).manual_seed(5)
    x0 = torch.randn(B, T, D, device=dev, generator=g).bfloat16()
    gh = torch.randn(B, T, Hd, device=dev, generator=g).bfloat16()
    gh_n = torch.randn(B, Hd, device=dev, generator=g).bfloat16()
    out = {}
    for force in (True, False):
        functional.FORCE_MEMORY_GRADIENT = force
        try:
            layer.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            h, h_n = layer(x)
            torch.autograd.backward([h, h_n], [gh, gh_n])
            out[force] = [x.grad.float()] + [p.grad.float().clone() for p in layer.parameters()]
        finally:
            functional.FORCE_MEMORY_GRADIENT = False
    # gW_h and gb_h do not depend on the route at all
    assert torch.equal(out[True][3], out[False][3]) and torch.equal(out[True][4], out[False][4])
    # gx, gw_u, gb_u see the encoder gradient: the routes round different intermediates to bf16 (gm and H against
    # Q), so they agree to bf16 noise -- 1e-2 of the rms -- not bit for bit
    for a, b, name in zip(out[True][:3], out[False][:3], ("gx", "gw_u", "gb_u")):
        scale = float(a.pow(2).mean().sqrt())
        if name == "gb_u":  # one cancelling sum: scale by the terms' size (the encoder gradient's rms over frames)
            scale = float(out[True][1].abs().max()) + abs(float(a))
        assert float((a - b).pow(2).mean().sqrt()) <= 1e-2 * scale + 1e-7, (name, float((a - b).abs().max()), scale)
    # and the composed route is deterministic
    functional.FORCE_MEMORY_GRADIENT = False
    layer.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_(True)
    h, h_n = layer(x)
    torch.autograd.backward([h, h_n], [gh, gh_n])
    assert torch.equal(x.grad.float(), out[False][0]) and torch.equal(layer.W_u.weight.grad.float(), out[False][1])


def test_null_memory_gradient_is_refused_where_the_composed_route_does_not_apply():
    from asr_amd import _lib

    for shape in ((2, 600, 64, 64, 128), (2, 100, 64, 64, 40), (2, 100, 64, 12, 128)):  # long, Hd % 32, M % 8
        shp = _lib.Shape(*shape, shape[1])
        assert _lib.lib.lmufft_backward_needs_gm_bf16(ctypes.byref(shp), _lib.CONV_AUTO, 0) == 1
    dev = torch.device("cuda:0")
    B, T, D, M, Hd = 2, 600, 64, 64, 128
    shp = _lib.Shape(B, T, D, M, Hd, T)
    z = lambda *s, dt=torch.bfloat16: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
    ws = z(_lib.lib.lmufft_backward_workspace_bytes(ctypes.byref(shp), _lib.BF16), dt=torch.uint8)
    wprep = z(_lib.lib.lmufft_wprep_bytes(ctypes.byref(shp)), dt=torch.uint8)
    hprep = z(_lib.lib.lmufft_hprep_bytes(ctypes.byref(shp)), dt=torch.uint8)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    f32 = torch.float32
    rc = _lib.lib.lmufft_backward_bf16(
        ctypes.byref(shp), p(z(B, T, D)), p(z(1, D, dt=f32)), p(wprep), p(hprep), None, p(z(B, T, dt=f32)), p(z(B, T, M)),
        p(z(B, T, Hd)), None, p(z(B, T, Hd)), None, p(z(B, T, D)), p(z(1, D, dt=f32)), p(z(1, dt=f32)), p(z(Hd, M + D, dt=f32)),
        p(z(Hd, dt=f32)), None, _lib.CONV_AUTO, p(ws), ws.numel(), None)
    assert rc != 0 and b"gm" in _lib.lib.lmufft_last_error()


@pytest.mark.parametrize("B,T,D,M,Hd", [(64, 400, 384, 192, 384), (3, 129, 80, 64, 128), (2, 77, 64, 12, 136)])
def test_relu_bit_mask_is_the_pattern_of_h_and_changes_nothing(B, T, D, M, Hd):
    """The forward's bit mask (bit j of byte (row, c) = h[row][8 c + j] > 0) against h itself -- written by the
    projection's epilogue on the first two shapes and by the separate pass on the third (M % 8 != 0) -- and a
    backward that masks with it against one that reads h: bit-equal gradients."""
    from asr_amd import LMUFFT, _lib, functional

    dev = torch.device("cuda:0")
    torch.manual_seed(T)
    layer = LMUFFT(D, Hd, M, T, 100.0)
    torch.nn.init.xavier_uniform_(layer.W_u.weight)
    torch.nn.init.xavier_uniform_(layer.W_h.weight)
    layer = layer.to(dev)
    g = torch.Generator(device=dev).manual_seed(9)
    x0 = torch.randn(B, T, D, device=dev, generator=g).bfloat16()
    gh = torch.randn(B, T, Hd, device=dev, generator=g).bfloat16()
    gh_n = torch.randn(B, Hd, device=dev, generator=g).bfloat16()
    out = {}
    for use in (True, False):
        functional.USE_RELU_MASK = use
        try:
            layer.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            h, h_n = layer(x)
            if use:
                mask = h.grad_fn.saved_tensors[-1] if hasattr(h.grad_fn, "saved_tensors") else None
                assert mask is not None and mask.dtype == torch.uint8 and mask.shape == (B * T, Hd // 8)
                want = np.packbits((h.detach().float() > 0).cpu().numpy().reshape(B * T, Hd), axis=1, bitorder="little")
                assert np.array_equal(mask.cpu().numpy(), want)
            torch.autograd.backward([h, h_n], [gh, gh_n])
            out[use] = [x.grad] + [p.grad.clone() for p in layer.parameters()]
        finally:
            functional.USE_RELU_MASK = True
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)
