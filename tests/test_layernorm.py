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


def test_oracle_backward_matches_torch_autograd():
    """The backward restatement against torch autograd of layer_norm(x + residual) on the CPU (fp64)."""
    rng = np.random.default_rng(9)
    for with_res in (False, True):
        x = rng.standard_normal((13, 24)).astype(np.float32)
        r = rng.standard_normal((13, 24)).astype(np.float32) if with_res else None
        w, dy = rng.standard_normal(24), rng.standard_normal((13, 24))
        ds, dg, db = lo.add_layer_norm_backward(x, r, w, dy, 1e-5)
        tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        tr = torch.tensor(r, dtype=torch.float64, requires_grad=True) if with_res else None
        tw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        tb = torch.zeros(24, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.layer_norm(tx if tr is None else tx + tr, (24,), tw, tb, 1e-5).backward(torch.tensor(dy))
        # the oracle forms x + residual in fp32, as the reference does; torch here adds in fp64
        tol = dict(rtol=1e-5, atol=1e-6) if with_res else dict(rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ds, tx.grad.numpy(), **tol)
        if with_res:
            np.testing.assert_allclose(ds, tr.grad.numpy(), **tol)
        np.testing.assert_allclose(dg, tw.grad.numpy(), rtol=1e-5 if with_res else 1e-10, atol=1e-6 if with_res else 1e-12)
        np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C", [(5, 8), (25600, 384), (1000, 1024), (33, 200)])
@pytest.mark.parametrize("x_dtype,res_dtype,dy_dtype", [(torch.float32, None, torch.float32),
                                                       (torch.bfloat16, torch.float32, torch.float32),
                                                       (torch.bfloat16, torch.bfloat16, torch.bfloat16),
                                                       (torch.float32, torch.bfloat16, torch.bfloat16)])
def test_training_layer_norm_matches_oracle(rows, C, x_dtype, res_dtype, dy_dtype):
    """add_layer_norm_train: forward as the inference kernel, backward (k_layernorm_bwd) against the oracle on the
    same (rounded) inputs.  fp32 bar: ds to 1e-4 of its scale per element (bf16 outputs: one more rounding);
    dweight / dbias: sums over the rows, 1e-4 of their scale."""
    from asr_amd.layernorm import add_layer_norm_train

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).to(x_dtype)
    r = None if res_dtype is None else torch.randn(rows, C, generator=g).to(res_dtype)
    w = torch.randn(C, generator=g) * 0.5 + 1.0
    b = torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g).to(dy_dtype)
    xd = x.to(dev).requires_grad_(True)
    rd = None if r is None else r.to(dev).requires_grad_(True)
    wd, bd = torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev))
    y = add_layer_norm_train(xd, rd, wd, bd, 1e-5, dy_dtype)
    assert y.dtype == dy_dtype
    y.backward(dy.to(dev))
    ds, dg, db = lo.add_layer_norm_backward(x.float().numpy(), None if r is None else r.float().numpy(), w.numpy(),
                                            dy.float().numpy(), 1e-5)
    scale = np.abs(ds).max()

    def close(t, tol):
        return np.abs(t.float().cpu().double().numpy() - ds).max() <= tol * scale

    assert xd.grad.dtype == x_dtype and close(xd.grad, 1e-4 if x_dtype == torch.float32 else 2.0 ** -7)
    if r is not None:
        assert rd.grad.dtype == res_dtype and close(rd.grad, 1e-4 if res_dtype == torch.float32 else 2.0 ** -7)
    assert np.abs(wd.grad.cpu().double().numpy() - dg).max() <= 1e-4 * np.abs(dg).max() + 1e-5
    assert np.abs(bd.grad.cpu().double().numpy() - db).max() <= 1e-4 * np.abs(db).max() + 1e-5
    ref_y = lo.add_layer_norm(x.float().numpy(), None if r is None else r.float().numpy(), w.numpy(), b.numpy(), 1e-5)
    assert np.abs(y.detach().float().cpu().numpy() - ref_y).max() <= (1e-5 if dy_dtype == torch.float32 else 2.0 ** -7) * \
        np.abs(ref_y).max() + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C,dtype,with_res", [(25600, 384, torch.float32, False), (777, 200, torch.bfloat16, True)])
def test_training_layer_norm_with_dropout(rows, C, dtype, with_res):
    """Dropout in the layer norm's passes: the forward is the plain result times keep / (1 - p) for the oracle's mask
    of the same seeds, the backward that of the plain layer norm for dy times keep / (1 - p); the keep rate; the
    default seeds follow torch.manual_seed and the call count."""
    from asr_amd import layernorm as ln

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, C, generator=g).to(dtype)
    r = torch.randn(rows, C, generator=g).to(dtype) if with_res else None
    w, b = torch.randn(C, generator=g) * 0.5 + 1.0, torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g).to(dtype)
    p, seeds = 0.1, (77, 78)
    keep = lo.dropout_keep((rows, C), p, *seeds)
    assert abs(keep.mean() - 0.9) < 0.01
    scale = torch.from_numpy(keep.astype(np.float32) / (1.0 - p)).to(dev)

    def run(p_drop, upstream):
        xd = x.to(dev).requires_grad_(True)
        rd = None if r is None else r.to(dev).requires_grad_(True)
        wd, bd = torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev))
        y = ln.add_layer_norm_train(xd, rd, wd, bd, 1e-5, dtype, p_drop, seeds)
        y.backward(upstream)
        return y.detach(), xd.grad, None if rd is None else rd.grad, wd.grad, bd.grad

    y_d, dx_d, dr_d, dw_d, db_d = run(p, dy.to(dev))
    y_0, dx_0, dr_0, dw_0, db_0 = run(0.0, (dy.to(dev).float() * scale).to(dtype))
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -7
    want_y = y_0.float() * scale
    assert torch.equal(y_d == 0, want_y == 0) or abs(float((y_d == 0).float().mean()) - 0.1) < 0.01
    assert torch.allclose(y_d.float(), want_y, rtol=tol, atol=tol)
    # the backward sees dy * keep / (1 - p) in fp32; the comparison run rounded that product to the tensor dtype
    for got, want in ((dx_d, dx_0), (dr_d, dr_0), (dw_d, dw_0), (db_d, db_0)):
        if want is not None:
            assert torch.allclose(got.float(), want.float(), rtol=4 * tol, atol=4 * tol * float(want.float().abs().max()))
    # default seeds: torch's seed and a call counter
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    torch.manual_seed(5)
    ln._dropout_calls = 0
    a1, a2 = ln.add_layer_norm_train(xd, None, wd, bd, 1e-5, dtype, p), ln.add_layer_norm_train(xd, None, wd, bd, 1e-5, dtype, p)
    torch.manual_seed(5)
    ln._dropout_calls = 0
    b1 = ln.add_layer_norm_train(xd, None, wd, bd, 1e-5, dtype, p)
    assert torch.equal(a1, b1) and not torch.equal(a1, a2)
    with pytest.raises(ValueError):
        ln.add_layer_norm_train(xd, None, wd, bd, 1e-5, dtype, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_training_layer_norm_twin_output(p):
    """The float32 result and its bfloat16 copy from one pass; the gradients arriving through the two are added
    inside the backward kernel: same results as one output fed the sum (to the rounding of that sum)."""
    from asr_amd import layernorm as ln

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    rows, C = 3000, 384
    x = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(dev)
    w, b = (torch.randn(C, generator=g) * 0.5 + 1.0).to(dev), torch.randn(C, generator=g).to(dev)
    d32 = torch.randn(rows, C, generator=g).to(dev)
    d16 = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(dev)
    seeds = (3, 4)

    def run(twin, use):
        xd = x.clone().requires_grad_(True)
        wd, bd = torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())
        out = ln.add_layer_norm_train(xd, None, wd, bd, 1e-5, torch.float32, p, seeds, twin=twin)
        if twin:
            y, y16 = out
            assert y16.dtype == torch.bfloat16 and torch.equal(y16, y.to(torch.bfloat16))
            torch.autograd.backward([t for t, u in ((y, use[0]), (y16, use[1])) if u],
                                    [d for d, u in ((d32, use[0]), (d16, use[1])) if u])
        else:
            y = out
            y.backward((d32 if use[0] else 0) + (d16.float() if use[1] else 0))
        return y.detach(), xd.grad, wd.grad, bd.grad

    for use in ((True, True), (True, False), (False, True)):
        got, want = run(True, use), run(False, use)
        assert torch.equal(got[0], want[0])
        for a, c in zip(got[1:], want[1:]):
            assert torch.allclose(a.float(), c.float(), rtol=2.0 ** -7, atol=1e-5 * float(c.float().abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("defer", [True, False])
def test_training_layer_norm_writes_into_gradient_buckets(defer, monkeypatch):
    """Parameters of a GradBuckets: dgamma / dbeta land in the bucket's slices; with DEFER_PARAM_SUMS the sums of the
    workgroups' partial results of ALL layer norms of the step are one launch made by finish() -- the same bits as
    the immediate sums, and as the gradients without buckets; a second use of a layer norm in the step adds."""
    import asr_amd.layernorm as lnm
    from asr_amd.layernorm import add_layer_norm_train
    from asr_amd.parallel import GradBuckets

    monkeypatch.setattr(lnm, "DEFER_PARAM_SUMS", defer)
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    lns = [torch.nn.LayerNorm(384).to(dev) for _ in range(3)]
    for ln in lns:
        torch.nn.init.normal_(ln.weight, 1.0, 0.1)
        torch.nn.init.normal_(ln.bias, 0.0, 0.1)
    x = torch.randn(700, 384, device=dev)
    res = torch.randn(700, 384, device=dev)

    def loss(mods):
        y = add_layer_norm_train(x, res, mods[0].weight, mods[0].bias)
        y = add_layer_norm_train(y, None, mods[1].weight, mods[1].bias)
        y = add_layer_norm_train(y, x, mods[2].weight, mods[2].bias)
        y = add_layer_norm_train(y, None, mods[1].weight, mods[1].bias)  # second use of the middle one
        return (y * torch.linspace(-1, 1, 384, device=dev)).sum()

    ref = [torch.nn.LayerNorm(384).to(dev) for _ in range(3)]
    for a, b in zip(ref, lns):
        a.load_state_dict(b.state_dict())
    loss(ref).backward()
    buckets = GradBuckets([list(ln.parameters()) for ln in reversed(lns)], overlap=False)
    for _ in range(2):
        buckets.zero_grad()
        buckets.arena.fill_(55.0)
        loss(lns).backward()
        assert bool(lnm._PENDING_PARAM_SUMS) == defer
        buckets.finish()
        assert not lnm._PENDING_PARAM_SUMS
        for a, b in zip(ref, lns):
            assert torch.allclose(b.weight.grad, a.weight.grad, rtol=1e-6, atol=1e-6 * float(a.weight.grad.abs().max()))
            assert torch.allclose(b.bias.grad, a.bias.grad, rtol=1e-6, atol=1e-6 * float(a.bias.grad.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("with_res", [False, True])
def test_training_layer_norm_masks_the_input_gradient_of_a_relu_output(with_res):
    """relu_input: dx comes back multiplied by (x > 0); dresidual, the parameter gradients and the output do not change."""
    from asr_amd import layernorm as ln

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    rows, C = 1000, 384
    x = torch.relu(torch.randn(rows, C, generator=g)).to(torch.bfloat16).to(dev)
    r = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(dev) if with_res else None
    w, b = (torch.randn(C, generator=g) * 0.5 + 1.0).to(dev), torch.randn(C, generator=g).to(dev)
    dy = torch.randn(rows, C, generator=g).to(dev)

    def run(relu_input):
        xd = x.clone().requires_grad_(True)
        rd = None if r is None else r.clone().requires_grad_(True)
        wd, bd = torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())
        y = ln.add_layer_norm_train(xd, rd, wd, bd, 1e-5, torch.float32, 0.1, (5, 6), relu_input=relu_input)
        y.backward(dy)
        return y.detach(), xd.grad, None if rd is None else rd.grad, wd.grad, bd.grad

    plain, masked = run(False), run(True)
    assert torch.equal(plain[0], masked[0])
    assert torch.equal(masked[1], plain[1] * (x > 0))
    assert float(plain[1][x <= 0].abs().max()) > 0  # the mask did something
    for a, c in zip(plain[2:], masked[2:]):
        assert (a is None and c is None) or torch.equal(a, c)
