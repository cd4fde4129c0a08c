"""Linear layers of the encoder (bf16 inference): the oracle against torch's linear on the CPU,
and the HIP kernel (through the C ABI) against the oracle, with and without the fused
residual + layer-norm epilogue."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import layernorm_oracle as lno  # noqa: E402
from oracle import linear_oracle as lo  # noqa: E402


def _case(rows, K, N, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    return x, W, b


def test_round_bf16_matches_torch():
    g = torch.Generator().manual_seed(3)
    a = torch.randn(4096, generator=g) * 10.0 ** torch.randint(-6, 6, (4096,), generator=g).float()
    want = a.to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(lo.round_bf16(a.numpy()), want)


@pytest.mark.parametrize("rows,K,N", [(5, 80, 384), (33, 384, 1152), (1, 8, 8)])
def test_oracle_matches_torch_linear(rows, K, N):
    x, W, b = _case(rows, K, N, rows + K + N)
    xb, Wb, bb = x.to(torch.bfloat16), W.to(torch.bfloat16), b.to(torch.bfloat16)
    # the op the reference runs under autocast, evaluated exactly: fp64 product of the bf16 operands
    want = (xb.double() @ Wb.double().T + bb.double()).float().to(torch.bfloat16).float().numpy()
    got = lo.linear_bf16(x.numpy(), W.numpy(), b.numpy())
    np.testing.assert_array_equal(got, want)
    # and torch's own bf16 linear on the CPU, to the bf16 bar (its accumulation order is its own)
    cpu = torch.nn.functional.linear(xb, Wb, bb).float().numpy()
    np.testing.assert_allclose(got, cpu, rtol=2e-2, atol=2e-2)


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    return torch.device("cuda:0")


def _assert_bf16_close(got, want):
    np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-2)  # bf16 bar
    # the kernel accumulates in fp32 and rounds once: apart from near-ties it IS the oracle's value
    assert np.mean(got == want) > 0.97
    assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-3)) < 1e-2  # at most one bf16 ulp


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K,N,with_bias", [
    (1, 8, 8, True), (127, 80, 384, True), (129, 384, 384, True), (300, 384, 1152, True), (64, 72, 40, False),
    (1000, 512, 512, True), (257, 200, 392, True), (33001, 384, 384, True), (20000, 96, 1152, True)])
def test_kernel_matches_oracle(rows, K, N, with_bias):
    from asr_amd.linear import linear_bf16

    dev = _gpu()
    x, W, b = _case(rows, K, N, rows * 7 + K + N)
    bq = b.to(torch.bfloat16).float()
    got = linear_bf16(x.to(dev).to(torch.bfloat16), W.to(dev).to(torch.bfloat16), bq.to(dev) if with_bias else None)
    assert got.dtype == torch.bfloat16 and got.shape == (rows, N)
    want = lo.linear_bf16(x.numpy(), W.numpy(), b.numpy() if with_bias else None)
    _assert_bf16_close(got.float().cpu().numpy(), want)


@pytest.mark.gpu
def test_kernel_keeps_leading_dimensions_and_empty_input():
    from asr_amd.linear import linear_bf16

    dev = _gpu()
    x, W, b = _case(6 * 50, 384, 384, 11)
    xb = x.view(6, 50, 384).to(dev).to(torch.bfloat16)
    got = linear_bf16(xb, W.to(dev).to(torch.bfloat16), b.to(dev))
    assert got.shape == (6, 50, 384)
    flat = linear_bf16(xb.view(300, 384), W.to(dev).to(torch.bfloat16), b.to(dev))
    assert torch.equal(got.view(300, 384), flat)
    empty = linear_bf16(xb[:0], W.to(dev).to(torch.bfloat16), b.to(dev))
    assert empty.shape == (0, 50, 384)


@pytest.mark.gpu
@pytest.mark.parametrize("res_dtype", [None, torch.float32, torch.bfloat16])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,K,N", [(1, 8, 8), (130, 384, 384), (515, 384, 256), (77, 80, 200), (40000, 384, 384),
                                     (300, 512, 384)])
def test_fused_layer_norm_matches_oracle(rows, K, N, res_dtype, out_dtype):
    from asr_amd.linear import linear_add_layer_norm_bf16

    dev = _gpu()
    x, W, b = _case(rows, K, N, rows + 3 * K + N)
    g = torch.Generator().manual_seed(rows)
    res = None if res_dtype is None else (torch.randn(rows, N, generator=g) * 2.0).to(res_dtype)
    gamma, beta = torch.randn(N, generator=g), torch.randn(N, generator=g)
    bq = b.to(torch.bfloat16).float()
    y, proj = linear_add_layer_norm_bf16(
        x.to(dev).to(torch.bfloat16), W.to(dev).to(torch.bfloat16), bq.to(dev), None if res is None else res.to(dev),
        gamma.to(dev), beta.to(dev), 1e-5, out_dtype=out_dtype, return_projection=True)
    assert y.dtype == out_dtype and y.shape == (rows, N)
    # the projection: the bf16 bar against the oracle
    _assert_bf16_close(proj.float().cpu().numpy(), lo.linear_bf16(x.numpy(), W.numpy(), b.numpy()))
    # the epilogue: layer norm of THAT projection plus the residual, at the fp32 bar
    want = lno.add_layer_norm(proj.float().cpu().numpy(), None if res is None else res.float().numpy(),
                              gamma.numpy(), beta.numpy(), 1e-5)
    got = y.float().cpu().numpy()
    if out_dtype == torch.float32:
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5)
    else:
        np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-2)
        assert np.mean(got == torch.from_numpy(want).to(torch.bfloat16).float().numpy()) > 0.99
    # end to end against the oracle of the whole block, at the bf16 bar of its bf16 intermediate
    full = lo.linear_add_layer_norm(x.numpy(), W.numpy(), b.numpy(), None if res is None else res.float().numpy(),
                                    gamma.numpy(), beta.numpy(), 1e-5)
    np.testing.assert_allclose(got, full, rtol=2e-2, atol=5e-2)
    # without the projection output the result is the same
    y2 = linear_add_layer_norm_bf16(
        x.to(dev).to(torch.bfloat16), W.to(dev).to(torch.bfloat16), bq.to(dev), None if res is None else res.to(dev),
        gamma.to(dev), beta.to(dev), 1e-5, out_dtype=out_dtype)
    assert torch.equal(y, y2)


@pytest.mark.gpu
def test_fused_layer_norm_in_place_on_the_residual():
    from asr_amd import _lib
    from asr_amd.linear import linear_add_layer_norm_bf16
    import ctypes

    dev = _gpu()
    x, W, b = _case(200, 384, 384, 5)
    g = torch.Generator().manual_seed(9)
    res = torch.randn(200, 384, generator=g)
    gamma, beta = torch.randn(384, generator=g), torch.randn(384, generator=g)
    xb, Wb = x.to(dev).to(torch.bfloat16), W.to(dev).to(torch.bfloat16)
    want = linear_add_layer_norm_bf16(xb, Wb, b.to(dev), res.to(dev), gamma.to(dev), beta.to(dev))
    buf = res.to(dev).clone()
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    gd, bd, bias = gamma.to(dev), beta.to(dev), b.to(dev)
    rc = _lib.lib.lmufft_linear_ln_bf16(p(xb), p(Wb), p(bias), 200, 384, 384, None, p(buf), _lib.F32, p(gd), p(bd),
                                        1e-5, p(buf), _lib.F32, None)
    _lib.check(rc, "lmufft_linear_ln_bf16")
    torch.cuda.synchronize()
    assert torch.equal(buf, want)


@pytest.mark.gpu
def test_errors():
    from asr_amd.linear import linear_add_layer_norm_bf16, linear_bf16

    dev = _gpu()
    x = torch.zeros(4, 16, device=dev, dtype=torch.bfloat16)
    W = torch.zeros(8, 16, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        linear_bf16(x.cpu(), W.cpu())
    with pytest.raises(RuntimeError, match="bfloat16"):
        linear_bf16(x.float(), W)
    with pytest.raises(RuntimeError, match="does not match"):
        linear_bf16(x[:, :8], W)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        linear_bf16(x[:, :12].contiguous(), W[:, :12].contiguous())
    Wide = torch.zeros(392, 16, device=dev, dtype=torch.bfloat16)
    ones = torch.ones(392, device=dev)
    with pytest.raises(RuntimeError, match="exceeds"):
        linear_add_layer_norm_bf16(x, Wide, None, None, ones, ones)


def test_oracle_backward_matches_torch_autograd():
    """The backward restatement against torch autograd on the CPU (fp64 on the bf16-rounded operands)."""
    rng = np.random.default_rng(4)
    x, w, dy = rng.standard_normal((37, 24)), rng.standard_normal((16, 24)), rng.standard_normal((37, 16))
    dx, dW, db = lo.linear_backward_bf16(x, w, dy)
    tx = torch.tensor(lo.round_bf16(x), dtype=torch.float64, requires_grad=True)
    tw = torch.tensor(lo.round_bf16(w), dtype=torch.float64, requires_grad=True)
    tb = torch.zeros(16, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.linear(tx, tw, tb).backward(torch.tensor(lo.round_bf16(dy), dtype=torch.float64))
    np.testing.assert_allclose(dW, tw.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-12)
    assert np.abs(dx - tx.grad.numpy()).max() <= 2.0 ** -8 * np.abs(tx.grad.numpy()).max()  # one bf16 rounding


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K,N,with_bias", [(64, 8, 8, True), (25600, 384, 384, True), (1000, 80, 384, False),
                                                (777, 384, 1152, True), (130, 200, 40, True),
                                                (2000, 384, 34, True), (300, 64, 5, False),
                                                (1, 8, 8, True), (33, 24, 16, True), (95, 384, 384, False)])
def test_training_linear_matches_oracle(rows, K, N, with_bias):
    """linear_train (forward, input gradient, weight and bias gradient through the HIP kernels) against the oracle:
    y and dx at the bf16 bar of the forward tests, dW and db (fp32 results of bf16 operands) to 1e-4 of their
    scale."""
    from asr_amd.linear import linear_train

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(rows + K)
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if with_bias else None
    dy = rng.standard_normal((rows, N)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    wt = torch.nn.Parameter(torch.from_numpy(w).to(dev))
    bt = torch.nn.Parameter(torch.from_numpy(b).to(dev)) if with_bias else None
    y = linear_train(xt, wt, bt)
    assert y.dtype == torch.bfloat16
    y.backward(torch.from_numpy(dy).to(dev).to(torch.bfloat16))
    ref_y = lo.linear_bf16(x, w, b)
    dx, dW, db = lo.linear_backward_bf16(x, w, dy)
    got_y = y.detach().float().cpu().numpy()
    assert (np.abs(got_y - ref_y) <= 2.0 ** -7 * np.abs(ref_y) + 1e-6).all()
    got_dx = xt.grad.cpu().numpy()
    assert xt.grad.dtype == torch.float32
    assert (np.abs(got_dx - dx) <= 2.0 ** -7 * np.abs(dx) + 1e-6 * np.abs(dx).max()).all()
    got_dW = wt.grad.cpu().double().numpy()
    assert wt.grad.dtype == torch.float32 and np.abs(got_dW - dW).max() <= 1e-4 * np.abs(dW).max()
    if with_bias:
        assert np.abs(bt.grad.cpu().double().numpy() - db).max() <= 1e-4 * np.abs(db).max()


@pytest.mark.gpu
@pytest.mark.parametrize("side_stream", [False, True, "sums", "immediate"])
def test_training_linear_writes_into_gradient_buckets(side_stream, monkeypatch):
    """Parameters of a GradBuckets: the weight-gradient kernel writes into the bucket, `.grad` IS the bucket's slice
    (no clearing between steps, no accumulation kernel), a second use in the same step adds; the padded
    vocabulary width takes the copy route and lands in the bucket as well.  With ASYNC_WEIGHT_GRAD the bucket-bound
    gradients run on the side stream (joined by finish(), and before a second use's gradient is accumulated)."""
    import asr_amd.linear as lin
    from asr_amd.linear import linear_train
    from asr_amd.parallel import GradBuckets

    monkeypatch.setattr(lin, "ASYNC_WEIGHT_GRAD", side_stream is True)
    monkeypatch.setattr(lin, "ASYNC_SLAB_SUMS", side_stream == "sums")  # only the slab sums on the side stream
    # default (False): the slab sums of the bucket-bound gradients wait for ONE launch in finish(); "immediate": each
    # weight gradient sums its own slabs at once
    monkeypatch.setattr(lin, "DEFER_SLAB_SUMS", side_stream is False)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    a, head = torch.nn.Linear(64, 48).to(dev), torch.nn.Linear(48, 34).to(dev)
    buckets = GradBuckets([list(head.parameters()), list(a.parameters())], overlap=False)
    x = torch.randn(200, 64, device=dev)
    want = None
    for step in range(2):
        buckets.zero_grad()
        assert a.weight.grad is None
        buckets.arena.fill_(123.0)  # what the last step left behind must not leak into this one
        h = linear_train(x, a.weight, a.bias)
        (linear_train(h, head.weight, head.bias).float().sum() + linear_train(2 * x, a.weight, a.bias).float().sum()).backward()
        buckets.finish()
        for p, v in zip(buckets._params[1] + buckets._params[0], buckets._views[1] + buckets._views[0]):
            assert p.grad.data_ptr() == v.data_ptr()
        got = [p.grad.clone() for p in (a.weight, a.bias, head.weight, head.bias)]
        if want is None:
            # the same arithmetic without buckets
            a2, head2 = torch.nn.Linear(64, 48).to(dev), torch.nn.Linear(48, 34).to(dev)
            a2.load_state_dict(a.state_dict())
            head2.load_state_dict(head.state_dict())
            h2 = linear_train(x, a2.weight, a2.bias)
            (linear_train(h2, head2.weight, head2.bias).float().sum()
             + linear_train(2 * x, a2.weight, a2.bias).float().sum()).backward()
            want = [p.grad for p in (a2.weight, a2.bias, head2.weight, head2.bias)]
        for g, w in zip(got, want):
            assert torch.allclose(g, w, rtol=1e-6, atol=1e-6 * float(w.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("adjacent", [True, False])
def test_fused_linear_train_matches_the_separate_layers(adjacent):
    """w_q, w_k, w_v as one product: outputs bitwise those of the three layers (the same kernel on column blocks),
    the input gradient one K = 3C product instead of three rounded ones (bf16 bar), the weight and bias gradients
    those of the separate layers to fp32 accumulation order; with the slices back to back in a gradient bucket the
    kernel writes them in place, otherwise they arrive by copy."""
    from asr_amd.linear import fused_linear_train, linear_train
    from asr_amd.parallel import GradBuckets

    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    C, rows = 384, 1500
    layers = [torch.nn.Linear(C, C).to(dev) for _ in range(3)]
    ref = [torch.nn.Linear(C, C).to(dev) for _ in range(3)]
    for a, b in zip(layers, ref):
        b.load_state_dict(a.state_dict())
    order = [l.weight for l in layers] + [l.bias for l in layers] if adjacent else [p for l in layers for p in l.parameters()]
    buckets = GradBuckets([order], overlap=False)
    x = torch.randn(rows, C, device=dev)
    go = torch.randn(rows, 3 * C, device=dev).to(torch.bfloat16)
    buckets.zero_grad()
    xa = x.clone().requires_grad_(True)
    ya = fused_linear_train(xa, layers)
    ya.backward(go)
    buckets.finish()
    xb = x.clone().requires_grad_(True)
    yb = torch.cat([linear_train(xb, l.weight, l.bias) for l in ref], dim=-1)
    yb.backward(go)
    assert torch.equal(ya, yb)
    assert torch.allclose(xa.grad, xb.grad, rtol=2.0 ** -6, atol=2.0 ** -6 * float(xb.grad.abs().max()))
    for a, b, views in zip(layers, ref, range(3)):
        assert torch.allclose(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-4 * float(b.weight.grad.abs().max()))
        assert torch.allclose(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-4 * float(b.bias.grad.abs().max()))
    lo, hi = buckets.arena.data_ptr(), buckets.arena.data_ptr() + buckets.arena.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for l in layers for p in l.parameters())


@pytest.mark.gpu
def test_prepare_weights_matches_the_per_layer_preparation():
    """One launch for several layers' training operands: the same bits as a launch per weight, handed out once and
    only to a call with the very same parameter objects."""
    from asr_amd import linear as L

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    mk = lambda n, k: (torch.nn.Parameter(torch.randn(n, k, device=dev)), torch.nn.Parameter(torch.randn(n, device=dev)))
    (wq, bq), (wk, bk), (wv, bv) = mk(384, 384), mk(384, 384), mk(384, 384)
    (wo, bo), (wi, bi), (wc, _) = mk(384, 384), mk(384, 80), mk(29, 384)
    groups = [([wi], [bi]), ([wq, wk, wv], [bq, bk, bv]), ([wo], [bo]), ([wc], None)]
    L._PREPARED.clear()
    ref = [L.linear_weight_prep(w, b) for w, b in groups]  # a launch per weight
    L.prepare_weights(groups)
    assert len(L._PREPARED) == len(groups)
    for (w, b), r in zip(groups, ref):
        got = L.linear_weight_prep(w, b)  # from the table: no launch
        for a, e in zip(got, r):
            assert (a is None) == (e is None)
            if a is not None:
                assert a.shape == e.shape and torch.equal(a, e)
    assert not L._PREPARED  # handed out once
    # other parameter objects under the same first weight: not taken from the table
    L.prepare_weights([([wq, wk, wv], [bq, bk, bv])])
    assert L._take_prepared([wq, wk], [bq, bk]) is None and not L._PREPARED
    # a vocabulary projection (29 rows) is padded to 32 with zero rows
    w16, w16t, b32 = ref[3]
    assert w16.shape == (32, 384) and w16t.shape == (384, 32) and b32 is None
    assert torch.equal(w16[:29], wc.detach().to(torch.bfloat16)) and not w16[29:].any()
