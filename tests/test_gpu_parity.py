"""GPU parity tests: HIP kernels (through the C ABI) against the CPU oracle and
against golden vectors produced by the reference.  fp32 bar: rtol 1e-4 (with an
absolute floor of 1e-5 x the tensor's largest magnitude, because elements that
cancel to ~0 have no meaningful relative error in fp32)."""
import os

import numpy as np
import pytest
import torch

from oracle import lmufft_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-4
ATOL_REL = 1e-5


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m 'not gpu' on CPU-only machines)")
    return torch.device("cuda:0")


def _close(got, ref, name, rtol=RTOL, atol_rel=ATOL_REL, cond=0.0):
    """|got-ref| <= rtol*|ref| + atol_rel*max|ref| + 1e-6*cond.

    `cond` is the sum of absolute values of the terms of a long cancelling
    reduction (gb_u = sum of ga over all frames can be 3000x smaller than
    sum|ga|): no fp32 summation order, the reference's included, is accurate
    beyond ~1e-7 of that."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = np.abs(got - ref)
    bound = atol_rel * scale + rtol * np.abs(ref) + 1e-6 * cond
    bad = err > bound
    assert not bad.any(), (
        f"{name}: {bad.sum()} of {bad.size} elements out of tolerance; max err {err.max():.3e} "
        f"at {np.unravel_index(err.argmax(), err.shape)}, scale {scale:.3e}")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def _rand_H(rng, M, T):
    # decaying, sign-alternating rows: same flavour as the Legendre impulse response
    t = np.arange(T)
    return (rng.standard_normal((M, T)) * np.exp(-t / max(T / 3, 1))[None, :]).astype(np.float32)


CONV_SHAPES = [
    (2, 16, 6, 16),      # tiny, scalar paths
    (3, 37, 5, 41),      # nothing divisible by 4, T < ldH
    (2, 100, 128, 100),  # reference smoke shape
    (2, 200, 192, 200),  # 4 s utterance after 2x downsampling
    (2, 400, 192, 400),  # 4 s utterance, 400 frames
    (1, 130, 200, 400),  # M > one 192-wide tile, T < ldH
]


@pytest.mark.parametrize("B,T,M,ldH", CONV_SHAPES)
def test_conv_forward_matches_oracle(B, T, M, ldH):
    from asr_amd import causal_conv

    dev = _dev()
    rng = np.random.default_rng(B * 1000 + T)
    u = np.maximum(rng.standard_normal((B, T)), 0).astype(np.float32)
    H = _rand_H(rng, M, ldH)
    got = causal_conv(_t(u, dev), _t(H, dev), "toeplitz").cpu().numpy()
    _close(got, orc.causal_conv_direct(u, H[:, :T]), "m")


@pytest.mark.parametrize("B,T,M,ldH", CONV_SHAPES)
def test_conv_backward_matches_oracle(B, T, M, ldH):
    from asr_amd import causal_conv_adjoint

    dev = _dev()
    rng = np.random.default_rng(B * 2000 + T)
    gm = rng.standard_normal((B, T, M)).astype(np.float32)
    H = _rand_H(rng, M, ldH)
    got = causal_conv_adjoint(_t(gm, dev), _t(H, dev), "toeplitz").cpu().numpy()
    _close(got, orc.anticausal_corr_direct(gm, H[:, :T]), "gu")


def test_conv_backward_toeplitz_is_deterministic():
    """The diagonal sums of the Toeplitz adjoint run in a fixed order: repeated calls agree bit for bit."""
    from asr_amd import causal_conv_adjoint

    dev = _dev()
    rng = np.random.default_rng(77)
    gm = _t(rng.standard_normal((8, 400, 192)).astype(np.float32), dev)
    H = _t(_rand_H(rng, 192, 400), dev)
    first = causal_conv_adjoint(gm, H, "toeplitz")
    for _ in range(5):
        assert torch.equal(causal_conv_adjoint(gm, H, "toeplitz"), first)


def test_conv_long_utterance_3000_frames():
    """The long-utterance shape (30 s audio: T = 3000, memory 192) on the Toeplitz kernels, against
    the oracle's FFT evaluation of the same sums (the direct double sum is O(T^2) on the CPU)."""
    from asr_amd import causal_conv, causal_conv_adjoint

    dev = _dev()
    rng = np.random.default_rng(3000)
    B, T, M = 2, 3000, 192
    u = np.maximum(rng.standard_normal((B, T)), 0).astype(np.float32)
    gm = rng.standard_normal((B, T, M)).astype(np.float32)
    H = _rand_H(rng, M, T)
    m = causal_conv(_t(u, dev), _t(H, dev)).cpu().numpy()
    _close(m, orc.causal_conv_fft(u.astype(np.float64), H.astype(np.float64)), "m")
    # adjoint through the identity <conv(v), gm> = <v, adj(gm)> for unit impulses v = e_s
    gu = causal_conv_adjoint(_t(gm, dev), _t(H, dev)).cpu().numpy()
    Hd = H.astype(np.float64)
    for s in (0, 1, 1499, 2999):
        want = np.einsum("btk,kt->b", gm[:, s:, :].astype(np.float64), Hd[:, : T - s])
        np.testing.assert_allclose(gu[:, s], want, rtol=1e-4, atol=1e-5 * np.abs(gu).max())


def test_conv_known_answers_full_size(golden_dir):
    """Size-independent properties at the benchmark's full size (B=64, T=400, M=192)."""
    from asr_amd import causal_conv, causal_conv_adjoint

    dev = _dev()
    H = _t(np.load(os.path.join(golden_dir, "buffers_theta1000_M192.npz"))["H400"], dev)
    B, T, M = 64, 400, 192
    g = torch.Generator(device="cpu").manual_seed(5)
    u = torch.relu(torch.randn(B, T, generator=g)).to(dev)
    v = torch.relu(torch.randn(B, T, generator=g)).to(dev)
    # impulse -> impulse response
    imp = torch.zeros(B, T, device=dev)
    imp[:, 0] = 1.0
    m_imp = causal_conv(imp, H)
    assert torch.equal(m_imp[7], H.t().contiguous())
    # shifted impulse -> shifted response, zeros before
    imp2 = torch.zeros(1, T, device=dev)
    imp2[0, 123] = 2.0
    m2 = causal_conv(imp2, H)[0]
    assert torch.count_nonzero(m2[:123]) == 0
    torch.testing.assert_close(m2[123:], 2.0 * H.t()[: T - 123], rtol=0, atol=0)
    # linearity
    a = causal_conv(u, H)
    b = causal_conv(v, H)
    ab = causal_conv(u + 0.5 * v, H)
    scale = float(ab.abs().max())
    assert float((ab - (a + 0.5 * b)).abs().max()) <= 2e-6 * scale
    # adjoint identity <conv(u), gm> == <u, adj(gm)>
    gm = torch.randn(B, T, M, generator=g).to(dev)
    lhs = (a.double() * gm.double()).sum().item()
    rhs = (u.double() * causal_conv_adjoint(gm, H).double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)
    # causality: changing u after frame t0 leaves m before t0 untouched
    u2 = u.clone()
    u2[:, 200:] = 0
    a2 = causal_conv(u2, H)
    assert torch.equal(a2[:, :200], a[:, :200])


GOLDEN_CASES = ["tiny_f64", "tiny_f32", "smoke100_f32", "cfg1a_f32", "cfg1c_f32"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_layer_matches_reference_golden(golden_dir, name):
    """Outputs and gradients of the HIP layer vs the reference's own LMUFFT."""
    from asr_amd import lmufft_layer

    dev = _dev()
    g = _load(golden_dir, name)
    x = _t(g["x"], dev).requires_grad_(True)
    w_u = _t(g["w_u"], dev).requires_grad_(True)
    b_u = _t(g["b_u"], dev).requires_grad_(True)
    W_h = _t(g["W_h"], dev).requires_grad_(True)
    b_h = _t(g["b_h"], dev).requires_grad_(True)
    H = _t(g["H"], dev)
    h, h_n = lmufft_layer(x, w_u, b_u, W_h, b_h, H)
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


LAYER_SHAPES = [
    # B, T, D, M, Hd, ldH
    (2, 16, 12, 6, 10, 16),
    (3, 37, 13, 5, 9, 41),      # all-odd sizes -> scalar load paths, ragged tiles
    (1, 1, 8, 4, 8, 4),         # single frame
    (2, 65, 80, 192, 384, 100),  # one frame past a 64-row tile
    (5, 129, 384, 192, 384, 200),
]


@pytest.mark.parametrize("B,T,D,M,Hd,ldH", LAYER_SHAPES)
@pytest.mark.parametrize("which", ["both", "h_only", "hn_only"])
def test_layer_matches_oracle(B, T, D, M, Hd, ldH, which):
    from asr_amd import lmufft_layer

    dev = _dev()
    rng = np.random.default_rng(B * 7 + T * 11 + D)
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    w_u = (rng.standard_normal((1, D)) / np.sqrt(D)).astype(np.float32)
    b_u = np.array([0.05], dtype=np.float32)
    W_h = (rng.standard_normal((Hd, M + D)) / np.sqrt(M + D)).astype(np.float32)
    b_h = (0.1 * rng.standard_normal(Hd)).astype(np.float32)
    H = _rand_H(rng, M, ldH)
    gh = rng.standard_normal((B, T, Hd)).astype(np.float32) if which != "hn_only" else None
    gh_n = rng.standard_normal((B, Hd)).astype(np.float32) if which != "h_only" else None
    ref = orc.lmufft_backward(x, w_u, b_u, W_h, b_h, H[:, :T],
                              gh if gh is not None else np.zeros((B, T, Hd)), gh_n)

    tx = _t(x, dev).requires_grad_(True)
    params = [_t(a, dev).requires_grad_(True) for a in (w_u, b_u, W_h, b_h)]
    h, h_n = lmufft_layer(tx, *params, _t(H, dev))
    _close(h.detach().cpu().numpy(), ref["h"], "h")
    _close(h_n.detach().cpu().numpy(), ref["h_n"], "h_n")
    loss = 0
    if gh is not None:
        loss = loss + (h * _t(gh, dev)).sum()
    if gh_n is not None:
        loss = loss + (h_n * _t(gh_n, dev)).sum()
    loss.backward()
    _close(tx.grad.cpu().numpy(), ref["gx"], "gx")
    ga_abs = float(np.abs(ref["gu"] * (ref["u"] > 0)).sum())
    conds = {"gw_u": ga_abs * float(np.abs(x).max()), "gb_u": ga_abs, "gW_h": 0.0, "gb_h": 0.0}
    for t, k in zip(params, ("gw_u", "gb_u", "gW_h", "gb_h")):
        _close(t.grad.cpu().numpy(), ref[k], k, cond=conds[k])


def test_module_state_dict_roundtrip_and_forward(golden_dir):
    from asr_amd import LMUFFT

    dev = _dev()
    g = _load(golden_dir, "cfg1a_f32")
    layer = LMUFFT(80, 384, 192, 400, 1000)
    sd = layer.state_dict()
    sd["W_u.weight"] = torch.from_numpy(g["w_u"])
    sd["W_u.bias"] = torch.from_numpy(g["b_u"])
    sd["W_h.weight"] = torch.from_numpy(g["W_h"])
    sd["W_h.bias"] = torch.from_numpy(g["b_h"])
    layer.load_state_dict(sd)
    layer = layer.to(dev)
    h, h_n = layer(_t(g["x"], dev))
    # H here is the module's own (built by the constructor, not injected)
    _close(h.detach().cpu().numpy(), g["h"], "h")
    assert torch.equal(h_n, h[:, -1, :])
    # shorter sequences than seq_len are accepted and equal the prefix (causality): bit for bit
    # on the Toeplitz route, to rounding when the shape policy picks different routes
    h_short, _ = layer(_t(g["x"][:, :150], dev))
    _close(h_short.detach().cpu().numpy(), h[:, :150].detach().cpu().numpy(), "h prefix")
    layer.conv_algo = "toeplitz"
    h_t, _ = layer(_t(g["x"], dev))
    h_short, _ = layer(_t(g["x"][:, :150], dev))
    assert torch.equal(h_short, h_t[:, :150])
    layer.conv_algo = "auto"
    # longer ones are rejected
    with pytest.raises(RuntimeError, match="exceeds"):
        layer(torch.zeros(1, 401, 80, device=dev))


def test_full_size_layer_properties(golden_dir):
    """Benchmark layer shape (B=64, T=400, 384/192): oracle on a slice,
    gradient identity on the whole."""
    from asr_amd import LMUFFT

    dev = _dev()
    torch.manual_seed(3)
    layer = LMUFFT(384, 384, 192, 400, 1000)
    torch.nn.init.xavier_uniform_(layer.W_u.weight)
    torch.nn.init.xavier_uniform_(layer.W_h.weight)
    torch.nn.init.zeros_(layer.W_u.bias)
    torch.nn.init.zeros_(layer.W_h.bias)
    layer = layer.to(dev)
    x = torch.randn(64, 400, 384, device=dev, requires_grad=True)
    h, h_n = layer(x)
    gh = torch.randn_like(h)
    (h * gh).sum().backward()
    # batch item 17 against the oracle (items are independent)
    b = 17
    sd = {k: v.detach().cpu().numpy() for k, v in layer.state_dict().items() if k != "fft_H"}
    ref = orc.lmufft_backward(x[b : b + 1].detach().cpu().numpy(), sd["W_u.weight"], sd["W_u.bias"],
                              sd["W_h.weight"], sd["W_h.bias"], sd["H"], gh[b : b + 1].cpu().numpy())
    _close(h[b : b + 1].detach().cpu().numpy(), ref["h"], "h[17]")
    _close(x.grad[b : b + 1].cpu().numpy(), ref["gx"], "gx[17]")
    # directional derivative: <grad_W, dW> == d/de loss(W + e dW) to first order,
    # evaluated with the kernels themselves at two step sizes (fp32-limited)
    gW = layer.W_h.weight.grad.double()
    gbh = layer.W_h.bias.grad.double()
    # bias gradient equals the sum of masked upstream gradient
    ref_gb = (gh.double() * (h > 0)).sum(dim=(0, 1))
    torch.testing.assert_close(gbh, ref_gb, rtol=1e-4, atol=1e-4 * float(ref_gb.abs().max()))
    # weight gradient against an fp64 torch evaluation from the kernel's own h mask
    with torch.no_grad():
        from asr_amd import causal_conv

        u = torch.relu(x.detach() @ layer.W_u.weight.t() + layer.W_u.bias).squeeze(-1)
        m = causal_conv(u, layer.H)
        mx = torch.cat([m, x.detach()], dim=-1).double().reshape(-1, 576)
        gz = (gh.double() * (h > 0)).reshape(-1, 384)
        ref_gW = gz.t() @ mx
    torch.testing.assert_close(gW, ref_gW, rtol=1e-4, atol=1e-5 * float(ref_gW.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gradient_sink_matches_autograd_accumulation(dtype):
    """Gradients written straight into a flat exchange bucket equal the ones autograd
    accumulates over two steps and a double backward within a step (to summation-order
    rounding: the fp32 Toeplitz adjoint sums its diagonals with LDS atomics)."""
    import copy

    from asr_amd import LMUFFT
    from asr_amd.parallel import GradBuckets

    dev = _dev()
    torch.manual_seed(7)
    a = LMUFFT(48, 40, 24, 64, 16).to(dev)
    b = copy.deepcopy(a)
    buckets = GradBuckets([b])
    x = torch.randn(3, 64, 48, device=dev).to(dtype)
    gh = torch.randn(3, 64, 40, device=dev).to(dtype)
    for step in range(2):
        a.zero_grad(set_to_none=True)
        buckets.zero_grad()
        # second step: two backwards accumulate -- legal only in accumulate mode, which keeps the
        # exchange back until finish()
        with buckets.accumulate():
            for rep in range(1 + step):
                a(x)[0].backward(gh)
                b(x)[0].backward(gh)
            buckets.finish()
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            _close(pb.grad.cpu().numpy(), pa.grad.cpu().numpy(), f"{n} step {step}", rtol=1e-5, atol_rel=1e-6)
            assert buckets.buckets[0].data_ptr() <= pb.grad.data_ptr() < buckets.buckets[0].data_ptr() + \
                buckets.buckets[0].numel() * 4


def test_deferred_weight_sums_match_the_immediate_ones(monkeypatch):
    """Three stacked bf16 layers whose weight-gradient slabs are summed by one launch at finish() give bit-identical
    gradients to the same layers summing their slabs inside their own backward."""
    import copy

    from asr_amd import LMUFFT
    from asr_amd import functional as F
    from asr_amd.parallel import GradBuckets

    dev = _dev()
    torch.manual_seed(11)
    sizes = [(48, 40), (40, 40), (40, 24)]
    ref = [LMUFFT(d, hd, 24, 96, 16).to(dev) for d, hd in sizes]
    x = torch.randn(5, 96, 48, device=dev).to(torch.bfloat16)
    gy = torch.randn(5, 96, 24, device=dev).to(torch.bfloat16)
    grads = {}
    for defer in (False, True):
        monkeypatch.setattr(F, "DEFER_WEIGHT_SUMS", defer)
        layers = copy.deepcopy(ref)
        buckets = GradBuckets(list(reversed(layers)))
        buckets.zero_grad()
        y = x
        for l in layers:
            y, _ = l(y)
        y.backward(gy)
        assert len(F._PENDING_WEIGHT_SUMS) == (3 if defer else 0)
        buckets.finish()
        assert not F._PENDING_WEIGHT_SUMS
        grads[defer] = [p.grad.clone() for l in layers for p in l.parameters()]
    for g0, g1 in zip(grads[False], grads[True]):
        assert torch.equal(g0, g1)


@pytest.mark.parametrize("memory", [24, 20])  # 20: not a multiple of 8 -- the backward forms the memory gradient
@pytest.mark.parametrize("chained", [False, True])
def test_fused_relu_backward_between_stacked_layers(chained, memory):
    """fuse_relu_backward(lower, upper): the upper layer masks its input gradient by (x > 0) and the lower layer skips
    its own masking pass -- every parameter gradient and the gradient of the stack's input are bit-identical to the
    unfused stack's.  `chained`: the layers joined through detached copies and one backward call per layer (the way
    the benchmark overlaps the exchange), else one autograd graph."""
    import copy

    from asr_amd import LMUFFT
    from asr_amd.lmu import fuse_relu_backward

    dev = _dev()
    torch.manual_seed(13)
    sizes = [(48, 64), (64, 64), (64, 32)]
    ref = [LMUFFT(d, hd, memory, 96, 16).to(dev) for d, hd in sizes]
    x0 = torch.randn(5, 96, 48, device=dev).to(torch.bfloat16)
    gy = torch.randn(5, 96, 32, device=dev).to(torch.bfloat16)
    results = {}
    for fused in (False, True):
        layers = copy.deepcopy(ref)
        if fused:
            for lo, up in zip(layers[:-1], layers[1:]):
                fuse_relu_backward(lo, up)
        x = x0.clone().requires_grad_(True)
        if chained:
            ins, outs, y = [], [], x
            for i, l in enumerate(layers):
                xin = y if i == 0 else y.detach().requires_grad_(True)
                y, _ = l(xin)
                ins.append(xin)
                outs.append(y)
            outs[-1].backward(gy)
            for k in range(len(layers) - 2, -1, -1):
                outs[k].backward(ins[k + 1].grad)
        else:
            y = x
            for l in layers:
                y, _ = l(y)
            y.backward(gy)
        results[fused] = [x.grad.clone()] + [p.grad.clone() for l in layers for p in l.parameters()]
        if fused:  # the upper layers masked by the bits the lower layers' forwards left, not by reading x
            assert all(l._mask_out._bits is not None for l in layers[:-1])
    assert any(float(g.abs().max()) > 0 for g in results[True])
    for g0, g1 in zip(results[False], results[True]):
        assert torch.equal(g0, g1)


def test_fused_relu_backward_falls_back_when_h_n_carries_gradient():
    """The lower layer of a fused pair runs its own masking pass when the gradient of its h_n is in use."""
    import copy

    from asr_amd import LMUFFT
    from asr_amd.lmu import fuse_relu_backward

    dev = _dev()
    torch.manual_seed(17)
    ref = [LMUFFT(48, 64, 24, 64, 16).to(dev), LMUFFT(64, 32, 24, 64, 16).to(dev)]
    x0 = torch.randn(3, 64, 48, device=dev).to(torch.bfloat16)
    results = {}
    for fused in (False, True):
        layers = copy.deepcopy(ref)
        if fused:
            fuse_relu_backward(layers[0], layers[1])
        x = x0.clone().requires_grad_(True)
        h, h_n = layers[0](x)
        y, _ = layers[1](h)
        (y.float().sum() + 3.0 * h_n.float().sum()).backward()
        results[fused] = [x.grad.clone()] + [p.grad.clone() for l in layers for p in l.parameters()]
    for g0, g1 in zip(results[False], results[True]):
        assert torch.equal(g0, g1)


def test_fuse_relu_backward_checks_its_layers():
    from asr_amd import LMUFFT
    from asr_amd.lmu import fuse_relu_backward

    a, b = LMUFFT(48, 64, 24, 64, 16), LMUFFT(40, 32, 24, 64, 16)
    with pytest.raises(ValueError):
        fuse_relu_backward(a, b)
    with pytest.raises(TypeError):
        fuse_relu_backward(a, torch.nn.Linear(4, 4))


def test_fused_relu_backward_without_the_bit_mask(monkeypatch):
    """The upper layer of a fused pair compares x itself when the lower layer left no bit mask (here: the mask is
    switched off) or when its x is not the tensor the mask was made for."""
    import copy

    from asr_amd import LMUFFT
    from asr_amd import functional as F
    from asr_amd.lmu import fuse_relu_backward

    dev = _dev()
    torch.manual_seed(19)
    ref = [LMUFFT(48, 64, 24, 64, 16).to(dev), LMUFFT(64, 32, 24, 64, 16).to(dev)]
    x0 = torch.randn(3, 64, 48, device=dev).to(torch.bfloat16)
    gy = torch.randn(3, 64, 32, device=dev).to(torch.bfloat16)
    results = {}
    for mode in ("plain", "bits", "no_bits", "copy"):
        monkeypatch.setattr(F, "USE_RELU_MASK", mode != "no_bits")
        layers = copy.deepcopy(ref)
        if mode != "plain":
            fuse_relu_backward(layers[0], layers[1])
        x = x0.clone().requires_grad_(True)
        h, _ = layers[0](x)
        if mode == "copy":  # same values at another address: the link's mask is not for this tensor
            h = h * 1.0
        y, _ = layers[1](h)
        y.backward(gy)
        results[mode] = [x.grad.clone()] + [p.grad.clone() for l in layers for p in l.parameters()]
    for mode in ("bits", "no_bits"):
        for g0, g1 in zip(results["plain"], results[mode]):
            assert torch.equal(g0, g1), mode
    # "copy": autograd's multiplication node sits between the layers; the numbers are still the same
    for g0, g1 in zip(results["plain"], results["copy"]):
        assert torch.equal(g0, g1)


def test_fused_relu_backward_on_the_state_space_route():
    """The same fold on a long sequence, where the convolution runs on the state-space route and the backward forms
    the memory gradient (no Q route): bit-identical gradients."""
    import copy

    from asr_amd import LMUFFT
    from asr_amd.lmu import fuse_relu_backward

    dev = _dev()
    torch.manual_seed(29)
    ref = [LMUFFT(64, 64, 64, 1100, 200.0, conv_algo="scan").to(dev), LMUFFT(64, 32, 64, 1100, 200.0).to(dev)]
    x0 = (0.5 * torch.randn(2, 1100, 64, device=dev)).to(torch.bfloat16)
    gy = torch.randn(2, 1100, 32, device=dev).to(torch.bfloat16)
    results = {}
    for fused in (False, True):
        layers = copy.deepcopy(ref)
        if fused:
            fuse_relu_backward(layers[0], layers[1])
        x = x0.clone().requires_grad_(True)
        y = x
        for l in layers:
            y, _ = l(y)
        y.backward(gy)
        results[fused] = [x.grad.clone()] + [p.grad.clone() for l in layers for p in l.parameters()]
    assert all(bool(torch.isfinite(g).all()) for g in results[True])
    for g0, g1 in zip(results[False], results[True]):
        assert torch.equal(g0, g1)


@pytest.mark.parametrize("share", [True, False])
def test_shared_q_preparation_gives_the_same_gradients(share, monkeypatch):
    """Layers whose gradient arrives masked get their backward's weights-only matrix made by the top layer's masking
    pass (or, when every layer is masked, by one launch for all of them): same gradients bit for bit as when every
    layer makes its own."""
    import copy

    from asr_amd import LMUFFT
    from asr_amd import functional as F
    from asr_amd.lmu import fuse_relu_backward

    dev = _dev()
    torch.manual_seed(37)
    sizes = [(48, 64), (64, 64), (64, 64), (64, 32)]
    ref = [LMUFFT(d, hd, 24, 96, 16).to(dev) for d, hd in sizes]
    x0 = torch.randn(4, 96, 48, device=dev).to(torch.bfloat16)
    gy = torch.randn(4, 96, 32, device=dev).to(torch.bfloat16)

    def run(fused, all_masked):
        layers = copy.deepcopy(ref)
        if fused:
            for lo, up in zip(layers[:-1], layers[1:]):
                fuse_relu_backward(lo, up)
        g = gy
        if all_masked:  # the top layer's gradient is masked by hand, and the layer is told so
            layers[-1].grad_is_masked = True
        x = x0.clone().requires_grad_(True)
        y = x
        for l in layers:
            y, _ = l(y)
        if all_masked:
            g = gy * (y > 0)
        y.backward(g)
        return [x.grad.clone()] + [p.grad.clone() for l in layers for p in l.parameters()]

    monkeypatch.setattr(F, "SHARE_QPREP", False)
    want = run(False, False)
    monkeypatch.setattr(F, "SHARE_QPREP", share)
    for all_masked in (False, True):
        got = run(True, all_masked)
        for g0, g1 in zip(want, got):
            assert torch.equal(g0, g1)
    if share:
        assert not [r for r in F._PENDING_Q.get(dev, []) if r() is not None and not r().done]
