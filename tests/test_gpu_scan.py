"""GPU parity of the state-space ("scan") route of the bf16 causal convolution (the conv-only entry
points lmufft_conv_forward_scan_bf16 / lmufft_conv_backward_scan_bf16 and the layer with
conv_algo="scan") against the fp64 oracle of src/lmu.py:303-312 and against the Toeplitz kernels.

The route evaluates the same convolution from other operands (powers of A and chunk states instead
of H), each rounded to bf16 once, so its results differ from the Toeplitz kernels' by bf16 rounding
noise: both are held to the bf16 bar |err| <= 2e-2 (|ref| + 2 termnorm) against the oracle on the
un-rounded H, and the numpy evaluation of the route from the same constants (tests/scan_common.py)
pins the kernels' arithmetic more tightly."""
import numpy as np
import pytest
import torch

from oracle import lmufft_oracle as orc
from tests.scan_common import round_bf16, scan_adjoint_numpy, scan_forward_numpy, scan_sections

pytestmark = pytest.mark.gpu

SHAPES = [
    (3, 300, 192, 1000.0),   # ragged last chunk (44 frames)
    (2, 128, 192, 1000.0),   # exactly one chunk: no state at all
    (2, 129, 64, 50.0),      # a last chunk of one frame
    (5, 90, 16, 40.0),       # shorter than a chunk
    (4, 1000, 64, 300.0),
    (2, 700, 8, 20.0),       # the smallest memory
    (1, 2048, 256, 1000.0),  # the largest memory, whole chunks only
    (33, 640, 192, 1000.0),  # more rows than one 128-row tile, batch not a multiple of anything
    (1, 16384, 192, 1000.0), # the longest sequence the layer's other long-sequence route takes: 128 chunks
]


def _setup(M, theta, T, dev):
    from asr_amd.functional import prepare_scan_bf16
    from asr_amd.lmu import impulse_response, legendre_state_space

    A, B = legendre_state_space(M, theta)
    H, _ = impulse_response(A, B, T)
    return prepare_scan_bf16(A, B, device=dev), H.numpy().astype(np.float64), H


@pytest.mark.parametrize("B,T,M,theta", SHAPES)
def test_scan_conv_forward_and_adjoint(B, T, M, theta):
    from asr_amd import _lib
    from asr_amd.functional import causal_conv_adjoint_scan_bf16, causal_conv_scan_bf16

    dev = torch.device("cuda:0")
    sp, H, _ = _setup(M, theta, T, dev)
    L = _lib.lib.lmufft_scan_chunk()
    sec = scan_sections(sp.cpu().numpy(), M, L)
    rng = np.random.default_rng(B * 1000 + T + M)
    u = np.maximum(rng.standard_normal((B, T)), 0.0).astype(np.float32)
    gm = round_bf16(rng.standard_normal((B, T, M)))

    m = causal_conv_scan_bf16(torch.from_numpy(u).to(dev), sp, M).float().cpu().numpy().astype(np.float64)
    # (1) the kernels against the numpy evaluation of the same route: one bf16 rounding of the result,
    #     fp32 accumulation, the state rounded from fp32 instead of fp64
    emu = scan_forward_numpy(u.astype(np.float64), sec, M, L)
    long_seq = T > 4096  # the all-positive evaluation that scales the element-wise bars grows like 10^chunks
    if long_seq:
        assert np.sqrt(((m - emu) ** 2).mean()) <= 2.0 ** -8 * np.sqrt((emu ** 2).mean())
    else:
        absemu = scan_forward_numpy(np.abs(u).astype(np.float64), {**sec, "Wc": np.abs(sec["Wc"]), "AL": np.abs(sec["AL"]),
                                                                  "Hs": np.abs(sec["Hs"])}, M, L)
        bad = np.abs(m - emu) > 2.0 ** -8 * np.abs(emu) + 2.0 ** -9 * 1e-1 * absemu + 1e-30
        assert bad.mean() < 1e-3, (float(bad.mean()), float(np.abs(m - emu).max()))
    # (2) the bf16 bar against the oracle on the un-rounded impulse response
    ub = round_bf16(u)
    ref = orc.causal_conv_direct(ub, H) if T <= 1000 else orc.causal_conv_fft(ub, H)
    term = np.sqrt(np.maximum(orc.causal_conv_direct(ub ** 2, H ** 2) if T <= 1000 else orc.causal_conv_fft(ub ** 2, H ** 2), 0))
    err = np.abs(m - ref)
    # (beyond 1000 frames the oracle is its FFT form, whose own round-off is ~1e-16 of the largest value: the floor)
    floor = 1e-30 if T <= 1000 else 1e-12 * np.abs(ref).max()
    assert np.all(err <= 2e-2 * (np.abs(ref) + 2 * term) + floor), float((err / (2e-2 * (np.abs(ref) + 2 * term) + floor)).max())
    assert np.sqrt((err ** 2).mean()) <= 5e-3 * np.sqrt((ref ** 2).mean())

    gu = causal_conv_adjoint_scan_bf16(torch.from_numpy(gm).to(dev).bfloat16(), sp).cpu().numpy().astype(np.float64)
    gemu = scan_adjoint_numpy(gm, sec, M, L)
    if long_seq:
        assert np.sqrt(((gu - gemu) ** 2).mean()) <= 1e-5 * np.sqrt((gemu ** 2).mean())
    else:
        gabs = scan_adjoint_numpy(np.abs(gm), {**sec, "Wc": np.abs(sec["Wc"]), "ALT": np.abs(sec["ALT"]),
                                               "HsT": np.abs(sec["HsT"])}, M, L)
        assert np.all(np.abs(gu - gemu) <= 1e-5 * np.abs(gemu) + 4e-6 * gabs + 1e-30), float(np.abs(gu - gemu).max())
    gref = orc.anticausal_corr_direct(gm, H) if T <= 1000 else orc.anticausal_corr_fft(gm, H)
    gterm = np.sqrt(np.maximum(orc.anticausal_corr_direct(gm ** 2, H ** 2) if T <= 1000 else orc.anticausal_corr_fft(gm ** 2, H ** 2), 0))
    gerr = np.abs(gu - gref)
    gfloor = 1e-30 if T <= 1000 else 1e-12 * np.abs(gref).max()
    assert np.all(gerr <= 2e-2 * (np.abs(gref) + 2 * gterm) + gfloor)
    assert np.sqrt((gerr ** 2).mean()) <= 5e-3 * np.sqrt((gref ** 2).mean())


def test_scan_route_properties_at_full_size():
    """The long-utterance shape (batch 32, 3000 frames, 192 channels): impulse -> impulse response,
    adjoint identity <conv u, g> = <u, adj g>, agreement with the Toeplitz kernels, run-to-run bits."""
    from asr_amd.functional import (causal_conv_adjoint_bf16, causal_conv_adjoint_scan_bf16, causal_conv_bf16,
                                    causal_conv_scan_bf16)

    dev = torch.device("cuda:0")
    B, T, M = 32, 3000, 192
    sp, H, Ht = _setup(M, 1000.0, T, dev)
    Ht = Ht.to(dev)
    u = torch.zeros(B, T, device=dev)
    u[:, 0] = 1.0
    u[5, 0] = 0.0
    u[5, 1777] = 1.0  # an impulse in the middle of a chunk
    m = causal_conv_scan_bf16(u, sp, M).float()
    want = Ht.t()
    assert (m[0] - want).abs().max() <= 2.0 ** -7 * want.abs().max()
    assert (m[5, 1777:] - want[:T - 1777]).abs().max() <= 2.0 ** -7 * want.abs().max() and m[5, :1777].abs().max() == 0
    g = torch.Generator(device=dev).manual_seed(7)
    u = torch.randn(B, T, device=dev, generator=g).clamp_min(0)
    gm = torch.randn(B, T, M, device=dev, generator=g).bfloat16()
    m = causal_conv_scan_bf16(u, sp, M)
    gu = causal_conv_adjoint_scan_bf16(gm, sp)
    lhs = (m.double() * gm.double()).sum()
    rhs = (u.double() * gu.double()).sum()
    scale = (m.double().abs() * gm.double().abs()).sum()
    assert abs(lhs - rhs) <= 1e-4 * scale, (float(lhs), float(rhs), float(scale))
    m_t = causal_conv_bf16(u, Ht)
    gu_t = causal_conv_adjoint_bf16(gm, Ht)
    assert (m.float() - m_t.float()).pow(2).mean().sqrt() <= 1e-2 * m_t.float().pow(2).mean().sqrt()
    assert (gu - gu_t).pow(2).mean().sqrt() <= 1e-2 * gu_t.pow(2).mean().sqrt()
    assert torch.equal(m, causal_conv_scan_bf16(u, sp, M)) and torch.equal(gu, causal_conv_adjoint_scan_bf16(gm, sp))


def test_layer_scan_equals_toeplitz():
    """The whole layer, forward and backward, with conv_algo="scan" against conv_algo="toeplitz" on the
    same bf16 inputs; "scan" on fp32 input and "scan" without the route's constants are rejected."""
    from asr_amd import LMUFFT
    from asr_amd.functional import lmufft_layer

    dev = torch.device("cuda:0")
    B, T, D, M, Hd = 4, 700, 64, 192, 96
    torch.manual_seed(3)
    layers = {}
    for algo in ("toeplitz", "scan"):
        torch.manual_seed(3)
        lay = LMUFFT(D, Hd, M, T, 1000.0, conv_algo=algo)
        torch.nn.init.xavier_uniform_(lay.W_u.weight)
        torch.nn.init.xavier_uniform_(lay.W_h.weight)
        layers[algo] = lay.to(dev)
    g = torch.Generator(device=dev).manual_seed(11)
    x0 = torch.randn(B, T, D, device=dev, generator=g).bfloat16()
    gh = torch.randn(B, T, Hd, device=dev, generator=g).bfloat16()
    out = {}
    for algo, lay in layers.items():
        x = x0.clone().requires_grad_(True)
        h, h_n = lay(x)
        (h.float() * gh.float()).sum().backward()
        out[algo] = [h.float(), h_n.float(), x.grad.float()] + [p.grad.float() for p in lay.parameters()]
    # The outputs of the two routes agree to bf16 rounding noise.  Their gradients are NOT compared with each other:
    # a few hidden units whose pre-activation lies within that noise of zero land on the other side of relu on the
    # other route, and each moves a whole term of gx / gW_h.  Instead every route is held, at the bf16 bar, to the
    # oracle evaluated on the bf16-rounded operands and on that route's own relu pattern.
    from tests.test_gpu_bf16 import _bf, _close, _termnorms

    for a, b, name in zip(out["scan"][:2], out["toeplitz"][:2], ("h", "h_n")):
        rms = b.pow(2).mean().sqrt()
        assert (a - b).pow(2).mean().sqrt() <= 2e-2 * rms + 1e-6, (name, float((a - b).pow(2).mean().sqrt()), float(rms))
    npy = lambda t: t.detach().float().cpu().numpy()  # noqa: E731
    for algo, lay in layers.items():
        sd = {k: npy(v) for k, v in lay.state_dict().items() if k != "fft_H"}
        ref = orc.lmufft_backward(npy(x0), sd["W_u.weight"], sd["W_u.bias"], _bf(sd["W_h.weight"]), sd["W_h.bias"],
                                  _bf(sd["H"]), npy(gh), None, h_mask=npy(out[algo][0]) > 0, conv="fft")
        tn = _termnorms(npy(x0), sd["W_u.weight"], sd["W_h.weight"], sd["H"], ref, conv="fft")
        for got, name in zip(out[algo], ("h", "h_n", "gx", "gw_u", "gb_u", "gW_h", "gb_h")):
            _close(npy(got).reshape(ref[name].shape), ref[name], f"{name} {algo}", tn[name])
    with pytest.raises(RuntimeError):
        layers["scan"](x0.float())
    lay = layers["scan"]
    with pytest.raises(RuntimeError, match="scanprep"):
        lmufft_layer(x0, lay.W_u.weight, lay.W_u.bias, lay.W_h.weight, lay.W_h.bias, lay.H, conv_algo="scan")


def test_long_sequence_layer_matches_reference_golden(golden_dir):
    """The layer at 2048 frames against the reference's own LMUFFT (tests/golden/long2048_f32.npz): fp32 on the
    route `auto` takes at this length (partitioned FFT) at the fp32 bar; bf16 on the state-space and on the
    Toeplitz route -- the forward at the bf16 bar, the gradients by their rms (rounding x to bf16 moves a few per
    cent of the encoder's relu decisions, each of which moves a whole term of gx; tests/test_gpu_c4.py holds the
    gradients to the bf16 bar on the kernels' own relu pattern)."""
    import os

    from asr_amd import lmufft_layer
    from asr_amd.functional import prepare_scan_bf16
    from asr_amd.lmu import legendre_state_space

    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(golden_dir, "long2048_f32.npz")))
    D, Hd, M, T, theta, B = (int(v) for v in g["meta"])
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)  # noqa: E731
    names = ("gx", "gw_u", "gb_u", "gW_h", "gb_h")

    def run(dt, algo, scanprep=None):
        x = t(g["x"], dt).requires_grad_(True)
        ps = [t(g[k]).requires_grad_(True) for k in ("w_u", "b_u", "W_h", "b_h")]
        h, h_n = lmufft_layer(x, *ps, t(g["H"]), conv_algo=algo, scanprep=scanprep)
        (h.float() * t(g["gh"], dt).float()).sum().backward()
        grads = [x.grad] + [p.grad for p in ps]
        return [v.detach().float().cpu().numpy().astype(np.float64) for v in [h, h_n] + grads]

    # fp32
    out = run(torch.float32, "auto")
    o = orc.lmufft_backward(g["x"], g["w_u"], g["b_u"], g["W_h"], g["b_h"], g["H"], g["gh"], None, conv="fft")
    ga_abs = float(np.abs(o["gu"] * (o["u"] > 0)).sum())
    cond = {"gw_u": ga_abs * float(np.abs(g["x"]).max()), "gb_u": ga_abs}
    for got, name in zip(out, ("h", "h_n") + names):
        ref = g[name].astype(np.float64)
        tol = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max() + 1e-6 * cond.get(name, 0.0)
        assert np.all(np.abs(got - ref) <= tol), (name, float(np.abs(got - ref).max()), float(np.abs(ref).max()))
    # bf16, both routes.  The chain that ties the kernels' numbers to the reference's at the stated rtol 2e-2:
    #   (1) h against the golden itself;
    #   (2) every gradient against the oracle evaluated on the bf16-rounded operands and on the KERNEL's ReLU
    #       patterns (hidden units and encoder), at the bf16 bar -- the patterns agree by construction;
    #   (3) the same against the oracle on the UN-rounded operands (the reference's arithmetic: the CPU suite holds
    #       that evaluation, on the reference's own patterns, to long2048_f32.npz at the fp32 bar), again on the
    #       kernel's patterns;
    #   (4) the patterns themselves: they differ from the reference's only on a few per cent of the units, all of
    #       them units whose pre-activation lies within bf16 rounding noise of zero;
    #   (5) gx of the golden itself on the frames whose patterns agree in everything gx[b, t] depends on: the
    #       encoder unit of the frame and the hidden units of the frame and of every LATER frame of the utterance
    #       (the adjoint convolution carries those into ga[b, t]).
    from tests.test_gpu_bf16 import _close, _termnorms

    A, Bm = legendre_state_space(M, float(theta))
    sp = prepare_scan_bf16(A, Bm, device=dev)
    xb, Wb, Hb, ghb = orc.round_bf16(g["x"]), orc.round_bf16(g["W_h"]), orc.round_bf16(g["H"]), orc.round_bf16(g["gh"])
    fwd = orc.lmufft_forward(xb, g["w_u"], g["b_u"], Wb, g["b_h"], g["H"], conv="fft")
    mx = np.concatenate([fwd["m"], xb.astype(np.float64)], axis=-1)
    term_h = np.sqrt((mx ** 2) @ (Wb.astype(np.float64) ** 2).T)
    fwd0 = orc.lmufft_forward(g["x"], g["w_u"], g["b_u"], g["W_h"], g["b_h"], g["H"], conv="fft")  # the reference's patterns
    z0 = fwd0["m"] @ g["W_h"][:, :M].T.astype(np.float64) + g["x"].astype(np.float64) @ g["W_h"][:, M:].T.astype(np.float64) + g["b_h"]
    a0 = g["x"].astype(np.float64) @ g["w_u"].T.astype(np.float64)[:, 0] + float(g["b_u"][0])
    term_a = np.sqrt((g["x"].astype(np.float64) ** 2) @ (g["w_u"].astype(np.float64) ** 2).T[:, 0])
    for algo, prep in (("scan", sp), ("toeplitz", None)):
        out = run(torch.bfloat16, algo, prep)
        ref_h = g["h"].astype(np.float64)
        assert np.all(np.abs(out[0] - ref_h) <= 2e-2 * (np.abs(ref_h) + 2 * term_h)), (algo, "h")          # (1)
        h_mask = out[0] > 0
        refk = orc.lmufft_backward(xb, g["w_u"], g["b_u"], Wb, g["b_h"], Hb, ghb, None, h_mask=h_mask, conv="fft")
        tn = _termnorms(xb, g["w_u"], Wb, Hb, refk, conv="fft")
        for got, name in zip(out[2:], names):                                                               # (2)
            _close(got, refk[name], f"{name} long2048 {algo}", tn[name])
        ref0 = orc.lmufft_backward(g["x"], g["w_u"], g["b_u"], g["W_h"], g["b_h"], g["H"], ghb, None, h_mask=h_mask,
                                   u_mask=refk["u"] > 0, conv="fft")
        for got, name in zip(out[2:], names):                                                               # (3)
            _close(got, ref0[name], f"{name} long2048 {algo} vs un-rounded", tn[name])
        hflip = h_mask != (z0 > 0)                                                                          # (4)
        uflip = (refk["u"] > 0) != (a0 > 0)
        assert hflip.mean() < 0.02 and uflip.mean() < 0.05, (algo, float(hflip.mean()), float(uflip.mean()))
        assert np.all(np.abs(z0[hflip]) <= 4e-2 * term_h[hflip]) and np.all(np.abs(a0[uflip]) <= 4e-2 * term_a[uflip])
        later = np.flip(np.maximum.accumulate(np.flip(hflip.any(axis=-1), axis=1), axis=1), axis=1)      # (5)
        same = ~uflip & ~later
        assert same.sum() >= 20, int(same.sum())
        gx_ref = g["gx"].astype(np.float64)
        err = np.abs(out[2] - gx_ref)[same]
        bound = (2e-2 * (np.abs(gx_ref) + 2 * tn["gx"]))[same]
        print(f"{algo}: {int(same.sum())} frames with the reference's patterns; worst err / bound {float((err / bound).max()):.2f}; "
              f"hidden flips {float(hflip.mean()):.4f}, encoder flips {float(uflip.mean()):.4f}")
        assert np.all(err <= bound), (algo, float((err / bound).max()))
