"""Kernel-level parity of the bf16 causal convolution and its adjoint (the conv-only entry points
of the C ABI) against the fp64 direct sums of the oracle evaluated on the bf16-rounded operands
(src/lmu.py:303-312 and its autograd adjoint).  Shapes on both sides of the short-sequence
kernels' limit (T <= 416) and of their tile / range / slice boundaries."""
import numpy as np
import pytest
import torch

from oracle import lmufft_oracle as orc

pytestmark = pytest.mark.gpu


def _rand_H(rng, M, T):
    # decaying, sign-alternating rows like the Legendre impulse response (no scipy needed at size)
    t = np.arange(T)[None, :]
    k = np.arange(M)[:, None]
    return (np.cos(0.013 * (k + 1) * t + 0.3 * k) * np.exp(-t / (0.6 * T + 5.0)) * 0.04).astype(np.float32)


SHAPES = [
    (2, 16, 8, 16),      # one partial tile
    (3, 33, 32, 40),     # two tiles, the first with one frame; ldH > T
    (5, 100, 64, 100),   # the reference's smoke length
    (2, 200, 192, 200),  # the benchmark's post-downsample length
    (4, 400, 192, 400),  # the benchmark's training shape
    (2, 416, 96, 416),   # the short-sequence kernels' limit
    (3, 400, 128, 400),  # channel count that is not a multiple of the 96-channel block
    (2, 300, 72, 320),   # channels a multiple of 8 only
    (2, 420, 64, 420),   # just beyond the limit: the streaming kernels
    (1, 1200, 192, 1200),
]


@pytest.mark.parametrize("B,T,M,ldH", SHAPES)
def test_conv_bf16_forward_and_adjoint(B, T, M, ldH):
    from asr_amd.functional import causal_conv_adjoint_bf16, causal_conv_bf16, prepare_impulse_bf16

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(B * 1000 + T + M)
    H = _rand_H(rng, M, ldH)
    u = np.maximum(rng.standard_normal((B, T)), 0.0).astype(np.float32)  # relu output: about half zeros
    gm = rng.standard_normal((B, T, M)).astype(np.float32)
    Ht = torch.from_numpy(H).to(dev)
    hprep = prepare_impulse_bf16(Ht)
    Hb = orc.round_bf16(H).astype(np.float64)
    ub = orc.round_bf16(u).astype(np.float64)
    gmb = orc.round_bf16(gm).astype(np.float64)

    m = causal_conv_bf16(torch.from_numpy(u).to(dev), Ht, hprep).float().cpu().numpy().astype(np.float64)
    ref = orc.causal_conv_direct(ub, Hb[:, :T])
    absref = orc.causal_conv_direct(np.abs(ub), np.abs(Hb[:, :T]))
    # one bf16 rounding of the result (2^-9 relative) + fp32 accumulation noise on the terms
    tol = 2.0 ** -8 * np.abs(ref) + 2e-6 * absref + 1e-30
    bad = np.abs(m - ref) > tol
    assert not bad.any(), (np.argwhere(bad)[:5], np.abs(m - ref).max(), np.abs(ref).max())

    gu = causal_conv_adjoint_bf16(torch.from_numpy(gm).to(dev).bfloat16(), Ht, hprep).cpu().numpy().astype(np.float64)
    gref = orc.anticausal_corr_direct(gmb, Hb[:, :T])
    gabs = orc.anticausal_corr_direct(np.abs(gmb), np.abs(Hb[:, :T]))
    tol = 1e-5 * np.abs(gref) + 2e-6 * gabs + 1e-30
    bad = np.abs(gu - gref) > tol
    assert not bad.any(), (np.argwhere(bad)[:5], np.abs(gu - gref).max(), np.abs(gref).max())


def test_conv_bf16_impulse_and_determinism():
    """u = delta at frame 0 returns the (bf16-rounded) impulse response itself, bit for bit; two
    runs of the adjoint on the same input agree bit for bit (fixed summation order)."""
    from asr_amd.functional import causal_conv_adjoint_bf16, causal_conv_bf16

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    B, T, M = 64, 400, 192
    H = _rand_H(rng, M, T)
    Ht = torch.from_numpy(H).to(dev)
    u = torch.zeros(B, T, device=dev)
    u[:, 0] = 1.0
    m = causal_conv_bf16(u, Ht)
    want = Ht.t().bfloat16()  # [T, M]
    assert torch.equal(m[0], want) and torch.equal(m[B - 1], want)
    gm = torch.randn(B, T, M, device=dev).bfloat16()
    a = causal_conv_adjoint_bf16(gm, Ht)
    b = causal_conv_adjoint_bf16(gm, Ht)
    assert torch.equal(a, b)


def test_conv_bf16_adjoint_large_batch_of_long_sequences():
    """Many utterances of a long sequence: the adjoint kernel's frame ranges must stay short enough for their
    window of H^T to fit the LDS however few ranges the grid would need to fill the chip (batch 256 x 3000 frames
    used to be refused).  The result per utterance does not depend on the batch it is computed in beyond the
    summation order of the ranges: compared with the same utterances run eight at a time."""
    from asr_amd.functional import causal_conv_adjoint_bf16, prepare_impulse_bf16

    dev = torch.device("cuda:0")
    B, T, M = 256, 3000, 192
    rng = np.random.default_rng(7)
    H = torch.from_numpy(_rand_H(rng, M, T)).to(dev)
    hprep = prepare_impulse_bf16(H)
    g = torch.Generator(device=dev).manual_seed(11)
    gm = torch.randn(B, T, M, device=dev, generator=g).bfloat16()
    gu = causal_conv_adjoint_bf16(gm, H, hprep=hprep)
    assert gu.shape == (B, T) and torch.isfinite(gu).all()
    for b0 in (0, 120, 248):
        part = causal_conv_adjoint_bf16(gm[b0:b0 + 8].contiguous(), H, hprep=hprep)
        scale = float(part.abs().max())
        assert float((gu[b0:b0 + 8] - part).abs().max()) <= 2e-5 * scale
