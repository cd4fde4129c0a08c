"""CTC head: the oracle against the reference's golden vectors (CPU), the HIP kernels against
both (GPU).  Token lists must be identical; log-probabilities within fp32 rounding."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc_oracle as co


def _golden(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "ctc_greedy.npz")))


def _lists(tokens, counts):
    return [list(map(int, tokens[i, : int(counts[i])])) for i in range(len(counts))]


def test_ctc_oracle_matches_reference_golden(golden_dir):
    g = _golden(golden_dir)
    lp = co.log_probs(g["enc"], g["W"], g["b"])
    np.testing.assert_allclose(lp, g["log_probs"], rtol=1e-5, atol=1e-5)
    assert co.greedy_decode(g["log_probs"], g["lengths"]) == _lists(g["tokens"], g["counts"])
    assert co.greedy_decode(lp, g["lengths"]) == _lists(g["tokens"], g["counts"])
    # collapse rules on a hand-made path: blank = 3
    lp2 = np.full((1, 8, 4), -9.0)
    for t, k in enumerate([3, 1, 1, 3, 1, 2, 2, 3]):
        lp2[0, t, k] = 0.0
    assert co.greedy_decode(lp2, [8]) == [[1, 1, 2]]
    assert co.greedy_decode(lp2, [3]) == [[1]]
    assert co.greedy_decode(lp2, [0]) == [[]]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ctc_head_matches_reference_golden(golden_dir, dtype):
    from asr_amd import CTCDecoder

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    g = _golden(golden_dir)
    B, T, Hd = g["enc"].shape
    V = g["W"].shape[0]
    head = CTCDecoder(Hd, V)
    with torch.no_grad():
        head.ctc_projection.weight.copy_(torch.from_numpy(g["W"]))
        head.ctc_projection.bias.copy_(torch.from_numpy(g["b"]))
    head = head.to(dev)
    enc = torch.from_numpy(g["enc"]).to(dev).to(dtype)
    lengths = torch.from_numpy(g["lengths"]).to(dev)
    want = _lists(g["tokens"], g["counts"])
    if dtype == torch.float32:
        lp = head(enc).cpu().numpy()
        np.testing.assert_allclose(lp, g["log_probs"], rtol=1e-4, atol=2e-5)
        assert head.decode(enc, lengths) == want  # token-identical
        tokens, counts = head.greedy(enc, lengths)
        assert (tokens.cpu().numpy() == g["tokens"]).all() and (counts.cpu().numpy() == g["counts"]).all()
    else:
        # bf16 encoder output: identical to the oracle run on the same rounded input
        enc_r = enc.float().cpu().numpy()
        ref = co.greedy_decode(co.log_probs(enc_r, g["W"], g["b"]), g["lengths"])
        assert head.decode(enc, lengths) == ref


@pytest.mark.gpu
def test_ctc_head_full_batch_and_edges():
    """Batch 256 x 400 frames (the inference config): every utterance against the oracle."""
    from asr_amd import CTCDecoder

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, T, Hd, V = 256, 400, 384, 34
    head = CTCDecoder(Hd, V).to(dev)
    torch.nn.init.xavier_uniform_(head.ctc_projection.weight, gain=4.0)
    base = torch.randn(B, T // 5 + 1, Hd).repeat_interleave(5, dim=1)[:, :T]
    enc = (base + 0.2 * torch.randn(B, T, Hd)).to(dev)
    lengths = torch.randint(0, T + 1, (B,))
    lengths[0], lengths[1] = T, 0
    got = head.decode(enc, lengths.to(dev))
    lp = co.log_probs(enc.cpu().numpy(), head.ctc_projection.weight.detach().cpu().numpy(),
                      head.ctc_projection.bias.detach().cpu().numpy())
    assert got == co.greedy_decode(lp, lengths.numpy())
    assert got[1] == []
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        head(enc.cpu())
    with pytest.raises(RuntimeError, match="one entry per lane"):
        CTCDecoder(Hd, 100).to(dev)(enc)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,Hd,V", [
    (3, 37, 64, 20),    # matrix-pipe kernel, one vocabulary tile, ragged last frame tile
    (2, 50, 384, 34),   # matrix-pipe kernel, two vocabulary tiles
    (1, 1, 96, 64),     # a single frame, full 64-entry vocabulary
    (5, 33, 1024, 29),  # widest encoder the head accepts
    (2, 45, 100, 34),   # width not a multiple of 32: lane-per-entry kernel
])
def test_ctc_head_kernel_routes(B, T, Hd, V, dtype):
    """Both logits kernels against the oracle: log-probabilities to 1e-4, tokens identical."""
    from asr_amd import CTCDecoder

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 1000 + Hd + V)
    head = CTCDecoder(Hd, V).to(dev)
    torch.nn.init.xavier_uniform_(head.ctc_projection.weight, gain=3.0)
    torch.nn.init.normal_(head.ctc_projection.bias, std=0.5)
    enc = torch.randn(B, T, Hd).to(dev).to(dtype)
    lengths = torch.randint(0, T + 1, (B,))
    lengths[0] = T
    W = head.ctc_projection.weight.detach().cpu().numpy()
    b = head.ctc_projection.bias.detach().cpu().numpy()
    ref_lp = co.log_probs(enc.float().cpu().numpy(), W, b)
    lp = head(enc).cpu().numpy()
    np.testing.assert_allclose(lp, ref_lp, rtol=1e-4, atol=5e-5)
    assert head.decode(enc, lengths.to(dev)) == co.greedy_decode(ref_lp, lengths.numpy())


def _ctc_case(seed, B, T, V, Lmax, full_lengths=False):
    g = torch.Generator().manual_seed(seed)
    lp = torch.randn(B, T, V, generator=g).log_softmax(-1)
    targets = torch.randint(0, V - 1, (B, max(Lmax, 1)), generator=g)
    tl = torch.randint(0 if not full_lengths else Lmax, Lmax + 1, (B,), generator=g)
    il = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
    il[0], tl[0] = T, Lmax
    return lp, targets, il, tl


def test_ctc_loss_oracle_matches_torch():
    """The alpha-beta restatement against torch's CTC loss and autograd on the CPU: the library the reference calls
    (nn.CTCLoss(blank=V-1, reduction='mean', zero_infinity=True), src/models/asr_model.py:105)."""
    for seed, (B, T, V, Lmax) in enumerate([(3, 12, 6, 4), (4, 30, 34, 9), (2, 8, 5, 6)]):
        lp, targets, il, tl = _ctc_case(seed, B, T, V, Lmax)
        if seed == 2:
            il[1], tl[1] = 3, 6  # no valid alignment: zero_infinity turns the term and its gradient into 0
        x = lp.double().clone().requires_grad_(True)
        packed = torch.cat([targets[i, : int(tl[i])] for i in range(B)])
        ref = torch.nn.functional.ctc_loss(x.transpose(0, 1), packed, il, tl, blank=V - 1, reduction="mean",
                                           zero_infinity=True)
        ref.backward()
        loss, grad = co.ctc_loss_and_grad(lp.double().numpy(), targets.numpy(), il.numpy(), tl.numpy(), V - 1, True)
        want = float(ref.detach())
        assert abs(loss - want) <= 1e-10 * max(1.0, abs(want))
        np.testing.assert_allclose(grad, x.grad.numpy(), rtol=1e-8, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,V,Lmax", [(1, 1, 3, 0), (3, 12, 6, 4), (64, 400, 34, 60), (5, 50, 34, 63), (4, 300, 34, 64),
                                       (2, 120, 10, 200)])
def test_ctc_loss_kernel_matches_oracle(B, T, V, Lmax):
    """k_ctc_loss through the C ABI (loss and gradient in one launch) against the oracle: fp32 log-domain recursions
    over up to 400 frames: loss rtol 1e-5, gradient |err| <= 1e-4 * max|grad| + 1e-7."""
    from asr_amd.ctc import ctc_loss

    dev = torch.device("cuda:0")
    lp, targets, il, tl = _ctc_case(B * 7 + T, B, T, V, Lmax)
    if B >= 3:
        il[1], tl[1] = max(1, Lmax // 2), Lmax        # too few frames: infinite term, zeroed
        targets[2, : max(Lmax, 1)] = targets[2, 0]    # repeated labels need blanks in between
    x = lp.to(dev).requires_grad_(True)
    loss = ctc_loss(x, targets.to(dev), il.to(dev), tl.to(dev), V - 1, True)
    (loss * 3.0).backward()
    want, grad = co.ctc_loss_and_grad(lp.double().numpy(), targets.numpy(), il.numpy(), tl.numpy(), V - 1, True)
    assert abs(float(loss.detach()) - want) <= 1e-5 * max(1.0, abs(want)), (float(loss.detach()), want)
    got = x.grad.cpu().double().numpy() / 3.0
    assert np.abs(got - grad).max() <= 1e-4 * np.abs(grad).max() + 1e-7
    # rows past an utterance's end carry no gradient; rows inside sum to zero (gradient through the log-softmax)
    for b in range(B):
        assert (got[b, int(il[b]):] == 0).all()
    assert np.abs(got.sum(-1)).max() <= 1e-5 * np.abs(grad).max() + 1e-7


@pytest.mark.gpu
def test_ctc_loss_kernel_keeps_tiny_probabilities():
    """Log-probabilities far below fp32's exp range (-87) on every alignment's way: the label the utterance needs
    has probability e^-100 .. e^-120 in every frame (the linear-domain recursion rescales every fourth frame by the
    largest state, so what must fit a double is four such frames in a row and the spread of a frame's states).  The loss is large but finite and its gradient is useful (torch's
    log-domain kernel agrees); computed with exp in fp32 the paths vanished, the term became infinite and
    zero_infinity zeroed loss and gradient."""
    from asr_amd.ctc import ctc_loss

    dev = torch.device("cuda:0")
    B, T, V, Lmax = 4, 40, 34, 6
    lp, targets, il, tl = _ctc_case(5, B, T, V, Lmax, full_lengths=True)
    il[:] = T
    logits = torch.randn(B, T, V, generator=torch.Generator().manual_seed(6))
    # utterance 0: one label repeated Lmax times needs Lmax label frames and Lmax - 1 blanks among 40 frames; with
    # only 10 frames every frame is on some needed path, and in all of them the label and the blank are crushed
    il[0] = 2 * Lmax - 1
    targets[0, :] = 3
    logits[0, :, 3] -= 120.0
    logits[0, :, V - 1] -= 100.0
    # utterance 1: every label crushed by -100 in every frame (its complete paths lie e^-600 below the all-blank
    # prefix, which sets the scale of a frame's states: inside the e^-700 a double gives the linear-domain recursion)
    for j in range(Lmax):
        logits[1, :, int(targets[1, j])] -= 100.0
    lp = logits.log_softmax(-1)
    assert float(lp.min()) < -110
    x = lp.to(dev).requires_grad_(True)
    loss = ctc_loss(x, targets.to(dev), il.to(dev), tl.to(dev), V - 1, True)
    loss.backward()
    want, grad = co.ctc_loss_and_grad(lp.double().numpy(), targets.numpy(), il.numpy(), tl.numpy(), V - 1, True)
    assert np.isfinite(want) and want > 80  # the crushed utterances dominate the mean
    assert abs(float(loss.detach()) - want) <= 1e-5 * abs(want), (float(loss.detach()), want)
    got = x.grad.cpu().double().numpy()
    assert np.abs(got[:2]).max() > 0  # not zeroed
    assert np.abs(got - grad).max() <= 1e-4 * np.abs(grad).max() + 1e-7


@pytest.mark.gpu
def test_ctc_loss_kernel_matches_torch_on_the_gpu_and_is_deterministic():
    from asr_amd.ctc import ctc_loss

    dev = torch.device("cuda:0")
    lp, targets, il, tl = _ctc_case(99, 16, 200, 34, 40, full_lengths=False)
    tl = tl.clamp(min=1)
    x = lp.to(dev).requires_grad_(True)
    a = ctc_loss(x, targets.to(dev), il.to(dev), tl.to(dev), 33, True)
    a.backward()
    y = lp.to(dev).requires_grad_(True)
    b = torch.nn.functional.ctc_loss(y.transpose(0, 1), targets.to(dev), il, tl, blank=33, reduction="mean", zero_infinity=True)
    b.backward()
    assert abs(float(a.detach()) - float(b.detach())) <= 1e-5 * abs(float(b.detach()))
    # torch's CUDA kernel runs the recursion in fp32 (about 1e-3 of relative error in a posterior after 200 frames);
    # the tight bound is the float64 oracle's, in test_ctc_loss_kernel_matches_oracle
    assert torch.allclose(x.grad, y.grad, rtol=1e-3, atol=1e-3 * float(y.grad.abs().max()))
    g1 = x.grad.clone()
    x.grad = None
    ctc_loss(x, targets.to(dev), il.to(dev), tl.to(dev), 33, True).backward()
    assert torch.equal(g1, x.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,V,ld", [((3, 7), 34, 40), ((64, 400), 34, 40), ((5,), 29, 32), ((2, 3), 64, 64), ((9,), 1, 8)])
def test_log_softmax_rows_matches_torch(rows, V, ld, dtype):
    """The log-softmax kernel around the CTC loss against torch.log_softmax(logits[..., :V].float(), -1) in float64,
    forward and backward; padding columns are ignored forward and come back as zeros."""
    from asr_amd.ctc import log_softmax_rows

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(V + ld)
    x = (4 * torch.randn(*rows, ld, device=dev, generator=g)).to(dtype).requires_grad_(True)
    dlp = torch.randn(*rows, V, device=dev, generator=g)
    lp = log_softmax_rows(x, V)
    assert lp.dtype == torch.float32 and lp.shape == (*rows, V)
    lp.backward(dlp)
    xr = x.detach().double()[..., :V].requires_grad_(True)
    ref = torch.log_softmax(xr, dim=-1)
    ref.backward(dlp.double())
    assert (lp.detach().double() - ref.detach()).abs().max() <= 2e-6 * (1 + ref.detach().abs().max())
    assert x.grad.dtype == dtype and x.grad.shape == x.shape
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert (x.grad.double()[..., :V] - xr.grad).abs().max() <= tol * (1e-3 + xr.grad.abs().max())
    assert not x.grad[..., V:].any()


@pytest.mark.gpu
def test_linear_train_keeps_the_padding_columns_when_asked():
    """keep_padding: the vocabulary projection returns its kernel's padded width and takes the gradient back with it:
    the same numbers as the sliced form."""
    from asr_amd.linear import linear_train

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    x = torch.randn(6, 50, 384, device=dev).bfloat16()
    w = torch.nn.Parameter(torch.randn(34, 384, device=dev) / 20)
    b = torch.nn.Parameter(torch.randn(34, device=dev))
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = linear_train(xa, w, b)
    yb = linear_train(xb, w, b, keep_padding=True)
    assert ya.shape[-1] == 34 and yb.shape[-1] == 40 and torch.equal(ya, yb[..., :34]) and not yb[..., 34:].any()
    dy = torch.randn_like(ya)
    ya.backward(dy)
    ga, gw, gb = xa.grad.clone(), w.grad.clone(), b.grad.clone()
    w.grad = b.grad = None
    yb.backward(torch.nn.functional.pad(dy, (0, 6)))
    assert torch.equal(ga, xb.grad) and torch.equal(gw, w.grad) and torch.equal(gb, b.grad)
