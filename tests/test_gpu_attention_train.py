"""GPU parity tests of the attention core's training form (k_attention_fwd with dropout and saved row
statistics, k_attention_bwd_dq, k_attention_bwd_dkv) through the C ABI, against oracle/attention_oracle.py
(pinned to torch autograd of the reference's formula by tests/test_attention_oracle.py).

Bar: fp32 tensors: |got - ref| <= 1e-4 * |ref| + 1e-5 * max|ref| per output (fp32 accumulation over up to T keys).
bf16 tensors (fp32 arithmetic inside, bf16 rounding of inputs and outputs): the oracle runs on the bf16-rounded
inputs; |got - ref| <= 2e-2 * |ref| + 2e-2 * rms(ref).  With dropout the oracle applies the SAME keep mask
(the hash is restated in the oracle); torch's own dropout stream cannot be matched by another implementation."""
import numpy as np
import pytest
import torch

from oracle import attention_oracle as ao

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    return torch.device("cuda:0")


def _close32(got, ref, name):
    got = got.detach().float().cpu().double().numpy()
    tol = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max()
    err = np.abs(got - ref)
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


def _close16(got, ref, name):
    got = got.detach().float().cpu().double().numpy()
    tol = 2e-2 * np.abs(ref) + 2e-2 * np.sqrt((ref ** 2).mean())
    err = np.abs(got - ref)
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


CASES = [
    # B, T, H, dk, lengths, p
    (2, 33, 2, 32, None, 0.0),
    (3, 100, 3, 64, [100, 57, 1], 0.0),
    (2, 160, 2, 64, [160, 99], 0.1),
    (2, 50, 1, 128, [50, 20], 0.25),
    (1, 400, 6, 64, [317], 0.1),
]


@pytest.mark.parametrize("B,T,H,dk,lengths,p", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_and_backward_match_oracle(B, T, H, dk, lengths, p, dtype):
    from asr_amd.attention import masked_attention, masked_attention_train

    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + T)
    mk = lambda: torch.randn(B, T, H * dk, generator=g).to(dtype)  # noqa: E731
    q, k, v, go = mk(), mk(), mk(), mk()
    # padded rows of the upstream gradient are not zeroed: their (uniform) weights pass gradient to v
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    ln = None if lengths is None else torch.tensor(lengths, device=dev)
    seeds = (20240611, 77)
    out = masked_attention_train(qd, kd, vd, H, ln, p, seeds)
    out.backward(go.to(dev))
    keep = ao.dropout_keep_mask(B, H, T, p, *seeds) if p else None
    ref, cache = ao.attention_train(q.double().numpy(), k.double().numpy(), v.double().numpy(), lengths, H, keep, p)
    rq, rk, rv = ao.attention_backward(go.double().numpy(), cache)
    close = _close32 if dtype == torch.float32 else _close16
    close(out, ref, "context")
    close(qd.grad, rq, "dq")
    close(kd.grad, rk, "dk")
    close(vd.grad, rv, "dv")
    if p == 0.0:  # without dropout the training forward is the inference kernel's result
        inf = masked_attention(q.to(dev), k.to(dev), v.to(dev), H, ln)
        assert torch.allclose(out.detach().float(), inf.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-5,
                              atol=2e-2 if dtype == torch.bfloat16 else 1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_packed_form_is_the_same_computation(dtype):
    """q | k | v as column blocks of one tensor (one projection GEMM in front): context and gradients are bitwise
    those of the three-tensor form, the gradient arrives as one tensor of the packed layout."""
    from asr_amd.attention import masked_attention_train, masked_attention_train_packed

    dev = _dev()
    B, T, H, dk = 3, 130, 6, 64
    C = H * dk
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, T, 3 * C, generator=g).to(dtype).to(dev)
    go = torch.randn(B, T, C, generator=g).to(dtype).to(dev)
    ln = torch.tensor([130, 64, 9], device=dev)
    seeds = (11, 12)
    a = qkv.clone().requires_grad_(True)
    out_p = masked_attention_train_packed(a, H, ln, 0.1, seeds)
    out_p.backward(go)
    q, k, v = (qkv[..., i * C:(i + 1) * C].clone().requires_grad_(True) for i in range(3))
    out = masked_attention_train(q, k, v, H, ln, 0.1, seeds)
    out.backward(go)
    assert torch.equal(out_p, out)
    assert a.grad.shape == qkv.shape
    for i, t in enumerate((q, k, v)):
        assert torch.equal(a.grad[..., i * C:(i + 1) * C], t.grad), "qkv"[i]
    with pytest.raises(RuntimeError, match="3 \\* channels"):
        masked_attention_train_packed(qkv[..., :-1], H)


def test_dropout_properties():
    """Keep rate, seeds name the mask, the default seeds follow torch.manual_seed and the call count, and
    masked positions receive no gradient."""
    from asr_amd import attention as at

    dev = _dev()
    torch.manual_seed(3)
    B, T, H, dk = 2, 128, 2, 32
    v = torch.eye(T, device=dev).repeat(B, 1, 1)[:, :, : H * dk].contiguous()  # context = weights[:, :, :64] per head
    q = torch.zeros(B, T, H * dk, device=dev)
    k = torch.zeros(B, T, H * dk, device=dev)
    # uniform weights 1/T; after dropout an element is 0 or 1/(T (1 - p))
    p = 0.2
    out = at.masked_attention_train(q, k, v, H, None, p, (5, 6))
    kept = (out[:, :, :dk] > 0).float().mean().item()
    assert abs(kept - 0.8) < 0.02
    assert torch.allclose(out[out > 0], torch.full_like(out[out > 0], 1.0 / (T * 0.8)), rtol=1e-5)
    keep = ao.dropout_keep_mask(B, H, T, p, 5, 6)
    assert np.array_equal((out[:, :, :dk] > 0).cpu().numpy(), keep[:, 0, :, :dk])
    assert torch.equal(out, at.masked_attention_train(q, k, v, H, None, p, (5, 6)))
    assert not torch.equal(out, at.masked_attention_train(q, k, v, H, None, p, (5, 7)))
    torch.manual_seed(3)
    at._dropout_calls = 0
    a1, a2 = at.masked_attention_train(q, k, v, H, None, p), at.masked_attention_train(q, k, v, H, None, p)
    torch.manual_seed(3)
    at._dropout_calls = 0
    b1 = at.masked_attention_train(q, k, v, H, None, p)
    assert torch.equal(a1, b1) and not torch.equal(a1, a2)
    # masked_fill semantics: queries and keys past the end get no gradient through the scores
    qd = torch.randn(B, T, H * dk, device=dev, requires_grad=True)
    kd = torch.randn(B, T, H * dk, device=dev, requires_grad=True)
    vd = torch.randn(B, T, H * dk, device=dev, requires_grad=True)
    ln = torch.tensor([T, 40], device=dev)
    at.masked_attention_train(qd, kd, vd, H, ln, 0.0).sum().backward()
    assert (qd.grad[1, 40:] == 0).all() and (kd.grad[1, 40:] == 0).all()
    assert (vd.grad[1, 40:] != 0).any()  # rows past the end attend uniformly to all positions, padding included


def test_errors():
    from asr_amd.attention import masked_attention_train

    dev = _dev()
    x = torch.zeros(1, 8, 64, device=dev)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        masked_attention_train(x.cpu(), x.cpu(), x.cpu(), 2)
    with pytest.raises(RuntimeError, match="head widths"):
        masked_attention_train(x, x, x, 4)
    with pytest.raises(ValueError):
        masked_attention_train(x, x, x, 2, None, 1.0)
