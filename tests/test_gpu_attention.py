"""GPU parity tests of the masked attention core (k_attention_fwd) through the C ABI against
oracle/attention_oracle.py.  fp32: rtol 1e-4 with an absolute floor of 1e-5 x max|ref|; bf16
inputs/outputs (fp32 arithmetic): the oracle on the rounded inputs, output rounding 2^-8."""
import numpy as np
import pytest
import torch

from oracle import attention_oracle as ao

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    return torch.device("cuda:0")


CASES = [
    # B, T, heads, d_k, lengths
    (2, 40, 2, 32, None),
    (3, 100, 4, 32, [100, 57, 1]),
    (2, 129, 3, 64, [129, 64]),
    (2, 400, 6, 64, [400, 333]),
    (1, 200, 2, 128, [150]),
    (2, 70, 2, 64, [0, 70]),       # an empty utterance: every row attends uniformly
    (1, 1, 1, 32, [1]),
    (2, 128, 2, 64, None),         # exactly four wave tiles: the four-wave workgroup
    (1, 257, 2, 64, [257]),        # eight-wave workgroups, the second with one query
    (2, 520, 2, 128, [520, 300]),  # eight waves at the widest head
]


@pytest.mark.parametrize("B,T,H,dk,lengths", CASES)
def test_attention_matches_oracle(B, T, H, dk, lengths):
    from asr_amd.attention import masked_attention

    dev = _dev()
    rng = np.random.default_rng(B * 100 + T)
    q, k, v = (rng.standard_normal((B, T, H * dk)).astype(np.float32) * s for s in (1.5, 1.5, 1.0))
    ln = None if lengths is None else torch.tensor(lengths)
    got = masked_attention(*(torch.from_numpy(a).to(dev) for a in (q, k, v)), H, ln).cpu().numpy()
    ref = ao.attention_core(q, k, v, lengths, H)
    err = np.abs(got - ref)
    assert (err <= 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max()).all(), err.max()
    if lengths is not None:  # rows past an utterance's end: the mean of V over all T positions
        for b, n in enumerate(lengths):
            if n < T:
                want = v[b].astype(np.float64).reshape(T, H, dk).mean(axis=0).reshape(-1)
                np.testing.assert_allclose(got[b, n:], np.broadcast_to(want, (T - n, H * dk)), rtol=1e-4, atol=1e-5)


def test_attention_bf16_and_long_sequence():
    from asr_amd.attention import masked_attention

    dev = _dev()
    rng = np.random.default_rng(3)
    B, T, H, dk = 2, 3000, 6, 64
    q, k, v = (torch.from_numpy(rng.standard_normal((B, T, H * dk)).astype(np.float32)).to(dev).bfloat16()
               for _ in range(3))
    lengths = torch.tensor([3000, 1234])
    got = masked_attention(q, k, v, H, lengths).float().cpu().numpy()
    sl = slice(1200, 1300)  # the oracle is O(T^2): check a band of queries of every head
    ref = ao.attention_core(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(),
                            lengths.numpy(), H)[:, sl]
    assert np.abs(got[:, sl] - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-3


@pytest.mark.parametrize("B,T,H,dk,lengths", CASES)
def test_attention_bf16_matches_oracle(B, T, H, dk, lengths):
    """bf16 tensors (bf16 MFMA, probabilities rounded to bf16): the oracle on the rounded inputs;
    errors of a convex combination of v's stay within a few bf16 ulps of max|v|."""
    from asr_amd.attention import masked_attention

    dev = _dev()
    rng = np.random.default_rng(B * 100 + T + 1)
    q, k, v = (torch.from_numpy(rng.standard_normal((B, T, H * dk)).astype(np.float32)).bfloat16() for _ in range(3))
    ln = None if lengths is None else torch.tensor(lengths)
    got = masked_attention(q.to(dev), k.to(dev), v.to(dev), H, ln).float().cpu().numpy()
    ref = ao.attention_core(q.float().numpy(), k.float().numpy(), v.float().numpy(), lengths, H)
    assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(v.float().numpy()).max()
    assert np.sqrt(((got - ref) ** 2).mean()) <= 2 ** -8 * np.sqrt((ref ** 2).mean()) + 1e-3


def test_attention_errors():
    from asr_amd.attention import masked_attention

    dev = _dev()
    x = torch.zeros(1, 8, 48, device=dev)
    with pytest.raises(RuntimeError, match="head width"):
        masked_attention(x, x, x, 2)  # d_k = 24
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        masked_attention(x.cpu(), x.cpu(), x.cpu(), 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_packed_qkv_matches_separate_tensors(dtype):
    """q, k, v read in place from one [B, T, 3C] projection output (row stride 3C) give exactly the
    context of three dense tensors."""
    from asr_amd.attention import masked_attention, masked_attention_packed

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, T, H, dk = 3, 77, 6, 64
    qkv = torch.randn(B, T, 3 * H * dk, device=dev).to(dtype)
    lengths = torch.tensor([77, 40, 1], device=dev)
    q, k, v = (t.contiguous() for t in qkv.split(H * dk, dim=-1))
    assert torch.equal(masked_attention_packed(qkv, H, lengths), masked_attention(q, k, v, H, lengths))
    with pytest.raises(RuntimeError, match="3 \\* heads"):
        masked_attention_packed(qkv[..., :-1], H, lengths)
