"""GPU parity tests of the log-mel frontend (k_mel_power, k_mel_norm) through the C ABI,
against oracle/mel_oracle.py.  Parity with the reference itself is UNPINNED for this step
(torchaudio is absent and the reference holds no fixtures; see the oracle's header).

Bar: fp32.  Un-normalised features are in dB: |got - ref| <= 1e-4 * |ref| + 2e-3 dB (a relative
error of 1e-4 in a mel-bin power is 4.3e-4 dB; bins far below the frame's peak carry the
DFT's absolute rounding error).  Normalised features are O(1): atol 2e-4, rtol 1e-4."""
import numpy as np
import pytest
import torch

from oracle import mel_oracle as mo

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    return torch.device("cuda:0")


def _check(got, ref, atol, rtol=1e-4, name=""):
    got = np.asarray(got, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    bad = err > atol + rtol * np.abs(ref)
    assert not bad.any(), f"{name}: {bad.sum()} of {bad.size} out of tolerance, max err {err.max():.3e} at " \
                          f"{np.unravel_index(err.argmax(), err.shape)}"


def _noise(rng, B, L):
    # speech-like dynamics: white noise with a slow amplitude envelope
    env = 0.02 + 0.1 * np.abs(np.sin(np.linspace(0, 9, L)))[None, :]
    return (rng.standard_normal((B, L)) * env).astype(np.float32)


@pytest.mark.parametrize("B,L", [(1, 201), (3, 16000), (2, 16123), (2, 63840), (1, 20800 + 37)])
def test_mel_matches_oracle(B, L):
    from asr_amd import MelFrontend

    dev = _dev()
    rng = np.random.default_rng(L)
    w = _noise(rng, B, L)
    for normalize in (False, True):
        fe = MelFrontend(normalize=normalize)
        mel, frames = fe(torch.from_numpy(w).to(dev))
        ref, fr = mo.batch_features(w.astype(np.float64), None, normalize)
        assert list(frames.cpu().numpy()) == list(fr)
        _check(mel.cpu().numpy(), ref, atol=2e-4 if normalize else 2e-3, name=f"normalize={normalize}")


def test_mel_ragged_lengths_and_padding():
    from asr_amd import MelFrontend

    dev = _dev()
    rng = np.random.default_rng(7)
    B, L = 5, 48000
    w = _noise(rng, B, L)
    lengths = np.array([48000, 47999, 24000, 16160, 777], dtype=np.int32)
    mel, frames = MelFrontend()(torch.from_numpy(w).to(dev), torch.from_numpy(lengths).to(dev))
    ref, fr = mo.batch_features(w.astype(np.float64), lengths, True)
    assert list(frames.cpu().numpy()) == list(fr)
    got = mel.cpu().numpy()
    _check(got, ref, atol=2e-4, name="ragged")
    for b in range(B):
        assert (got[b, fr[b]:] == 0).all()


def test_mel_tone_silence_and_long_utterance():
    from asr_amd import MelFrontend

    dev = _dev()
    fe_raw, fe = MelFrontend(normalize=False), MelFrontend()
    # silence: -100 dB everywhere, normalised to exactly 0
    z = torch.zeros(2, 8000, device=dev)
    assert float((fe_raw(z)[0] + 100.0).abs().max()) < 1e-4  # log10f of the clamp value, to an ulp
    assert (fe(z)[0] == 0).all()
    # a pure tone: leakage bins sit ~60-100 dB under the peak; looser absolute floor there
    t = np.arange(32000) / 16000.0
    tone = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)[None, :]
    got = fe_raw(torch.from_numpy(tone).to(dev))[0].cpu().numpy()
    ref = mo.batch_features(tone.astype(np.float64), None, False)[0]
    peak = ref.max()
    near = ref > peak - 60.0
    _check(got[near], ref[near], atol=2e-3, name="tone, bins within 60 dB of the peak")
    assert np.abs(got - ref)[~near].max() < 1.0
    assert got[0, 100].argmax() == ref[0, 100].argmax()
    # 30 s utterance (3000 frames), the long-utterance benchmark shape
    rng = np.random.default_rng(3)
    w = _noise(rng, 1, 479840)
    mel, frames = fe(torch.from_numpy(w).to(dev))
    assert mel.shape == (1, 3000, 80) and int(frames[0]) == 3000
    _check(mel.cpu().numpy(), mo.batch_features(w.astype(np.float64), None, True)[0], atol=2e-4, name="30 s")


def test_mel_full_batch_properties():
    """Batch 256 x 4 s (the inference benchmark shape): a few utterances against the oracle,
    zero mean / unit unbiased std per utterance and bin for all."""
    from asr_amd import MelFrontend

    dev = _dev()
    B, L = 256, 63840
    g = torch.Generator(device="cpu").manual_seed(1239)
    w = torch.randn(B, L, generator=g) * 0.1
    mel, frames = MelFrontend()(w.to(dev))
    assert mel.shape == (B, 400, 80) and (frames == 400).all()
    for b in (0, 131, 255):
        _check(mel[b].cpu().numpy(), mo.extract_mel_features(w[b].numpy().astype(np.float64)), atol=2e-4,
               name=f"utterance {b}")
    m = mel.double()
    assert float(m.mean(dim=1).abs().max()) < 1e-4
    assert float((m.std(dim=1) - 1).abs().max()) < 1e-4


def test_mel_errors():
    from asr_amd import MelFrontend

    dev = _dev()
    fe = MelFrontend()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fe(torch.zeros(1, 1000))
    with pytest.raises(RuntimeError, match="reflect"):
        fe(torch.zeros(1, 200, device=dev))
    with pytest.raises(ValueError):
        MelFrontend(n_mels=40)
    # a ragged batch with an utterance too short for the reflect padding: lengths given from the host are
    # refused (torchaudio raises for such an utterance too); on the GPU they are clamped, frames with them
    wave = torch.zeros(3, 4000, device=dev)
    with pytest.raises(RuntimeError, match="reflect"):
        fe(wave, torch.tensor([4000, 150, 3000]))
    with pytest.raises(RuntimeError, match="reflect"):
        fe(wave, torch.tensor([4000, 4001, 3000]))
    _, frames = fe(wave, torch.tensor([4000, 0, 3000], device=dev))
    assert frames.tolist() == [26, 1, 19]
