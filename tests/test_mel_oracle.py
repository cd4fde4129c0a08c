"""CPU checks of the log-mel oracle (oracle/mel_oracle.py).  The reference computes this step
with torchaudio, which is not installed, and holds no fixtures for it: parity is UNPINNED.
What can be checked here: the restatement against an independent STFT (torch.stft), the
filterbank's documented properties, and the normalisation against torch's mean/std."""
import numpy as np
import torch

from oracle import mel_oracle as mo


def test_power_spectrogram_matches_torch_stft():
    rng = np.random.default_rng(0)
    for n in (201, 1000, 16000, 16123):
        w = rng.standard_normal(n) * 0.1
        got = mo.power_spectrogram(w)
        st = torch.stft(torch.from_numpy(w), n_fft=400, hop_length=160, win_length=400,
                        window=torch.hann_window(400, periodic=True, dtype=torch.float64), center=True,
                        pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        ref = (st.abs() ** 2).T.numpy()
        assert got.shape == ref.shape == (1 + n // 160, 201)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * ref.max())


def test_filterbank_properties():
    fb = mo.mel_filterbank()
    assert fb.shape == (201, 80)
    assert (fb >= 0).all() and fb.max() <= 1.0
    assert (fb[0] == 0).all() and fb[200].max() < 1e-12  # 0 Hz and 8000 Hz are filter edges
    # triangles: every filter is unimodal, and neighbours overlap so that interior bins are covered
    for m in range(80):
        nz = np.nonzero(fb[:, m])[0]
        assert nz.size >= 1 and (np.diff(nz) == 1).all()
    # centre frequencies are equally spaced on the HTK mel scale
    peak_hz = np.array([(fb[:, m] * np.arange(201) * 40.0).sum() / fb[:, m].sum() for m in range(40, 80)])
    mel = 2595.0 * np.log10(1.0 + peak_hz / 700.0)
    assert np.abs(np.diff(mel) - np.diff(mel).mean()).max() < 0.1 * np.diff(mel).mean()
    # a known value: filter 79 is centred on mel point 80 of 81 = 7737 Hz
    assert fb[:, 79].argmax() in (193, 194)


def test_normalisation_matches_torch():
    rng = np.random.default_rng(1)
    f = rng.standard_normal((57, 80)) * 7 - 30
    f[:, 5] = -100.0  # a constant bin: std 0 -> 1
    got = mo.normalize_features(f)
    t = torch.from_numpy(f)
    mean, std = t.mean(0, keepdim=True), t.std(0, keepdim=True)
    std = torch.where(std == 0, torch.ones_like(std), std)
    np.testing.assert_allclose(got, ((t - mean) / std).numpy(), rtol=1e-12, atol=1e-12)
    assert (got[:, 5] == 0).all()


def test_log_mel_known_answers():
    # silence: every bin clamps to 1e-10 -> -100 dB -> constant -> normalised to 0
    z = np.zeros(4000)
    assert (mo.log_mel(z) == -100.0).all()
    assert (mo.extract_mel_features(z) == 0).all()
    # a 1 kHz tone peaks in the filter whose triangle covers 1000 Hz (bin 25)
    t = np.arange(16000) / 16000.0
    lm = mo.log_mel(np.sin(2 * np.pi * 1000.0 * t))
    fb = mo.mel_filterbank()
    assert lm[50].argmax() == fb[25].argmax()
    assert mo.num_frames(63840) == 400 and mo.num_frames(479840) == 3000


def test_batch_features_pads_with_zeros():
    rng = np.random.default_rng(2)
    w = rng.standard_normal((2, 3200)) * 0.1
    out, frames = mo.batch_features(w, [3200, 1700])
    assert out.shape == (2, 21, 80) and list(frames) == [21, 11]
    assert (out[1, 11:] == 0).all() and np.abs(out[1, :11]).max() > 0
    np.testing.assert_allclose(out[1, :11], mo.extract_mel_features(w[1, :1700]))
