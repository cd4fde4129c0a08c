"""CPU restatement (numpy fp64) of the reference's log-mel frontend.  TEST INFRASTRUCTURE ONLY:
used by tests/, by __graft_entry__.smoke() and by nothing in asr_amd/.

PARITY UNPINNED.  The reference computes these features with torchaudio
(src/data/preprocessing.py:41-51, 76-117: MelSpectrogram(sample_rate=16000, n_fft=400,
hop_length=160, win_length=400, n_mels=80, power=2.0) and AmplitudeToDB()), pinned at
torchaudio 2.1.0 in pytorch_lmu_environment.yml.  torchaudio is not installed here and the
reference holds no fixtures or tests for this step, so this restatement follows torchaudio's
documented defaults and is only self-pinned (against torch.stft, which IS available, in
tests/test_mel_oracle.py):

  Spectrogram: centre=True with reflect padding of n_fft/2, periodic Hann window, one-sided
      400-point DFT (201 bins), power 2, no normalisation;
  MelScale: f_min 0, f_max 8000, HTK scale mel(f) = 2595 log10(1 + f/700), n_mels + 2 points
      equally spaced in mel, triangular filters max(0, min(down, up)) over the bin centres
      linspace(0, 8000, 201), norm=None;
  AmplitudeToDB: stype "power", top_db None: 10 log10(clamp(x, min=1e-10));
  then [T, 80] and (x - mean_t) / std_t with the unbiased std, std == 0 -> 1
      (src/data/preprocessing.py:93, 111-117).
"""
import numpy as np

SAMPLE_RATE, N_FFT, HOP, N_MELS, N_BINS = 16000, 400, 160, 80, 201


def mel_filterbank():
    """[201, 80] triangular HTK filters (torchaudio.functional.melscale_fbanks defaults)."""
    freqs = np.linspace(0.0, SAMPLE_RATE // 2, N_BINS)
    m_min = 2595.0 * np.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * np.log10(1.0 + (SAMPLE_RATE / 2) / 700.0)
    m_pts = np.linspace(m_min, m_max, N_MELS + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def num_frames(n_samples):
    return 1 + n_samples // HOP


def power_spectrogram(wave):
    """[T, 201] = |STFT|^2 of a 1-D waveform (src/data/preprocessing.py:87, Spectrogram part)."""
    wave = np.asarray(wave, dtype=np.float64)
    assert wave.ndim == 1 and wave.size > N_FFT // 2, "reflect padding needs more than n_fft/2 samples"
    padded = np.pad(wave, N_FFT // 2, mode="reflect")
    T = num_frames(wave.size)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N_FFT) / N_FFT)  # periodic Hann
    idx = np.arange(T)[:, None] * HOP + np.arange(N_FFT)[None, :]
    spec = np.fft.rfft(padded[idx] * window[None, :], n=N_FFT, axis=1)
    return spec.real ** 2 + spec.imag ** 2


def log_mel(wave):
    """[T, 80] log-mel in dB before normalisation (src/data/preprocessing.py:87-93)."""
    mel = power_spectrogram(wave) @ mel_filterbank()
    return 10.0 * np.log10(np.maximum(mel, 1e-10))


def normalize_features(feat):
    """src/data/preprocessing.py:101-117."""
    mean = feat.mean(axis=0, keepdims=True)
    std = feat.std(axis=0, ddof=1, keepdims=True)
    std = np.where(std == 0, 1.0, std)
    return (feat - mean) / std


def extract_mel_features(wave, normalize=True):
    """src/data/preprocessing.py:76-99."""
    f = log_mel(wave)
    return normalize_features(f) if normalize else f


def batch_features(waves, lengths, normalize=True):
    """Zero-padded batch [B, Tmax, 80] as the reference's collate_fn builds it
    (src/data/preprocessing.py:347-353) and the per-utterance frame counts."""
    waves = np.asarray(waves)
    B, L = waves.shape
    Tmax = num_frames(L)
    out = np.zeros((B, Tmax, N_MELS))
    frames = []
    for b in range(B):
        n = int(lengths[b]) if lengths is not None else L
        f = extract_mel_features(waves[b, :n], normalize)
        out[b, : f.shape[0]] = f
        frames.append(f.shape[0])
    return out, np.asarray(frames)
