"""A checkpoint written in the reference's format by the reference's own model
(tests/golden/checkpoint_epoch_3.pt, made by tests/golden/make_golden_ckpt.py) loads into the
modules of this package."""
import os

import torch

from asr_amd.checkpoint import ctc_decoder_from_checkpoint, lmu_layers_from_checkpoint, read_checkpoint


def test_reference_checkpoint_loads(golden_dir):
    ck = read_checkpoint(os.path.join(golden_dir, "checkpoint_epoch_3.pt"))
    assert ck["epoch"] == 3 and abs(ck["loss"] - 1.25) < 1e-12 and ck["metrics"] == {"wer": 0.5}
    sd = ck["model_state_dict"]
    layers = lmu_layers_from_checkpoint(ck, theta=20.0)
    assert len(layers) == 2
    for i, layer in enumerate(layers):
        assert (layer.hidden_size, layer.memory_size, layer.seq_len) == (24, 8, 32)
        assert layer.W_h.weight.shape == (24, 8 + 24)
        for k, v in layer.state_dict().items():
            ref = sd[f"encoder.lmu_layers.{i}.{k}"]
            assert v.dtype == ref.dtype and torch.equal(v, ref), k
        # the buffers the constructor builds equal the reference's (same SciPy / fp32 recipe)
        fresh = type(layer)(24, 24, 8, 32, 20.0)
        torch.testing.assert_close(fresh.H, layer.H, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(fresh.A, layer.A, rtol=1e-6, atol=1e-8)
    dec = ctc_decoder_from_checkpoint(ck)
    assert (dec.input_size, dec.vocab_size) == (24, 7)
    assert torch.equal(dec.ctc_projection.weight, sd["decoder.ctc_projection.weight"])
    assert torch.equal(dec.ctc_projection.bias, sd["decoder.ctc_projection.bias"])


def test_wrapped_model_checkpoint_and_theta(golden_dir, tmp_path):
    """The reference's single-GPU trainer saves the model as it holds it (src/training/trainer.py:294);
    wrapped in DataParallel / DDP every key carries a "module." prefix.  Such a file loads like the
    plain one, and a theta that is not passed leaves the constructed attribute alone."""
    ck = torch.load(os.path.join(golden_dir, "checkpoint_epoch_3.pt"), map_location="cpu", weights_only=True)
    wrapped = dict(ck)
    wrapped["model_state_dict"] = {"module." + k: v for k, v in ck["model_state_dict"].items()}
    path = tmp_path / "wrapped.pt"
    torch.save(wrapped, path)
    ck2 = read_checkpoint(str(path))
    assert set(ck2["model_state_dict"]) == set(ck["model_state_dict"])
    layers = lmu_layers_from_checkpoint(ck2)
    assert len(layers) == 2 and layers[0].theta is not None
    assert torch.equal(layers[1].W_h.weight, ck["model_state_dict"]["encoder.lmu_layers.1.W_h.weight"])
    assert ctc_decoder_from_checkpoint(ck2).vocab_size == 7


def test_fft_spectra_cache_is_bounded():
    """forward() accepts any T <= seq_len; the per-length spectra cache keeps the most recent few."""
    from asr_amd import LMUFFT

    layer = LMUFFT(8, 8, 4, 64, 10.0)
    calls = []
    import asr_amd.lmu as lmu_mod

    orig = lmu_mod.prepare_impulse_fft
    lmu_mod.prepare_impulse_fft = lambda H, T: calls.append(T) or ("spectra", T)
    try:
        for T in range(20, 20 + 3 * layer._FFTPREP_MAX):
            layer._impulse_fft(T)
        assert len(layer._fftprep) == layer._FFTPREP_MAX
        n = len(calls)
        layer._impulse_fft(20 + 3 * layer._FFTPREP_MAX - 1)  # most recent: a hit
        assert len(calls) == n
        layer.H.add_(0.0)  # an in-place update of H invalidates every entry
        layer._impulse_fft(30)
        assert list(layer._fftprep) == [30]
    finally:
        lmu_mod.prepare_impulse_fft = orig
