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
