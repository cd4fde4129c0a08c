"""End to end on the GPU: the reference's full model (encoder with two FFT-LMU layers, attention,
layer norms, CTC head), evaluated by the harness of this package from the reference's own
state_dict, against log-probabilities and greedy token lists that the reference produced
(tests/golden/model_e2e.npz, made by tests/golden/make_golden_model.py).  LMUFFT layers and the
CTC head are this package's HIP kernels; the rest of the harness is torch ops.
fp32: log-probabilities to 1e-3 (two layers of attention + layer norm amplify fp32 rounding),
greedy token lists identical."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "model_e2e.npz")))
    sd = {}
    for k, v in g.items():
        if k.startswith("sd."):
            sd[k[3:]] = torch.from_numpy(v)
        elif k.startswith("sd_re."):
            sd[k[6:]] = torch.complex(torch.from_numpy(v), torch.from_numpy(g["sd_im." + k[6:]]))
    return g, sd


def test_full_model_matches_reference(golden_dir):
    from asr_amd.harness import LMUASRModel

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    g, sd = _load(golden_dir)
    model = LMUASRModel(input_size=80, hidden_size=128, memory_size=64, num_lmu_layers=2, theta=100.0, seq_len=200,
                        vocab_size=34, num_attention_heads=4)
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    model = model.to(dev).eval()
    spec = torch.from_numpy(g["spec"]).to(dev)
    lengths = torch.from_numpy(g["lengths"]).to(dev)
    lp, states = model(spec, lengths)
    assert len(states) == 2 and states[0].shape == (4, 128)
    ref = g["log_probs"]
    err = np.abs(lp.cpu().numpy() - ref)
    assert err.max() <= 1e-3 * max(1.0, np.abs(ref).max()), err.max()
    want = [list(map(int, g["tokens"][i, : int(g["counts"][i])])) for i in range(4)]
    assert model.transcribe(spec, lengths) == want
    # both convolution routes of the fp32 layer give the same transcript
    for layer in model.encoder.lmu_layers:
        layer.conv_algo = "toeplitz"
    assert model.transcribe(spec, lengths) == want


def test_audio_to_tokens_pipeline():
    """Raw 16 kHz audio in HBM -> mel frontend -> encoder -> greedy CTC, nothing touching the host
    until the token lists (the inference benchmark pipeline at a small batch); bf16 and fp32 runs
    of the same model agree on most tokens."""
    from asr_amd import MelFrontend
    from asr_amd.harness import LMUASRModel

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    model = LMUASRModel(input_size=80, hidden_size=384, memory_size=192, num_lmu_layers=3, theta=1000.0, seq_len=400,
                        vocab_size=34, num_attention_heads=6).to(dev).eval()
    wave = 0.1 * torch.randn(8, 63840, device=dev)
    samples = torch.tensor([63840, 63840, 60000, 48000, 32000, 16000, 8000, 4000], device=dev)
    mel, frames = MelFrontend()(wave, samples)
    assert mel.shape == (8, 400, 80) and int(frames[-1]) == 26
    tokens = model.transcribe(mel, frames)
    assert len(tokens) == 8 and all(len(t) <= int(f) for t, f in zip(tokens, frames))
    assert all(0 <= k < 33 for t in tokens for k in t)
    # bf16 autocast (the benchmark's dtype) against the fp32 run: log-probabilities within the bf16
    # bar of their magnitude, with fp32 layer-norm outputs and with the bf16 activation stream
    with torch.no_grad():
        lp32 = model(mel, frames)[0].float()
        valid = (torch.arange(400, device=dev)[None, :] < frames[:, None])
        scale = float(lp32.abs().max())
        for stream in (False, True):
            model.encoder.bf16_stream = stream
            with torch.autocast("cuda", dtype=torch.bfloat16):
                lpb = model(mel, frames)[0].float()
            err = float((lpb - lp32)[valid].abs().max())
            assert err <= 2e-2 * max(1.0, scale) * 4, (stream, err, scale)
            agree = float((lpb.argmax(-1) == lp32.argmax(-1))[valid].float().mean())
            assert agree > 0.9, (stream, agree)


def test_cached_bf16_weights_follow_parameter_updates():
    """The harness keeps bf16 copies of its projection weights for inference under autocast; an
    in-place update of a parameter (an optimiser step, a checkpoint load) must invalidate them."""
    from asr_amd.harness import LMUEncoder

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    enc = LMUEncoder(80, 128, 64, 1, 100.0, seq_len=64, num_attention_heads=4).to(dev).eval()
    x = torch.randn(2, 64, 80, device=dev)
    lengths = torch.tensor([64, 30], device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y0 = enc(x, lengths)[0].float()
        assert float(y0.abs().max()) > 0
        enc.output_projection.weight.zero_()
        enc.output_projection.bias.fill_(0.25)
        y1 = enc(x, lengths)[0].float()
        assert torch.allclose(y1, torch.full_like(y1, 0.25))
        for p in enc.attention_layers[0].attention.w_v.parameters():
            p.zero_()  # values all zero -> the attention context is zero -> the block output is LN(w_o.bias + x)
        y2 = enc(x, lengths)[0].float()
        assert torch.allclose(y2, torch.full_like(y2, 0.25))
        enc.output_projection.weight.copy_(torch.eye(128, device=dev))
        enc.output_projection.bias.zero_()
        a = enc(x, lengths)[0].float()
        enc.attention_layers[0].attention.w_q.weight.mul_(3.0)  # context is zero whatever the scores are
        b = enc(x, lengths)[0].float()
        assert torch.equal(a, b)
