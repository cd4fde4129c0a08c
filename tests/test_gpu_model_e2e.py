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


def test_full_model_at_benchmark_shape_matches_reference(golden_dir):
    """The reference's model at the benchmark's shape -- 80 -> 384 / 192, three FFT-LMU layers with
    6-head attention, 400 frames, vocabulary 34, four utterances of 400 / 400 / 317 / 146 frames
    (tests/golden/model_c2.npz, made by tests/golden/make_golden_model_c2.py from the reference's own
    LMUASRModel and decode()): log-probabilities to 1e-3, greedy token lists identical, on both
    convolution routes of the fp32 layer; and the bf16 run agrees with it on nearly every frame."""
    from asr_amd.harness import LMUASRModel

    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(golden_dir, "model_c2.npz")))
    model = LMUASRModel(input_size=80, hidden_size=384, memory_size=192, num_lmu_layers=3, theta=1000.0, seq_len=400,
                        vocab_size=34, num_attention_heads=6)
    sd = model.state_dict()  # A and fft_H are this package's own (pinned by buffers_theta1000_M192.npz)
    given = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    assert set(given) <= set(sd) and {k for k in sd if k not in given} == {
        f"encoder.lmu_layers.{i}.{b}" for i in range(3) for b in ("A", "fft_H")}
    for k, v in given.items():
        assert sd[k].shape == v.shape and sd[k].dtype == v.dtype, k
    # the module's own impulse response equals the reference's to fp32 round-off of the same recursion
    assert float((sd["encoder.lmu_layers.0.H"] - given["encoder.lmu_layers.0.H"]).abs().max()) <= 1e-6
    sd.update(given)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    spec = torch.from_numpy(g["spec"]).to(dev)
    lengths = torch.from_numpy(g["lengths"]).to(dev)
    want = [list(map(int, g["tokens"][i, : int(g["counts"][i])])) for i in range(4)]
    ref = g["log_probs"]
    valid = (np.arange(400)[None, :] < g["lengths"][:, None])
    for algo in ("auto", "toeplitz"):
        for layer in model.encoder.lmu_layers:
            layer.conv_algo = algo
        with torch.no_grad():
            lp, states = model(spec, lengths)
        assert len(states) == 3 and states[0].shape == (4, 384)
        err = np.abs(lp.cpu().numpy() - ref)[valid]
        assert err.max() <= 1e-3 * max(1.0, np.abs(ref).max()), (algo, err.max())
        assert model.transcribe(spec, lengths) == want, algo
    # bf16 autocast: the same arg-max on nearly every valid frame
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        lpb = model(spec, lengths)[0].float().cpu().numpy()
    agree = (lpb.argmax(-1) == ref.argmax(-1))[valid].mean()
    assert agree > 0.97, agree


def test_audio_to_tokens_pipeline_batch256():
    """The inference configuration at its full batch: 256 utterances of 4 s of raw audio -> mel ->
    encoder (bf16) -> greedy CTC.  Shape and structural checks (there is no reference output at this
    size): every utterance decodes on its own -- a permutation of the batch permutes the
    transcripts -- and padding does not leak: an utterance decoded alone gives the same tokens."""
    from asr_amd import MelFrontend
    from asr_amd.harness import LMUASRModel

    dev = torch.device("cuda:0")
    torch.manual_seed(10)
    model = LMUASRModel(input_size=80, hidden_size=384, memory_size=192, num_lmu_layers=3, theta=1000.0, seq_len=400,
                        vocab_size=34, num_attention_heads=6).to(dev).eval()
    with torch.no_grad():
        model.decoder.ctc_projection.weight.mul_(6.0)
    g = torch.Generator(device=dev).manual_seed(1234 + 5)
    wave = 0.1 * torch.randn(256, 63840, device=dev, generator=g)
    samples = torch.randint(16000, 63841, (256,), device=dev, generator=g)
    samples[0] = 63840
    mel, frames = MelFrontend()(wave, samples)
    assert mel.shape == (256, 400, 80)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        tokens = model.transcribe(mel, frames)
        perm = torch.randperm(256, device=dev, generator=g)
        tokens_p = model.transcribe(mel[perm], frames[perm])
        alone = model.transcribe(mel[37:38, : int(frames[37])].contiguous(), frames[37:38])
    assert len(tokens) == 256 and all(len(t) <= int(f) for t, f in zip(tokens, frames))
    assert all(0 <= k < 33 for t in tokens for k in t)
    assert [tokens[int(i)] for i in perm] == tokens_p
    assert alone[0] == tokens[37]


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


def test_strict_kernels_and_fp16_input():
    """Under bf16 autocast in inference every op of the harness runs a HIP kernel of this package:
    inside strict_kernels() nothing raises and no torch formulation is counted.  With gradients enabled under
    bf16 autocast the training kernels (attention, Linear, layer norm, LMUFFT) run and nothing raises either; in
    fp32 training the Linear layers are torch ops, and there the call raises instead of silently timing them.
    fp16 activations (the reference trainer's autocast dtype) are widened to fp32 by the layer."""
    from asr_amd import LMUFFT, harness
    from asr_amd.harness import LMUASRModel

    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    model = LMUASRModel(input_size=80, hidden_size=384, memory_size=192, num_lmu_layers=3, theta=1000.0, seq_len=400,
                        vocab_size=34, num_attention_heads=6).to(dev).eval()
    spec = torch.randn(4, 400, 80, device=dev)
    lengths = torch.tensor([400, 300, 200, 100], device=dev)
    harness.PATHS.clear()
    with harness.strict_kernels(), torch.autocast("cuda", dtype=torch.bfloat16):
        model.transcribe(spec, lengths)
    assert harness.PATHS and not [k for k in harness.PATHS if k.startswith("torch:")], harness.PATHS
    harness.PATHS.clear()
    with harness.strict_kernels(), torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        enc, _ = model.encoder(spec, lengths)
    assert enc.requires_grad and not [k for k in harness.PATHS if k.startswith("torch:")], harness.PATHS
    with pytest.raises(RuntimeError, match="strict_kernels"):
        with harness.strict_kernels(), torch.enable_grad():
            model.encoder(spec, lengths)
    layer = LMUFFT(80, 96, 64, 100, 50.0).to(dev)
    x = torch.randn(2, 100, 80, device=dev)
    h16, _ = layer(x.half())
    h32, _ = layer(x.half().float())
    assert h16.dtype == torch.float32 and torch.equal(h16, h32)


def test_training_step_of_the_whole_model_lowers_the_loss():
    """The reference's train step with the hot path swapped in (LMUFFT layers, packed CTC targets, flat AdamW with
    clipping as HIP kernels; the other ops through torch): the training forward without dropout equals the
    inference path's log-probabilities to the bf16/fp32 bar, and twenty steps lower the CTC loss."""
    from asr_amd import FlatAdamW, LearningRateScheduler
    from asr_amd.harness import LMUASRModel
    from asr_amd.parallel import GradBuckets

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    model = LMUASRModel(80, 64, 32, 2, 100.0, 60, 34, num_attention_heads=4, dropout=0.1).to(dev)
    B, T = 6, 60
    spec = torch.randn(B, T, 80, device=dev)
    in_len = torch.tensor([60, 60, 41, 60, 17, 60], device=dev)
    tgt_len = torch.tensor([9, 12, 5, 10, 3, 7], device=dev)
    targets = torch.randint(0, 33, (B, 12), device=dev)
    model.eval()
    with torch.no_grad():
        want = model(spec, in_len)[0]
    got = model.log_probs_for_training(spec, in_len)  # eval mode: dropout is the identity, torch ops with autograd
    assert got.requires_grad and torch.allclose(got, want, rtol=1e-3, atol=1e-3)

    model.train()
    enc = model.encoder
    groups = [[*model.decoder.parameters(), *enc.output_projection.parameters()]]
    for i in (1, 0):
        groups += [list(enc.layer_norms[i].parameters()), enc.lmu_layers[i], list(enc.attention_layers[i].parameters())]
    groups.append(list(enc.input_projection.parameters()))
    buckets = GradBuckets(groups, bucket_mb=4.0)
    assert sum(len(p) for p in buckets._params) == len(list(model.parameters()))
    opt = FlatAdamW(buckets=buckets, lr=2e-3, weight_decay=0.01, max_grad_norm=10.0)
    sched = LearningRateScheduler(opt, warmup_steps=5, max_lr=2e-3)
    losses = []
    for _ in range(20):
        buckets.zero_grad()
        lp = model.log_probs_for_training(spec, in_len)
        loss = model.compute_loss(lp, targets, in_len, tgt_len)
        loss.backward()
        buckets.finish()
        assert torch.isfinite(opt.grad_norm())
        opt.step()
        sched.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.7 * losses[0], losses


def test_relu_backward_folded_into_the_layer_norm_gives_the_same_gradients(monkeypatch):
    """Training under bf16 autocast: with harness.FOLD_RELU_BACKWARD the layer norm behind every LMUFFT layer hands
    back the layer's gradient already multiplied by (h > 0) and the layer skips its masking pass -- every parameter
    gradient of the model is bit-identical to the unfolded run (dropout on, the same seeds)."""
    from asr_amd import harness
    from asr_amd.harness import LMUASRModel

    dev = torch.device("cuda:0")
    torch.manual_seed(23)
    model = LMUASRModel(80, 64, 32, 2, 100.0, 60, 34, num_attention_heads=4, dropout=0.1).to(dev).train()
    B, T = 6, 60
    spec = torch.randn(B, T, 80, device=dev)
    in_len = torch.tensor([60, 60, 41, 60, 17, 60], device=dev)
    tgt_len = torch.tensor([9, 12, 5, 10, 3, 7], device=dev)
    targets = torch.randint(0, 33, (B, 12), device=dev)
    grads = {}
    for fold in (False, True):
        monkeypatch.setattr(harness, "FOLD_RELU_BACKWARD", fold)
        from asr_amd import attention, layernorm

        monkeypatch.setattr(layernorm, "_dropout_calls", 0)  # the same dropout masks in both runs
        monkeypatch.setattr(attention, "_dropout_calls", 0)
        torch.manual_seed(101)
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lp = model.log_probs_for_training(spec, in_len)
        loss = model.compute_loss(lp, targets, in_len, tgt_len)
        loss.backward()
        assert all(l.grad_is_masked == fold for l in model.encoder.lmu_layers)
        grads[fold] = (float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters()})
    assert grads[False][0] == grads[True][0]
    for n, g in grads[False][1].items():
        assert torch.equal(g, grads[True][1][n]), n
