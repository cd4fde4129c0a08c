"""End-to-end golden of the reference's full model at the benchmark's model shape: 80 mel bins ->
hidden 384 / memory 192, three FFT-LMU layers with 6-head attention, 400 frames, vocabulary 34.

Run ONLY in the build container:  python tests/golden/make_golden_model_c2.py

Builds the reference's LMUASRModel (src/models/asr_model.py) in eval mode with seeded weights
rounded to bf16-representable values (so that the file compresses; they are then the reference
model's exact fp32 weights), runs it on seeded mel-like features with ragged lengths, and stores
the parameters and the H / B buffers, the input, the log-probabilities and the greedy token lists of
its own decode().  A and fft_H are left out: they follow from (memory, theta, seq_len) and are pinned
by buffers_theta1000_M192.npz.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, "/root/reference/src/models")
from models.asr_model import LMUASRModel  # noqa: E402

torch.manual_seed(4242)
cfg = types.SimpleNamespace(input_size=80, hidden_size=384, memory_size=192, num_lmu_layers=3, theta=1000.0, dropout=0.1,
                            use_fft_lmu=True, seq_len=400, use_attention=True, num_attention_heads=6, vocab_size=34,
                            use_downsampling=False)
model = LMUASRModel(cfg).eval()
with torch.no_grad():
    # a trained-like head: decisive posteriors with frequent blanks
    model.decoder.ctc_projection.weight.mul_(6.0)
    model.decoder.ctc_projection.bias[cfg.vocab_size - 1] = 1.0
    for p in model.parameters():
        p.copy_(p.bfloat16().float())
g = torch.Generator().manual_seed(4243)
B, T = 4, 400
spec = torch.randn(B, T // 4, 80, generator=g).repeat_interleave(4, dim=1) + 0.5 * torch.randn(B, T, 80, generator=g)
spec = spec.bfloat16().float()
lengths = torch.tensor([400, 400, 317, 146])
with torch.no_grad():
    lp, states = model(spec, lengths)
    tokens = model.decode(lp, lengths, beam_width=1)
out = {"spec": spec.numpy(), "lengths": lengths.numpy().astype(np.int32), "log_probs": lp.numpy().astype(np.float32)}
flat = np.full((B, T), -1, dtype=np.int32)
for i, t in enumerate(tokens):
    flat[i, : len(t)] = t
out["tokens"] = flat
out["counts"] = np.array([len(t) for t in tokens], dtype=np.int32)
for k, v in model.state_dict().items():
    if k.endswith(".fft_H") or k.endswith(".A"):
        continue
    out["sd." + k] = v.numpy()
np.savez_compressed(os.path.join(HERE, "model_c2.npz"), **out)
print("tokens per utterance", out["counts"], "params", sum(p.numel() for p in model.parameters()))
