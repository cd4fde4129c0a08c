"""End-to-end golden of the reference's full model (encoder with FFT-LMU layers + CTC head).

Run ONLY in the build container:  python tests/golden/make_golden_model.py

Builds the reference's LMUASRModel (src/models/asr_model.py) in eval mode, runs it on seeded
mel-like features with ragged lengths, and stores the state_dict (as plain arrays), the input,
the log-probabilities and the greedy token lists of its own decode().
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

torch.manual_seed(2024)
cfg = types.SimpleNamespace(input_size=80, hidden_size=128, memory_size=64, num_lmu_layers=2, theta=100.0, dropout=0.1,
                            use_fft_lmu=True, seq_len=200, use_attention=True, num_attention_heads=4, vocab_size=34,
                            use_downsampling=False)
model = LMUASRModel(cfg).eval()
with torch.no_grad():
    # a trained-like head: decisive posteriors with frequent blanks
    model.decoder.ctc_projection.weight.mul_(6.0)
    model.decoder.ctc_projection.bias[cfg.vocab_size - 1] = 1.0
g = torch.Generator().manual_seed(2025)
B, T = 4, 200
spec = torch.randn(B, T // 4, 80, generator=g).repeat_interleave(4, dim=1) + 0.5 * torch.randn(B, T, 80, generator=g)
lengths = torch.tensor([200, 200, 160, 73])
with torch.no_grad():
    lp, states = model(spec, lengths)
    tokens = model.decode(lp, lengths, beam_width=1)
out = {"spec": spec.numpy(), "lengths": lengths.numpy().astype(np.int32), "log_probs": lp.numpy()}
flat = np.full((B, T), -1, dtype=np.int32)
for i, t in enumerate(tokens):
    flat[i, : len(t)] = t
out["tokens"] = flat
out["counts"] = np.array([len(t) for t in tokens], dtype=np.int32)
for k, v in model.state_dict().items():
    a = v.numpy()
    if np.iscomplexobj(a):
        out["sd_re." + k], out["sd_im." + k] = a.real.copy(), a.imag.copy()
    else:
        out["sd." + k] = a
np.savez_compressed(os.path.join(HERE, "model_e2e.npz"), **out)
print("tokens per utterance", out["counts"], "params", sum(p.numel() for p in model.parameters()))
