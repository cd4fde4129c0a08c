"""A small checkpoint in the reference's on-disk format, written by the reference's own model.

Run ONLY in the build container:  python tests/golden/make_golden_ckpt.py

Builds the reference's LMUASRModel (src/models/asr_model.py, FFT LMU layers, tiny sizes), takes
its state_dict and stores it the way src/training/utils.py:328-340 does:
{'epoch', 'model_state_dict', 'optimizer_state_dict', 'loss', 'metrics'} through torch.save.
Only tensors and plain Python values are stored.
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, "/root/reference/src/models")
from models.asr_model import LMUASRModel  # noqa: E402

torch.manual_seed(11)
cfg = types.SimpleNamespace(input_size=16, hidden_size=24, memory_size=8, num_lmu_layers=2, theta=20.0, dropout=0.0,
                            use_fft_lmu=True, seq_len=32, use_attention=True, num_attention_heads=2, vocab_size=7,
                            use_downsampling=False)
model = LMUASRModel(cfg)
with torch.no_grad():  # biases away from zero so that a swapped tensor would be noticed
    for p in model.parameters():
        if p.dim() == 1:
            p.uniform_(-0.3, 0.3)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
ckpt = {"epoch": 3, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict(), "loss": 1.25,
        "metrics": {"wer": 0.5}}
torch.save(ckpt, os.path.join(HERE, "checkpoint_epoch_3.pt"))
print(sorted(k for k in ckpt["model_state_dict"] if "lmu_layers.0" in k or "decoder" in k))
