"""Training golden of the reference's full model at the benchmark's shape: CTC loss and EVERY parameter
gradient of one train-mode step without dropout.

Run ONLY in the build container:  python tests/golden/make_golden_model_c2_train.py

Builds the reference's LMUASRModel (src/models/asr_model.py) with dropout 0.0 in train() mode, loads the
weights and the input of model_c2.npz (made by make_golden_model_c2.py: the same model in eval mode), runs
forward -> compute_loss (nn.CTCLoss(blank = vocab - 1, mean, zero_infinity), src/models/asr_model.py:105,
130-156) -> backward on four ragged utterances with seeded label sequences, and stores the loss and, for every
parameter, its gradient's L2 norm, its sum and every eighth element (the whole gradient is 10.5 MB; the sample
pins every tensor at 1/8 of that).  Read-only hooks on the reference's modules also record, per LMU layer, the ReLU
patterns (hidden units h > 0, encoder u > 0; bit-packed) and the smallest |pre-activation| magnitudes -- an
implementation whose fp32 round-off puts one of the ~1.8 M pre-activations on the other side of zero
legitimately differs in every gradient below that unit -- and the root-sum-square of the terms of the two
cancelling sums gw_u = sum ga x and gb_u = sum ga (ga = gradient at the encoder's pre-activation).
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

STRIDE = 8

g = dict(np.load(os.path.join(HERE, "model_c2.npz")))
torch.manual_seed(4242)
cfg = types.SimpleNamespace(input_size=80, hidden_size=384, memory_size=192, num_lmu_layers=3, theta=1000.0, dropout=0.0,
                            use_fft_lmu=True, seq_len=400, use_attention=True, num_attention_heads=6, vocab_size=34,
                            use_downsampling=False)
model = LMUASRModel(cfg)
sd = model.state_dict()
for k, v in g.items():
    if k.startswith("sd."):
        sd[k[3:]] = torch.from_numpy(v)
model.load_state_dict(sd)
model.train()
spec = torch.from_numpy(g["spec"])
lengths = torch.from_numpy(g["lengths"]).long()
gen = torch.Generator().manual_seed(4244)
tgt_len = torch.tensor([37, 52, 29, 11])
targets = torch.randint(0, 33, (4, 52), generator=gen)
rec = {}
for i, lay in enumerate(model.encoder.lmu_layers):
    lay.register_forward_hook(lambda mod, inp, outp, i=i: rec.__setitem__(f"h{i}", outp[0].detach()))
    lay.f_u.register_forward_hook(lambda mod, inp, outp, i=i: rec.__setitem__(f"u{i}", outp.detach()))
    lay.W_u.register_forward_hook(lambda mod, inp, outp, i=i: rec.__setitem__(f"x{i}", inp[0].detach()))
    lay.f_u.register_full_backward_hook(lambda mod, gin, gout, i=i: rec.__setitem__(f"ga{i}", gin[0].detach()))
log_probs, _ = model(spec, lengths)
loss = model.compute_loss(log_probs, targets, lengths, tgt_len)
loss.backward()
out = {"targets": targets.numpy().astype(np.int64), "target_lengths": tgt_len.numpy().astype(np.int64),
       "loss": np.array([float(loss)], dtype=np.float64), "stride": np.array([STRIDE], dtype=np.int32)}
for i in range(3):
    out[f"hmask{i}"] = np.packbits((rec[f"h{i}"] > 0).numpy().reshape(-1))
    out[f"umask{i}"] = np.packbits((rec[f"u{i}"] > 0).numpy().reshape(-1))
    ga, x = rec[f"ga{i}"].double().reshape(4, 400, 1), rec[f"x{i}"].double()
    out[f"termnorm.encoder.lmu_layers.{i}.W_u.weight"] = torch.sqrt(((ga * x) ** 2).sum(dim=(0, 1))).numpy()[None, :]
    out[f"termnorm.encoder.lmu_layers.{i}.W_u.bias"] = np.array([float(torch.sqrt((ga ** 2).sum()))])
total = 0.0
for name, p in model.named_parameters():
    gr = p.grad.detach().double().reshape(-1)
    out["norm." + name] = np.array([float(gr.norm())])
    out["sum." + name] = np.array([float(gr.sum())])
    out["sample." + name] = p.grad.detach().reshape(-1)[::STRIDE].numpy().astype(np.float32)
    total += float(gr.norm()) ** 2
out["total_norm"] = np.array([total ** 0.5])
np.savez_compressed(os.path.join(HERE, "model_c2_train.npz"), **out)
print("loss", float(loss), "gradient norm", total ** 0.5, "tensors", len(list(model.parameters())))
