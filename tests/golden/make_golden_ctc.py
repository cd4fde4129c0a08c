"""Golden vectors of the CTC head + greedy decode, from the reference's own classes.

Run ONLY in the build container (the reference checkout at /root/reference):

    python tests/golden/make_golden_ctc.py

Imports src/models/asr_model.py (torch only), builds its CTCDecoder with seeded weights, feeds
seeded "encoder outputs" and stores inputs, weights, log-probs and the greedy token lists that
LMUASRModel.decode produces.  Only numbers are stored.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, "/root/reference/src/models")
from models.asr_model import CTCDecoder, LMUASRModel  # noqa: E402

torch.manual_seed(77)
B, T, Hd, V = 4, 400, 384, 34
dec = CTCDecoder(Hd, V)
torch.nn.init.xavier_uniform_(dec.ctc_projection.weight, gain=4.0)  # peaky posteriors: long runs and blanks
torch.nn.init.uniform_(dec.ctc_projection.bias, -0.5, 0.5)
with torch.no_grad():
    dec.ctc_projection.bias[V - 1] += 1.5  # blanks must be frequent, as in a trained CTC model
g = torch.Generator().manual_seed(78)
# slowly varying encoder output so that consecutive frames often agree (repeats to collapse)
base = torch.randn(B, T // 8 + 1, Hd, generator=g).repeat_interleave(8, dim=1)[:, :T]
enc = base + 0.3 * torch.randn(B, T, Hd, generator=g)
lengths = torch.tensor([400, 399, 250, 1])
with torch.no_grad():
    lp = dec(enc)
# LMUASRModel.decode / _greedy_decode only touch self.config.vocab_size: call them on a bare object
shell = types.SimpleNamespace(config=types.SimpleNamespace(vocab_size=V))
shell._greedy_decode = types.MethodType(LMUASRModel._greedy_decode, shell)
tokens = LMUASRModel.decode(shell, lp, lengths, beam_width=1)
counts = np.array([len(t) for t in tokens], dtype=np.int32)
flat = np.full((B, T), -1, dtype=np.int32)
for i, t in enumerate(tokens):
    flat[i, : len(t)] = t
np.savez_compressed(os.path.join(HERE, "ctc_greedy.npz"), enc=enc.numpy(), W=dec.ctc_projection.weight.detach().numpy(),
                    b=dec.ctc_projection.bias.detach().numpy(), lengths=lengths.numpy().astype(np.int32),
                    log_probs=lp.numpy(), tokens=flat, counts=counts)
print("token counts", counts, "blank share", float((lp.argmax(-1) == V - 1).float().mean()))
