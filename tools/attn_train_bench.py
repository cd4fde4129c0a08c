"""Times the training attention (forward with dropout + backward) at the training and inference batch shapes.
Argument: the dropout probability (default 0.1; 0 skips the dropout hash)."""
import sys

import torch
from asr_amd import _lib
from asr_amd.attention import masked_attention_train

P = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
dev = torch.device("cuda:0")
for (B, T, H, dk, dt) in [(64, 400, 6, 64, torch.bfloat16), (64, 400, 6, 64, torch.float32), (32, 3000, 6, 64, torch.bfloat16)]:
    q, k, v = (torch.randn(B, T, H * dk, device=dev).to(dt).requires_grad_(True) for _ in range(3))
    go = torch.randn(B, T, H * dk, device=dev).to(dt)
    ln = torch.full((B,), T, device=dev)
    for _ in range(3):
        masked_attention_train(q, k, v, H, ln, P).backward(go)
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_reset()
    for _ in range(10):
        masked_attention_train(q, k, v, H, ln, P).backward(go)
    torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    us = {kk: round(vv[0] / vv[1] * 1e3, 1) for kk, vv in prof.items() if vv[1]}
    flops = 4.0 * B * H * T * T * dk
    print(B, T, dt, "p", P, us, f"fwd {flops / us['k_attention_fwd_train'] / 1e6:.0f} TF/s, bwd {2.5 * flops / us['k_attention_bwd'] / 1e6:.0f} TF/s (2.5x the forward's flops)", flush=True)
