"""Attention kernels (6 heads x 64) at several batch / length combinations."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from asr_amd.attention import masked_attention

dev = torch.device("cuda:0")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for dt in (torch.bfloat16, torch.float32):
    for B, T in ((256, 400), (64, 400), (32, 3000), (128, 1000), (256, 250), (256, 100)):
        q, k, v = (torch.randn(B, T, 384, device=dev).to(dt) for _ in range(3))
        lengths = torch.full((B,), T, device=dev, dtype=torch.int32)
        us = t(lambda: masked_attention(q, k, v, 6, lengths))
        print(f"{dt} B={B} T={T}: {us:.1f} us ({4 * B * 6 * T * T * 64 / us / 1e6:.0f} TF/s)")
