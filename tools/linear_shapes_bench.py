"""Times the bf16 Linear kernel on the training step's shapes (rows = 64 x 400 and 96 x 400)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from asr_amd.linear import linear_bf16

dev = torch.device("cuda:0")


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n // 10 * 10) * 1e3


for rows in (25600, 38400, 102400):
    for K, N in ((384, 384), (384, 1152), (1152, 384), (384, 80), (384, 32)):
        x = torch.randn(rows, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        b = torch.randn(N, device=dev)
        us = t(lambda: linear_bf16(x, W, b))
        print(f"rows={rows} K={K} N={N}: {us:.1f} us ({2 * rows * K * N / us / 1e6:.0f} TF/s)")
