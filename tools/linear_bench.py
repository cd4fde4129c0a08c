"""Times the bf16 Linear kernel against torch's linear (hipBLASLt) on the inference shapes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from asr_amd.linear import linear_bf16, linear_add_layer_norm_bf16
from asr_amd.layernorm import add_layer_norm

dev = torch.device("cuda:0")
rows = 256 * 400


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K, N in ((80, 384), (384, 384), (384, 1152)):
    x = torch.randn(rows, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev)
    b16 = b.bfloat16()
    us = t(lambda: linear_bf16(x, W, b))
    ref = t(lambda: torch.nn.functional.linear(x, W, b16))
    print(f"K={K} N={N}: kernel {us:.1f} us ({2 * rows * K * N / us / 1e6:.0f} TF/s), torch {ref:.1f} us")
    if N <= 384:
        res = torch.randn(rows, N, device=dev)
        g, be = torch.randn(N, device=dev), torch.randn(N, device=dev)
        for rd, od in ((torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)):
            r = res.to(rd)
            fused = t(lambda: linear_add_layer_norm_bf16(x, W, b, r, g, be, 1e-5, out_dtype=od))
            sep = t(lambda: add_layer_norm(linear_bf16(x, W, b), r, g, be, 1e-5, out_dtype=od))
            print(f"   + residual {rd} + layer norm -> {od}: fused {fused:.1f} us, two kernels {sep:.1f} us")

# launch + weight prologue + one tile per workgroup, against the full batch
for r in (64 * 256, 64 * 256 * 2, 64 * 256 * 4):
    x = torch.randn(r, 384, device=dev).bfloat16()
    W = (torch.randn(384, 384, device=dev) / 20).bfloat16()
    b = torch.randn(384, device=dev)
    print(f"rows={r} 384->384: {t(lambda: linear_bf16(x, W, b)):.1f} us")
