"""Times the 384 -> 384 Linear on 102,400 rows with the library named by LMUFFT_LIBRARY."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from asr_amd.linear import linear_bf16

dev = torch.device("cuda:0")
for rows in (16384, 102400):
    x = torch.randn(rows, 384, device=dev).bfloat16()
    W = (torch.randn(384, 384, device=dev) / 20).bfloat16()
    b = torch.randn(384, device=dev)
    for _ in range(5):
        linear_bf16(x, W, b)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        linear_bf16(x, W, b)
    e.record()
    torch.cuda.synchronize()
    print(os.environ.get("LMUFFT_LIBRARY", "in-tree"), rows, f"{a.elapsed_time(e) / 30 * 1e3:.1f} us")
