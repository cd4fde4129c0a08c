"""Event-bracketed time of the backward's kernels for one layer at the benchmark shape (LMUFFT_LIBRARY selects the build)."""
import sys, torch
from asr_amd import LMUFFT, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
l = LMUFFT(384, 384, 192, 400, 1000).to(dev)
x = torch.randn(B, 400, 384, device=dev).bfloat16().requires_grad_()
gy = torch.randn(B, 400, 384, device=dev).bfloat16()
for _ in range(5):
    y, hn = l(x); y.backward(gy)
torch.cuda.synchronize()
_lib.profile_reset(); _lib.profile_enable(True)
for _ in range(20):
    y, hn = l(x); y.backward(gy)
torch.cuda.synchronize()
_lib.profile_enable(False)
print(" ".join(f"{k}={v[0]/v[1]*1e3:.2f}us" for k, v in sorted(_lib.profile_read().items())))
