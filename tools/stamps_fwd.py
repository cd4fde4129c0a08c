"""Per-workgroup timeline of k_conv_proj_fwd_bf16 from a library built with -DLMUFFT_STAMPS=6
(tools/build_stamps.sh 6; run with LMUFFT_LIBRARY pointing at it)."""
import ctypes, sys, numpy as np, torch
from asr_amd import LMUFFT, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
l = LMUFFT(384, 384, 192, 400, 1000).to(dev)
x = torch.randn(B, 400, 384, device=dev).bfloat16().requires_grad_()
for _ in range(3):
    y, hn = l(x)
torch.cuda.synchronize()
N = 4096
buf = (ctypes.c_longlong * (8 * N))()
raw = ctypes.CDLL(_lib.LIB_PATH)
print("rc", raw.lmufft_debug_stamps(buf, 8 * N))
a = np.array(buf[:], dtype=np.int64).reshape(N, 8)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print("blocks", len(a), "kernel span us", (a[:, 6].max() - t0) / 100.0)
for j in sorted(set(a[:, 4])):
    q = a[a[:, 4] == j]
    rel = (q - t0) / 100.0
    med = lambda v: f"{np.median(v):.2f}/{v.max():.2f}"
    print(f"tile {j}: start {med(rel[:,0])} | u {med(rel[:,1]-rel[:,0])} | conv loop {med(rel[:,2]-rel[:,1])} | handover {med(rel[:,3]-rel[:,2])} | "
          f"proj loop {med(rel[:,5]-rel[:,3])} | epilogue {med(rel[:,6]-rel[:,5])} | end {med(rel[:,6])}")
