"""Per-workgroup timeline of k_conv_bwd2_bf16 from a library built with -DLMUFFT_STAMPS=3."""
import ctypes, numpy as np, torch
from asr_amd import LMUFFT, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
l = LMUFFT(384, 384, 192, 400, 1000).to(dev)
x = torch.randn(64, 400, 384, device=dev).bfloat16().requires_grad_()
for _ in range(3):
    y, hn = l(x)
    (y.float().square().mean() + hn.float().mean()).backward()
torch.cuda.synchronize()
N = 4096
buf = (ctypes.c_longlong * (8 * N))()
raw = ctypes.CDLL(_lib.LIB_PATH)
print("rc", raw.lmufft_debug_stamps(buf, 8 * N))
a = np.array(buf[:], dtype=np.int64).reshape(N, 8)
a = a[a[:, 0] > 0]
rel = (a[:, :4] - a[:, 0].min()) / 100.0
print("blocks", len(a), "kernel span us", rel[:, 3].max())
for s0 in sorted(set(a[:, 4])):
    r = rel[a[:, 4] == s0]
    print(f"s0 {s0:4d} n={len(r):3d} start med {np.median(r[:,0]):5.2f} max {r[:,0].max():5.2f} | window {np.median(r[:,1]-r[:,0]):5.2f} loop {np.median(r[:,2]-r[:,1]):5.2f} reduce+write {np.median(r[:,3]-r[:,2]):5.2f} | end med {np.median(r[:,3]):5.2f} max {r[:,3].max():5.2f}")
