"""Per-workgroup timeline of k_conv_fwd_bf16 from a library built with -DLMUFFT_STAMPS=2
(scratch/build_stamps.sh 2; run with LMUFFT_LIBRARY pointing at it)."""
import ctypes, numpy as np, torch
from asr_amd import LMUFFT, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
l = LMUFFT(384, 384, 192, 400, 1000).to(dev)
x = torch.randn(64, 400, 384, device=dev).bfloat16()
with torch.no_grad():
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
rel = (a[:, :4] - t0) / 100.0
print("blocks", len(a), "kernel span us", rel[:, 3].max())
for kend in sorted(set(a[:, 4])):
    s = a[:, 4] == kend
    r = rel[s]
    print(f"kend {kend:4d} n={s.sum():3d} start med {np.median(r[:,0]):6.2f} max {r[:,0].max():6.2f} | setup {np.median(r[:,1]-r[:,0]):5.2f} loop {np.median(r[:,2]-r[:,1]):5.2f} epi {np.median(r[:,3]-r[:,2]):5.2f} | end med {np.median(r[:,3]):6.2f} max {r[:,3].max():6.2f}")
