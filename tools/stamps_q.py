"""Per-workgroup timeline of k_gu_q_bf16 from a library built with -DLMUFFT_STAMPS=5
(tools/build_stamps.sh 5; run with LMUFFT_LIBRARY pointing at it)."""
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
t0 = a[:, 0].min()
print("blocks", len(a), "kernel span us", (a[:, 3].max() - t0) / 100.0)
for j in sorted(set(a[:, 4])):
    q = a[a[:, 4] == j]
    rel = (q[:, :4] - t0) / 100.0
    print(f"slot {j}: start med {np.median(rel[:,0]):.2f} max {rel[:,0].max():.2f} | loop med {np.median(rel[:,1]-rel[:,0]):.2f} max {(rel[:,1]-rel[:,0]).max():.2f} | "
          f"reduce med {np.median(rel[:,2]-rel[:,1]):.2f} | combine+store med {np.median(rel[:,3]-rel[:,2]):.2f} | end med {np.median(rel[:,3]):.2f} max {rel[:,3].max():.2f}")
