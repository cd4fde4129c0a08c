"""Per-workgroup timeline of k_proj_fwd_glds_bf16 from a library built with -DLMUFFT_STAMPS=4."""
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
rel = (a[:, :3] - a[:, 0].min()) / 100.0
print("blocks", len(a), "span to the end of the tile stores (us)", rel[:, 2].max())
print("loop start  med/max", np.median(rel[:, 0]), rel[:, 0].max())
print("gemm loop   med/max", np.median(rel[:, 1] - rel[:, 0]), (rel[:, 1] - rel[:, 0]).max())
print("tile store  med/max", np.median(rel[:, 2] - rel[:, 1]), (rel[:, 2] - rel[:, 1]).max())
