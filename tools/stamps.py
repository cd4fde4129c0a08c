"""Per-workgroup timeline of k_wgrad_bf16 from a library built with -DLMUFFT_STAMPS
(tools/build_stamps.sh; run with LMUFFT_LIBRARY pointing at it)."""
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
rel = (a[:, :4] - t0) / 100.0
print("blocks", len(a))
print("start   min/med/max", np.min(rel[:, 0]), np.median(rel[:, 0]), np.max(rel[:, 0]))
for i, name in enumerate(["setup", "loop", "epilogue"]):
    d = (a[:, i + 1] - a[:, i]) / 100.0
    print(f"{name:18s} min/med/max us", d.min(), np.median(d), d.max())
print("end     min/med/max", np.min(rel[:, 3]), np.median(rel[:, 3]), np.max(rel[:, 3]))
for i, name in [(4, "mfma phase"), (5, "store+fetch"), (6, "barrier")]:
    d = a[:, i]
    print(f"{name:18s} shader clocks min/med/max", d.min(), np.median(d), d.max())
