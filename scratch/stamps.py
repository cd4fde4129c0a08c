import ctypes, numpy as np, torch
from asr_amd import LMUFFT, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
l = LMUFFT(384, 384, 192, 400, 1000).to(dev)
x = torch.randn(64, 400, 384, device=dev).bfloat16()
with torch.no_grad():
    for _ in range(3):
        y, _ = l(x)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (8 * 256))()
raw = ctypes.CDLL(_lib.LIB_PATH)
print("rc", raw.lmufft_debug_stamps(buf, 8 * 256))
a = np.array(buf[:], dtype=np.int64).reshape(256, 8)
a = a[a[:, 0] > 0][:, :4]
t0 = a[:, 0].min()
rel = (a - t0) / 100.0
print("blocks", len(a))
print("start   min/med/max", np.min(rel[:,0]), np.median(rel[:,0]), np.max(rel[:,0]))
for i, name in enumerate(["setup", "loop", "epilogue"]):
    d = (a[:, i+1] - a[:, i]) / 100.0
    print(f"{name:18s} min/med/max us", d.min(), np.median(d), d.max())
print("end     min/med/max", np.min(rel[:,3]), np.median(rel[:,3]), np.max(rel[:,3]))
