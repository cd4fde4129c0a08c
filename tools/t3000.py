"""Long-utterance check (3000 frames): both dtypes against the oracle, and kernel timings at batch 32."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import lmufft_oracle as orc
from asr_amd import LMUFFT, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
T, D, M, Hd = 3000, 384, 192, 384
layer = LMUFFT(D, Hd, M, T, 1000)
torch.nn.init.xavier_uniform_(layer.W_u.weight); torch.nn.init.xavier_uniform_(layer.W_h.weight)
layer = layer.to(dev)
sd = {k: v.detach().cpu().numpy() for k, v in layer.state_dict().items() if k != "fft_H"}
x = torch.randn(2, T, D, device=dev, requires_grad=True)
gh = torch.randn(2, T, Hd, device=dev)
h, hn = layer(x); (h * gh).sum().backward()
ref = orc.lmufft_backward(x.detach().cpu().numpy(), sd["W_u.weight"], sd["W_u.bias"], sd["W_h.weight"], sd["W_h.bias"], sd["H"], gh.cpu().numpy())
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
print("fp32 h", rel(h.detach().cpu().numpy(), ref["h"]), "gx", rel(x.grad.cpu().numpy(), ref["gx"]),
      "gW_h", rel(layer.W_h.weight.grad.cpu().numpy(), ref["gW_h"]), "gw_u", rel(layer.W_u.weight.grad.cpu().numpy(), ref["gw_u"]), flush=True)
xb = x.detach().bfloat16().requires_grad_(True)
layer.zero_grad()
hb, _ = layer(xb); (hb * gh.bfloat16()).sum().backward()
print("bf16 h", rel(hb.detach().float().cpu().numpy(), ref["h"]), "gx", rel(xb.grad.float().cpu().numpy(), ref["gx"]),
      "gW_h", rel(layer.W_h.weight.grad.cpu().numpy(), ref["gW_h"]), flush=True)
for dt, algo in ((torch.bfloat16, "scan"), (torch.bfloat16, "toeplitz"), (torch.float32, "toeplitz"), (torch.float32, "fft")):
    layer.conv_algo = algo
    xx = torch.randn(32, T, D, device=dev).to(dt).requires_grad_(True)
    gg = torch.randn(32, T, Hd, device=dev).to(dt)
    for _ in range(2):
        layer.zero_grad(); o, _ = layer(xx); o.backward(gg)
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        layer.zero_grad(); o, _ = layer(xx); o.backward(gg)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / 3
    prof = _lib.profile_read(); _lib.profile_enable(False)
    print(dt, algo, f"{dtm*1e3:.2f} ms/step, {32*T/dtm/1e6:.1f} M frames/s", {k: round(v[0]/v[1]*1e3) for k, v in prof.items() if v[1]}, flush=True)
