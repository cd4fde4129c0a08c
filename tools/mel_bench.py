"""Times the mel frontend at the inference benchmark shape (256 x 4 s) and a 30 s batch."""
import torch
from asr_amd import MelFrontend, _lib

dev = torch.device("cuda:0")
fe = MelFrontend()
for (B, L) in [(256, 63840), (32, 479840), (16, 63840)]:
    w = torch.randn(B, L, device=dev) * 0.1
    for _ in range(3):
        fe(w)
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_reset()
    for _ in range(10):
        fe(w)
    torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    frames = B * (1 + L // 160)
    us = {k: v[0] / v[1] * 1e3 for k, v in prof.items() if v[1]}
    tot = sum(us.values())
    print(B, L, {k: round(v, 1) for k, v in us.items()}, f"{frames / tot:.1f} frames/us",
          f"{frames * 960 / tot / 1e3:.1f} GB/s algorithmic", flush=True)
