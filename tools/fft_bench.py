"""Times the two convolution routes of the fp32 path at a few shapes (per-kernel HIP events)."""
import sys
import torch
from asr_amd import _lib, causal_conv, causal_conv_adjoint, prepare_impulse_fft

dev = torch.device("cuda:0")
for (B, T, M) in [(64, 100, 192), (64, 128, 192), (64, 160, 192), (64, 200, 192), (64, 256, 192), (64, 300, 192), (64, 400, 192), (64, 512, 192), (16, 400, 192), (256, 400, 192)]:
    u = torch.relu(torch.randn(B, T, device=dev))
    gm = torch.randn(B, T, M, device=dev)
    H = torch.randn(M, T, device=dev) * 0.05
    prep = prepare_impulse_fft(H, T)
    for algo in ("toeplitz", "fft"):
        fp = prep if algo == "fft" else None
        for _ in range(3):
            causal_conv(u, H, algo, fp); causal_conv_adjoint(gm, H, algo, fp)
        torch.cuda.synchronize()
        _lib.profile_enable(True); _lib.profile_reset()
        for _ in range(10):
            causal_conv(u, H, algo, fp); causal_conv_adjoint(gm, H, algo, fp)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        per = {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items() if v[1]}
        print(B, T, M, algo, f"pair {sum(v[0] for v in prof.values()) / 10 * 1e3:.1f} us", per, flush=True)
