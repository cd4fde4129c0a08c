"""Attention core: HIP kernel vs the materialised torch formulation (the reference's), inference shapes."""
import math, time, torch
from asr_amd import _lib
from asr_amd.attention import masked_attention
dev = torch.device("cuda:0")
def torch_ref(q, k, v, H, lengths):
    B, T, C = q.shape; dk = C // H
    sp = lambda t: t.view(B, T, H, dk).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-2, -1) / math.sqrt(dk)
    valid = torch.arange(T, device=q.device)[None] < lengths[:, None]
    s = s.masked_fill(~(valid[:, :, None] & valid[:, None, :])[:, None], -1e9)
    return (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(B, T, C)
for (B, T, H, dk, dt) in [(64, 400, 6, 64, torch.float32), (256, 400, 6, 64, torch.float32), (256, 400, 6, 64, torch.bfloat16), (32, 3000, 6, 64, torch.float32)]:
    q, k, v = (torch.randn(B, T, H * dk, device=dev).to(dt) for _ in range(3))
    lengths = torch.randint(T // 2, T + 1, (B,), device=dev)
    def timeit(f, n=5):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    tk = timeit(lambda: masked_attention(q, k, v, H, lengths))
    try:
        tt = timeit(lambda: torch_ref(q, k, v, H, lengths))
    except Exception as e:
        tt = float("nan")
    flops = 4.0 * B * H * T * T * dk
    print(f"B={B} T={T} {str(dt)[6:]}: kernel {tk:.3f} ms ({flops / tk / 1e9:.1f} TF/s), torch materialised {tt:.3f} ms", flush=True)
