"""Is the bench step bound by the host (launch/enqueue time) or by the GPU?"""
import time, torch
from asr_amd import LMUFFT
from asr_amd.parallel import GradBuckets
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = [LMUFFT(384, 384, 192, 400, 1000).to(dev) for _ in range(3)]
buckets = GradBuckets(list(reversed(layers)))
x = torch.randn(64, 400, 384, device=dev).bfloat16().requires_grad_(True)
gy = torch.randn(64, 400, 384, device=dev).bfloat16()
def step():
    buckets.zero_grad(); x.grad = None
    y = x
    for l in layers: y, _ = l(y)
    y.backward(gy); buckets.finish()
for _ in range(10): step()
torch.cuda.synchronize()
for K in (30, 100):
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step", flush=True)
# forward only / backward only split of host time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
