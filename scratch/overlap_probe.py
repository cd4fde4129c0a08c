"""Do two independent LMUFFT layer steps (fwd + bwd, bf16) finish sooner when a HIP graph runs them on
two parallel branches than back to back on one?  A probe for how much co-scheduling the layer's
kernels could buy inside one step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from asr_amd import LMUFFT

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T = 64, 400
layers = [LMUFFT(384, 384, 192, T, 1000).to(dev) for _ in range(2)]
xs = [torch.randn(B, T, 384, device=dev).bfloat16().requires_grad_() for _ in range(2)]
gs = [torch.randn(B, T, 384, device=dev).bfloat16() for _ in range(2)]


def step(i):
    y, hn = layers[i](xs[i])
    y.backward(gs[i])


def capture(parallel):
    side = torch.cuda.Stream()
    for _ in range(3):
        step(0); step(1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if parallel:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                step(1)
            step(0)
            main.wait_stream(side)
        else:
            step(0); step(1)
    return g


for parallel in (False, True):
    g = capture(parallel)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print("parallel branches" if parallel else "one stream      ", f"{a.elapsed_time(b) / 50 * 1e3:.1f} us per two layer steps")
