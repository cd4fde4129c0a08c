"""One-off probe: the bench's three-layer step at batch 64 as ONE chain against TWO independent chains of batch 32
on parallel branches of one HIP graph (same total work; do the branches fill each other's launch / ramp / tail gaps?)."""
import sys, time, torch
from asr_amd import LMUFFT

dev = torch.device("cuda:0")
T, D, M, Hd = 400, 384, 192, 384


def make(B, seed):
    torch.manual_seed(seed)
    layers = []
    for _ in range(3):
        l = LMUFFT(D, Hd, M, T, 1000.0)
        torch.nn.init.xavier_uniform_(l.W_u.weight); torch.nn.init.xavier_uniform_(l.W_h.weight)
        layers.append(l.to(dev))
    x = torch.randn(B, T, D, device=dev).bfloat16().requires_grad_(True)
    gy = torch.randn(B, T, Hd, device=dev).bfloat16()
    return layers, x, gy


def step(layers, x, gy):
    for l in layers:
        for p in l.parameters():
            p.grad = None
    x.grad = None
    y = x
    for l in layers:
        y, _ = l(y)
    y.backward(gy)


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def capture(chains):
    for c in chains:
        for _ in range(3):
            step(*c)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream(dev) for _ in chains[1:]]
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream(dev)
        for s in side:
            s.wait_stream(cur)
        for c, s in zip(chains[1:], side):
            with torch.cuda.stream(s):
                step(*c)
        step(*chains[0])
        for s in side:
            cur.wait_stream(s)
    return g.replay


one = capture([make(64, 1)])
print("one chain of 64:        %.4f ms" % timed(one))
two = capture([make(32, 2), make(32, 3)])
print("two chains of 32:       %.4f ms" % timed(two))
four = capture([make(16, 4), make(16, 5), make(16, 6), make(16, 7)])
print("four chains of 16:      %.4f ms" % timed(four))
half = capture([make(32, 8)])
print("one chain of 32:        %.4f ms" % timed(half))
