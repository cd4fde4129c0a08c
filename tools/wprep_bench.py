"""Times the preparation of a training step's Linear operands: one launch per weight against one launch for all."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from asr_amd import linear as L

dev = torch.device("cuda:0")
torch.manual_seed(0)
mk = lambda n, k: (torch.nn.Parameter(torch.randn(n, k, device=dev)), torch.nn.Parameter(torch.randn(n, device=dev)))
groups = [([w], [b]) for w, b in [mk(384, 80)]]
for _ in range(3):
    (wq, bq), (wk, bk), (wv, bv), (wo, bo) = mk(384, 384), mk(384, 384), mk(384, 384), mk(384, 384)
    groups += [([wq, wk, wv], [bq, bk, bv]), ([wo], [bo])]
groups += [([w], [b]) for w, b in [mk(384, 384), mk(29, 384)]]


def t(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n // 10 * 10) * 1e3


def per_layer():
    L._PREPARED.clear()
    for w, b in groups:
        L.linear_weight_prep(w, b)


print(f"one launch per group: {t(per_layer):.1f} us; one launch for all: {t(lambda: L.prepare_weights(groups)):.1f} us")
