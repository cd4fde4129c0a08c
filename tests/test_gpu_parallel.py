"""GPU tests of the data-parallel step machinery on one rank: HIP-graph replay of a step cut into
segments, with RCCL (backend "nccl") collectives actually issued in a one-rank group, buckets that
mix LMUFFT layers with ordinary modules, and the reference's own wrapper, DistributedDataParallel
(scripts/train_flexible.py:285-289)."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def nccl_one_rank():
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        torch.cuda.synchronize(dev)
        dist.destroy_process_group()


def _close(got, want, what, rtol=2e-2):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    want = want.detach().float().cpu().numpy().astype(np.float64)
    err = np.abs(got - want).max()
    assert err <= rtol * np.abs(want).max() + 1e-12, (what, err, np.abs(want).max())


def _stack(dev, n=3):
    from asr_amd import LMUFFT

    torch.manual_seed(3)
    return [LMUFFT(48, 48, 24, 64, 16).to(dev) for _ in range(n)]


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("bucket_mb", [None, float("inf")])
def test_graph_pipeline_with_rccl_one_rank(nccl_one_rank, overlap, bucket_mb):
    """capture_pipeline + finish with the all-reduces really issued on RCCL (one-rank group): the
    replayed, exchanged gradients equal plain autograd's, on three replays in a row."""
    from asr_amd.parallel import GradBuckets

    dev = nccl_one_rank
    layers = _stack(dev)
    ref_layers = copy.deepcopy(layers)
    buckets = GradBuckets(list(reversed(layers)), bucket_mb=bucket_mb, overlap=overlap, exchange_single_rank=True)
    x = torch.randn(4, 64, 48, device=dev).requires_grad_(True)
    gy = torch.randn(4, 64, 48, device=dev)
    st = {}

    def seg_first():
        ins, outs = [], []
        y = x
        for i, l in enumerate(layers):
            xin = y if i == 0 else y.detach().requires_grad_(True)
            y, _ = l(xin)
            ins.append(xin)
            outs.append(y)
        st["ins"], st["outs"] = ins, outs
        outs[-1].backward(gy)

    segs = [seg_first] + [(lambda k=k: st["outs"][k].backward(st["ins"][k + 1].grad)) for k in (1, 0)]
    for _ in range(2):  # eager warm-up through the same protocol
        buckets.zero_grad()
        x.grad = None
        for s in segs:
            s()
        buckets.finish()
    st.clear()  # drop the eager steps' autograd graphs (their accumulation nodes are tied to the eager stream)
    x.grad = None
    step = buckets.capture_pipeline(segs, overlap=overlap)
    # reference: plain autograd on copies
    xr = x.detach().clone().requires_grad_(True)
    y = xr
    for l in ref_layers:
        y, _ = l(y)
    y.backward(gy)
    for rep in range(3):
        step()
        torch.cuda.synchronize(dev)
        for l, r in zip(layers, ref_layers):
            for (n, p), q in zip(l.named_parameters(), r.parameters()):
                _close(p.grad, q.grad, f"{n} replay {rep}", rtol=1e-4)
        _close(x.grad, xr.grad, f"gx replay {rep}", rtol=1e-4)
    buckets.remove()


def test_graph_replay_mixed_bucket_matches_eager():
    """A bucket that holds an LMUFFT layer (gradient sink) AND an ordinary Linear head (autograd
    accumulates `grad += g` into the bucket view): the captured step clears the head's views
    itself, so replays do not pile up on earlier steps' gradients."""
    from asr_amd import LMUFFT
    from asr_amd.parallel import GradBuckets

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    layer = LMUFFT(48, 48, 24, 64, 16).to(dev)
    head = torch.nn.Linear(48, 7).to(dev)
    ref_layer, ref_head = copy.deepcopy(layer), copy.deepcopy(head)
    buckets = GradBuckets([head, layer], bucket_mb=float("inf"))
    assert len(buckets.buckets) == 1
    x = torch.randn(3, 64, 48, device=dev)
    gy = torch.randn(3, 64, 7, device=dev)

    def fn():
        h, _ = layer(x)
        head(h).backward(gy)

    for _ in range(2):
        buckets.zero_grad()
        fn()
        buckets.finish()
    step = buckets.capture_step(fn)
    h, _ = ref_layer(x)
    ref_head(h).backward(gy)
    for rep in range(3):
        step()
        torch.cuda.synchronize(dev)
        for (n, p), q in zip(list(layer.named_parameters()) + list(head.named_parameters()),
                             list(ref_layer.parameters()) + list(ref_head.parameters())):
            _close(p.grad, q.grad, f"{n} replay {rep}", rtol=1e-4)
    buckets.remove()


def test_distributed_data_parallel_wrapper_one_rank(nccl_one_rank):
    """The reference wraps its model in DistributedDataParallel (scripts/train_flexible.py:287);
    the layer must work under it: ordinary Parameters, gradients through autograd, RCCL buckets."""
    from torch.nn.parallel import DistributedDataParallel as DDP

    dev = nccl_one_rank
    layers = _stack(dev, 2)
    model = torch.nn.Sequential()
    ref = copy.deepcopy(layers)

    class Stack(torch.nn.Module):
        def __init__(self, ls):
            super().__init__()
            self.ls = torch.nn.ModuleList(ls)

        def forward(self, x):
            for l in self.ls:
                x, _ = l(x)
            return x

    del model
    ddp = DDP(Stack(layers), device_ids=[0], broadcast_buffers=False)
    x = torch.randn(3, 64, 48, device=dev)
    gy = torch.randn(3, 64, 48, device=dev)
    for _ in range(2):
        ddp.zero_grad(set_to_none=True)
        ddp(x).backward(gy)
    y = x
    for l in ref:
        y, _ = l(y)
    y.backward(gy)
    torch.cuda.synchronize(dev)
    for l, r in zip(layers, ref):
        for (n, p), q in zip(l.named_parameters(), r.parameters()):
            _close(p.grad, q.grad, n, rtol=1e-4)


@pytest.mark.parametrize("overlap", [True, False])
def test_graph_with_captured_exchange_matches_eager(nccl_one_rank, overlap):
    """capture_graph_exchange: the whole step and its RCCL all-reduces in one HIP graph (one-rank group, so the
    exchange changes nothing): replays reproduce the eager step's gradients bit for bit, twice in a row."""
    from asr_amd import LMUFFT
    from asr_amd.parallel import GradBuckets

    dev = nccl_one_rank
    torch.manual_seed(5)
    layers = [LMUFFT(64, 64, 32, 50, 40.0).to(dev) for _ in range(3)]
    buckets = GradBuckets(list(reversed(layers)), overlap=overlap, exchange_single_rank=True)
    x = torch.randn(4, 50, 64, device=dev).bfloat16().requires_grad_(True)
    gy = torch.randn(4, 50, 64, device=dev).bfloat16()
    st = {}

    def seg_first():
        ins, outs = [], []
        y = x
        for i, l in enumerate(layers):
            xin = y if i == 0 else y.detach().requires_grad_(True)
            y, _ = l(xin)
            ins.append(xin)
            outs.append(y)
        st["ins"], st["outs"] = ins, outs
        outs[-1].backward(gy)

    def seg_back(k):
        return lambda: st["outs"][k].backward(st["ins"][k + 1].grad)

    segments = [seg_first] + [seg_back(k) for k in range(len(layers) - 2, -1, -1)]
    for _ in range(2):
        buckets.zero_grad()
        x.grad = None
        for s in segments:
            s()
        buckets.finish()
    want = buckets.arena.clone()
    gx = x.grad.clone()
    st.clear()
    x.grad = None
    step = buckets.capture_graph_exchange(segments, overlap=overlap)
    for _ in range(2):
        buckets.arena.zero_()
        step()
        torch.cuda.synchronize()
        assert torch.equal(buckets.arena, want)
        assert torch.equal(x.grad, gx)
    buckets.remove()


def test_async_weight_grad_matches_synchronous():
    """Layers with `async_weight_grad` run their parameter gradients on a side stream (C ABI
    lmufft_backward_data_bf16 + lmufft_backward_weights_bf16); GradBuckets joins them.  Eager steps and a
    replayed one-graph step give the synchronous step's gradients bit for bit."""
    from asr_amd import LMUFFT
    from asr_amd.parallel import GradBuckets

    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    layers = [LMUFFT(64, 64, 32, 80, 40.0).to(dev) for _ in range(3)]
    buckets = GradBuckets(list(reversed(layers)), overlap=False)
    x = torch.randn(6, 80, 64, device=dev).bfloat16().requires_grad_(True)
    gy = torch.randn(6, 80, 64, device=dev).bfloat16()
    st = {}

    def seg_first():
        ins, outs = [], []
        y = x
        for i, l in enumerate(layers):
            xin = y if i == 0 else y.detach().requires_grad_(True)
            y, _ = l(xin)
            ins.append(xin)
            outs.append(y)
        st["ins"], st["outs"] = ins, outs
        outs[-1].backward(gy)

    def seg_back(k):
        return lambda: st["outs"][k].backward(st["ins"][k + 1].grad)

    segments = [seg_first] + [seg_back(k) for k in range(len(layers) - 2, -1, -1)]

    def eager():
        buckets.zero_grad()
        x.grad = None
        for s in segments:
            s()
        buckets.finish()
        torch.cuda.synchronize()
        return buckets.arena.clone(), x.grad.clone()

    want, gx = eager()
    for l in layers:
        l.async_weight_grad = True
    for _ in range(2):
        got, gx2 = eager()
        assert torch.equal(got, want) and torch.equal(gx2, gx)
    st.clear()
    x.grad = None
    step = buckets.capture_graph_exchange(segments, overlap=False)
    for _ in range(2):
        buckets.arena.zero_()
        step()
        torch.cuda.synchronize()
        assert torch.equal(buckets.arena, want) and torch.equal(x.grad, gx)
    buckets.remove()


def test_async_weight_grad_with_accumulated_backwards():
    """Two backwards in one step (gradient accumulation) on a layer with `async_weight_grad`: the first backward's
    weight gradient runs on the side stream and writes the bucket; the second returns its gradients to autograd,
    which adds them into the same memory on the main stream -- only after the side stream has been joined.  The
    accumulated gradients equal the synchronous step's bit for bit, at a size where the side-stream kernel is
    still running when the second backward starts."""
    from asr_amd import LMUFFT
    from asr_amd.parallel import GradBuckets

    dev = torch.device("cuda:0")
    torch.manual_seed(13)
    layer = LMUFFT(384, 384, 192, 400, 1000.0).to(dev)
    buckets = GradBuckets([layer], overlap=False)
    xs = [torch.randn(32, 400, 384, device=dev).bfloat16() for _ in range(2)]
    gys = [torch.randn(32, 400, 384, device=dev).bfloat16() for _ in range(2)]

    def step():
        buckets.zero_grad()
        with buckets.accumulate():
            for x, gy in zip(xs, gys):
                y, _ = layer(x)
                y.backward(gy)
        buckets.finish()
        torch.cuda.synchronize()
        return buckets.arena.clone()

    want = step()
    layer.async_weight_grad = True
    for _ in range(3):
        assert torch.equal(step(), want)
    buckets.remove()
