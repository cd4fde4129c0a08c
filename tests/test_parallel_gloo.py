"""World-size-2 test of the data-parallel gradient exchange on CPU (gloo).

No kernel runs here: the test drives asr_amd.parallel.GradBuckets with the
real parameter set of a stack of LMUFFT layers and hand-made gradients, and
checks that every rank ends up with the mean gradient, that the flat buckets
alias the parameters' .grad tensors, and that constant buffers (A, B, H, fft_H)
are never communicated."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _emulate_layer_backward(layer, value):
    """What the layer's backward does with a gradient sink (asr_amd/functional.py): the kernels
    write the four parameter gradients into the sink's views, then commit."""
    sink = layer.grad_sink
    assert sink is not None and sink.usable([True] * 4)
    for v in sink.views:
        v.fill_(value)
    sink.commit()


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from asr_amd import LMUFFT
        from asr_amd.parallel import GradBuckets

        torch.manual_seed(0)  # same parameters on every rank
        layers = [LMUFFT(12, 10, 6, 16, 8) for _ in range(2)]
        head = torch.nn.Linear(10, 5)  # an ordinary module fed by autograd itself
        buckets = GradBuckets(list(reversed(layers)) + [head.parameters()])
        assert len(buckets.buckets) == 3
        n_params = sum(p.numel() for l in layers for p in l.parameters())
        # buffers (A, B, H, fft_H) are not in the buckets; every gradient starts 256-byte aligned
        assert n_params <= sum(b.numel() for b in buckets.buckets[:2]) < n_params + 2 * 4 * 64
        lo = min(b.data_ptr() for b in buckets.buckets)
        hi = max(b.data_ptr() + b.numel() * 4 for b in buckets.buckets)
        for step in range(3):
            buckets.zero_grad()
            xin = torch.full((3, 10), float(rank + 1 + step))
            head(xin).sum().backward()  # real autograd: hooks fire, fresh tensors are moved into the bucket
            if step < 2:
                for l in reversed(layers):
                    _emulate_layer_backward(l, float(rank + 1 + step))
            else:
                # a step in which autograd delivers the layer gradients itself (sink not used)
                for l in reversed(layers):
                    for p in l.parameters():
                        p.grad = torch.full_like(p, float(rank + 1 + step))
                        for hook in p._post_accumulate_grad_hooks.values():
                            hook(p)
            buckets.finish()
            want = sum(r + 1 + step for r in range(world)) / world
            for l in layers:
                for p in l.parameters():
                    assert torch.allclose(p.grad, torch.full_like(p, want)), (rank, step)
                    assert lo <= p.grad.data_ptr() < hi and p.grad.data_ptr() % 64 == 0
            assert torch.allclose(head.weight.grad, torch.full_like(head.weight, 3 * want))
            assert torch.allclose(head.bias.grad, torch.full_like(head.bias, 3.0))
            assert lo <= head.weight.grad.data_ptr() < hi
        # the form used by graph-replayed steps: nothing is launched during the backward, finish()
        # exchanges the whole arena in one all-reduce
        assert buckets.arena is not None and buckets.arena.numel() == sum(b.numel() for b in buckets.buckets)
        assert all(b.data_ptr() >= buckets.arena.data_ptr() for b in buckets.buckets)
        buckets.overlap = False
        for step in range(2):
            buckets.zero_grad()
            head(torch.full((3, 10), float(rank + 1 + step))).sum().backward()
            for l in reversed(layers):
                _emulate_layer_backward(l, float(rank + 1 + step))
            assert not buckets._handles
            buckets.finish()
            want = sum(r + 1 + step for r in range(world)) / world
            for l in layers:
                for p in l.parameters():
                    assert torch.allclose(p.grad, torch.full_like(p, want)), (rank, step)
            assert torch.allclose(head.weight.grad, torch.full_like(head.weight, 3 * want))
        # two backwards per step (gradient accumulation): only legal in accumulate mode, where the
        # exchange waits for finish(); the means must cover both backwards of both ranks
        buckets.overlap = True
        buckets.zero_grad()
        with buckets.accumulate():
            for rep in range(2):
                head(torch.full((3, 10), float(rank + 1))).sum().backward()
            for l in reversed(layers):
                _emulate_layer_backward(l, float(rank + 1))
                for p in l.parameters():  # the second backward: autograd adds into the same views
                    p.grad.add_(10.0 * (rank + 1))
                    for hook in p._post_accumulate_grad_hooks.values():
                        hook(p)
            assert not buckets._handles
            buckets.finish()
        want = sum(11.0 * (r + 1) for r in range(world)) / world
        for l in layers:
            for p in l.parameters():
                assert torch.allclose(p.grad, torch.full_like(p, want)), rank
        want_head = 2 * 3 * sum(r + 1 for r in range(world)) / world
        assert torch.allclose(head.weight.grad, torch.full_like(head.weight, want_head))
        buckets.remove()
        assert all(l.grad_sink is None for l in layers)
        # merged buckets: the two layers and the head share one flat buffer and one all-reduce
        merged = GradBuckets(list(reversed(layers)) + [head.parameters()], bucket_mb=float("inf"))
        assert len(merged.buckets) == 1
        merged.zero_grad()
        head(torch.full((3, 10), float(rank + 1))).sum().backward()
        assert not merged._handles  # the layers' gradients are still missing
        for l in reversed(layers):
            _emulate_layer_backward(l, float(rank + 1))
        assert len(merged._handles) == 1
        merged.finish()
        want = sum(r + 1 for r in range(world)) / world
        assert all(torch.allclose(p.grad, torch.full_like(p, want)) for l in layers for p in l.parameters())
        assert torch.allclose(head.weight.grad, torch.full_like(head.weight, 3 * want))
        merged.remove()
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_grad_buckets_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def _fire_hooks(p):
    for hook in p._post_accumulate_grad_hooks.values():
        hook(p)


def test_grad_buckets_single_process():
    from asr_amd import LMUFFT
    from asr_amd.parallel import GradBuckets

    layer = LMUFFT(12, 10, 6, 16, 8)
    b = GradBuckets([layer])
    b.zero_grad()
    assert all(p.grad is None for p in layer.parameters())
    _emulate_layer_backward(layer, 1.0)
    assert not layer.grad_sink.usable([True] * 4)
    # overlap mode: the bucket's all-reduce was launched when its last gradient arrived, so a second
    # backward in the same step (autograd accumulating into the same views) must be refused loudly
    with pytest.raises(RuntimeError, match="accumulate"):
        layer.grad_sink.on_late()  # what the layer's backward calls when it runs again in the same step
    # accumulate mode: nothing is launched before finish(), any number of backwards is fine
    b.zero_grad()
    with b.accumulate():
        _emulate_layer_backward(layer, 1.0)
        layer.grad_sink.on_late()  # no complaint here
        for p in layer.parameters():
            p.grad.add_(1.0)
            _fire_hooks(p)
        b.finish()
    assert all(torch.equal(p.grad, torch.full_like(p, 2.0)) for p in layer.parameters())
    b.zero_grad()
    assert layer.grad_sink.usable([True] * 4) and not layer.grad_sink.usable([True, True, False, True])
    # a bucket that did not receive all of its gradients is an error, not a silent skip
    head = torch.nn.Linear(4, 3)
    b2 = GradBuckets([head])
    b2.zero_grad()
    assert head.weight.grad is None and head.bias.grad is None  # a step starts empty; nothing is cleared
    head.weight.grad = torch.ones_like(head.weight)
    _fire_hooks(head.weight)  # the bias never gets its gradient
    assert head.weight.grad.data_ptr() == b2._views[0][0].data_ptr()  # moved into the bucket
    with pytest.raises(RuntimeError, match="incomplete"):
        b2.finish()
    # a backward that knows of the bucket writes into the parameter's slice and hands autograd a fresh tensor on it
    from asr_amd.parallel import adopt, grad_slot

    class _Direct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x)
            ctx.w = w
            return x @ w.t()

        @staticmethod
        def backward(ctx, gy):
            (x,) = ctx.saved_tensors
            slot = grad_slot(ctx.w)
            if slot is None:
                return None, gy.t() @ x
            torch.mm(gy.t(), x, out=slot)
            return None, adopt(slot)

    lin = torch.nn.Linear(4, 3, bias=False)
    b3 = GradBuckets([lin], overlap=False)
    x = torch.randn(5, 4)
    for rep in range(2):
        b3.zero_grad()
        b3.buckets[0].fill_(7.0)  # stale contents of the last step
        with b3.accumulate():
            _Direct.apply(x, lin.weight).sum().backward()
            assert lin.weight.grad.data_ptr() == b3._views[0][0].data_ptr()
            assert grad_slot(lin.weight) is None  # taken for this step
            _Direct.apply(x, lin.weight).sum().backward()  # second gradient of the step: added in place
            b3.finish()
        want = 2 * torch.ones(3, 5) @ x
        assert torch.allclose(b3._views[0][0], want, atol=1e-6)
    # the parameter used twice in ONE backward: autograd sums both gradients before it touches .grad, so only the
    # first use may write into the slice
    b3.zero_grad()
    (_Direct.apply(x, lin.weight).sum() + _Direct.apply(3 * x, lin.weight).sum()).backward()
    b3.finish()
    assert torch.allclose(b3._views[0][0], 4 * torch.ones(3, 5) @ x, atol=1e-5)
    assert lin.weight.grad.data_ptr() == b3._views[0][0].data_ptr()
    b3.remove()
    assert not hasattr(lin.weight, "_grad_view")


def test_bucket_plan():
    from asr_amd import LMUFFT
    from asr_amd.parallel import GradBuckets

    layers = [LMUFFT(12, 10, 6, 16, 8) for _ in range(4)]
    per_layer = sum(-(-p.numel() // 64) * 64 for p in layers[0].parameters()) * 4 / (1 << 20)
    assert len(GradBuckets(layers).buckets) == 4
    assert len(GradBuckets(layers, bucket_mb=float("inf")).buckets) == 1
    two = GradBuckets(layers, bucket_mb=2.1 * per_layer)
    assert len(two.buckets) == 2 and all(len(s) == 2 for s in two._sinks)
    # every layer keeps its own sink inside the shared bucket, and the views do not overlap
    ptrs = sorted(v.data_ptr() for vs in two._views for v in vs)
    assert len(set(ptrs)) == len(ptrs) == 16
    two.zero_grad()
    for l in layers[:2]:
        _emulate_layer_backward(l, 1.0)
    assert two._launched == [True, False]
