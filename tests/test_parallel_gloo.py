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


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from asr_amd import LMUFFT
        from asr_amd.parallel import GradBuckets

        torch.manual_seed(0)  # same parameters on every rank
        layers = [LMUFFT(12, 10, 6, 16, 8) for _ in range(2)]
        buckets = GradBuckets([list(l.parameters()) for l in reversed(layers)])
        assert len(buckets.buckets) == 2
        n_params = sum(p.numel() for l in layers for p in l.parameters())
        assert sum(b.numel() for b in buckets.buckets) == n_params  # buffers are not in the buckets
        for step in range(2):
            buckets.zero_grad()
            # emulate autograd: accumulate a rank-dependent gradient into every parameter, last layer first
            for l in reversed(layers):
                for p in l.parameters():
                    g = torch.full_like(p, float(rank + 1 + step))
                    # AccumulateGrad adds in place into the existing .grad (a view of the flat bucket)
                    p.grad.add_(g)
                    for hook in p._post_accumulate_grad_hooks.values():
                        hook(p)
            buckets.finish()
            want = sum(r + 1 + step for r in range(world)) / world
            for l in layers:
                for p in l.parameters():
                    assert torch.allclose(p.grad, torch.full_like(p, want)), (rank, step)
                    assert p.grad.data_ptr() >= min(b.data_ptr() for b in buckets.buckets)
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


def test_grad_buckets_single_process_is_noop():
    from asr_amd import LMUFFT
    from asr_amd.parallel import GradBuckets

    layer = LMUFFT(12, 10, 6, 16, 8)
    b = GradBuckets([list(layer.parameters())])
    b.zero_grad()
    for p in layer.parameters():
        p.grad.add_(1.0)
    b.finish()
    assert all(torch.equal(p.grad, torch.ones_like(p)) for p in layer.parameters())
