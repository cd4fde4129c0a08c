"""GPU parity tests of the flat optimizer step (lmufft_grad_norm2_f32, lmufft_adamw_step_f32) and of the CTC target
packing (lmufft_ctc_pack_targets), through the C ABI, against oracle/optim_oracle.py (pinned to torch on the CPU by
tests/test_optim_oracle.py).

Bar: fp32 arithmetic against the float64 oracle.  Parameters after k steps: |got - ref| <= 2e-6 * |ref| + 2e-7 * k
(every step adds rounding of order eps32 * lr-sized updates and one fp32 sqrt / division); the norm: rtol 1e-6;
the moments (sums of O(1) terms in fp32): rtol 1e-5 + atol 1e-6.
Packing is integer work: exact."""
import math

import numpy as np
import pytest
import torch

from oracle import optim_oracle as oo

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    return torch.device("cuda:0")


def _make(shapes, seed, dev):
    g = torch.Generator().manual_seed(seed)
    host = [torch.randn(s, generator=g) for s in shapes]
    return host, [torch.nn.Parameter(h.clone().to(dev)) for h in host]


SHAPES = [(384, 576), (384,), (1, 384), (1,), (34, 384), (7, 5, 3)]


@pytest.mark.parametrize("max_norm", [None, 10.0, 0.25])
def test_flat_adamw_matches_oracle(max_norm):
    from asr_amd.optim import FlatAdamW, LearningRateScheduler

    dev = _dev()
    host, params = _make(SHAPES, 11, dev)
    opt = FlatAdamW(params, lr=1.5e-3, weight_decay=0.01, max_grad_norm=max_norm)
    sch = LearningRateScheduler(opt, warmup_steps=3, max_lr=1.5e-3, decay_steps=2, decay_rate=0.9)
    ref = [h.double().numpy().copy() for h in host]
    m = [np.zeros_like(r) for r in ref]
    v = [np.zeros_like(r) for r in ref]
    g = torch.Generator().manual_seed(12)
    lr = 1.5e-3
    steps = 7
    for t in range(1, steps + 1):
        grads = [torch.randn(s, generator=g) * 2 for s in SHAPES]
        opt.zero_grad()
        for p, x in zip(params, grads):
            p.grad.add_(x.to(dev))  # what autograd's accumulation does with a pre-set .grad
        if max_norm is not None:
            want = oo.global_norm([x.double().numpy() for x in grads])
            assert abs(float(opt.grad_norm()) - want) <= 1e-6 * want
        opt.step()
        oo.adamw_step(ref, [x.double().numpy() for x in grads], m, v, t, lr, weight_decay=0.01, max_grad_norm=max_norm)
        lr = sch.step()
        assert lr == oo.scheduled_lr(t, 3, 1.5e-3, 2, 0.9)
    for p, r in zip(params, ref):
        got = p.detach().cpu().double().numpy()
        assert np.all(np.abs(got - r) <= 2e-6 * np.abs(r) + 2e-7 * steps), float(np.abs(got - r).max())
        assert p.data_ptr() >= opt.flat.data_ptr() and p._version > 0
    # moments, through the torch-compatible state dict
    sd = opt.state_dict()
    for i, (mm, vv) in enumerate(zip(m, v)):
        np.testing.assert_allclose(sd["state"][i]["exp_avg"].cpu().double().numpy(), mm, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sd["state"][i]["exp_avg_sq"].cpu().double().numpy(), vv, rtol=1e-5, atol=1e-6)
        assert float(sd["state"][i]["step"]) == steps


def test_nan_gradient_skips_update_and_state_dict_round_trips_through_torch():
    from asr_amd.optim import FlatAdamW

    dev = _dev()
    host, params = _make(SHAPES, 21, dev)
    opt = FlatAdamW(params, lr=1e-3, max_grad_norm=10.0)
    g = torch.Generator().manual_seed(22)
    for p in params:
        p.grad.copy_(torch.randn(p.shape, generator=g).to(dev))
    opt.step()
    before = [p.detach().clone() for p in params]
    mom = opt.exp_avg.clone()
    params[2].grad[0, 5] = float("nan")
    opt.step()
    assert math.isnan(float(opt.grad_norm()))
    for p, b in zip(params, before):
        assert torch.equal(p.detach(), b)
    assert torch.equal(opt.exp_avg, mom)
    # torch.optim.AdamW accepts the state dict, and a second FlatAdamW continues from it identically
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]
    topt = torch.optim.AdamW(twin, lr=1e-3)
    sd = opt.state_dict()
    for st in sd["state"].values():
        st["step"] = torch.tensor(1.0)  # the skipped step does not count for the comparison below
    topt.load_state_dict(sd)
    other = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt2 = FlatAdamW(other, lr=1e-3)
    opt2.load_state_dict(sd)
    assert opt2.steps == 1
    for p, q, r in zip(twin, other, params):
        x = torch.randn(p.shape, generator=g).to(dev)
        p.grad = x.clone()
        q.grad.copy_(x)
    topt.step()
    opt2.step()
    for p, q in zip(twin, other):
        assert torch.allclose(p.detach(), q.detach(), rtol=2e-6, atol=2e-7)


def test_device_clock_follows_the_reference_loop_and_replays_from_a_graph():
    """FlatAdamW(device_clock=True): learning rate and step count live on the device (lmufft_adamw_step_clock_f32).
    Against the float64 oracle run the way the reference's loop runs it (src/training/distributed.py:207-234:
    optimizer.step() with the rate the scheduler set after the previous step, no optimizer.step() on a NaN norm,
    lr_scheduler.step() every iteration); bit-equal to the host-clocked optimizer while no step is skipped; and the
    same sequence again as replays of ONE captured step -- nothing of the clock is frozen in the graph."""
    from asr_amd.optim import FlatAdamW, LearningRateScheduler

    dev = _dev()
    host, params = _make(SHAPES, 31, dev)
    _, params_h = _make(SHAPES, 31, dev)
    _, params_g = _make(SHAPES, 31, dev)
    kw = dict(lr=1.5e-3, weight_decay=0.01, max_grad_norm=0.5)
    opt = FlatAdamW(params, device_clock=True, **kw)
    sch = LearningRateScheduler(opt, warmup_steps=3, max_lr=2e-3, decay_steps=2, decay_rate=0.9)
    opt_h = FlatAdamW(params_h, **kw)
    sch_h = LearningRateScheduler(opt_h, warmup_steps=3, max_lr=2e-3, decay_steps=2, decay_rate=0.9)
    opt_g = FlatAdamW(params_g, device_clock=True, **kw)
    LearningRateScheduler(opt_g, warmup_steps=3, max_lr=2e-3, decay_steps=2, decay_rate=0.9)
    # the graph: one optimizer step on whatever the gradient arena holds
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        opt_g.step()
    ref = [h.double().numpy().copy() for h in host]
    m = [np.zeros_like(r) for r in ref]
    v = [np.zeros_like(r) for r in ref]
    g = torch.Generator().manual_seed(32)
    lr, t_adam, steps, nan_at = 1.5e-3, 0, 9, 5
    for it in range(1, steps + 1):
        grads = [torch.randn(s, generator=g) * 2 for s in SHAPES]
        if it == nan_at:
            grads[1][7] = float("nan")
        for o in (opt, opt_h, opt_g):
            o.zero_grad()
            for p, x in zip(o.params, grads):
                p.grad.add_(x.to(dev))
        opt.step()
        opt_h.step()
        graph.replay()
        if it != nan_at:
            t_adam += 1
            oo.adamw_step(ref, [x.double().numpy() for x in grads], m, v, t_adam, lr, weight_decay=0.01, max_grad_norm=0.5)
        assert opt.device_lr() == np.float32(lr) == opt_g.device_lr()
        lr = sch.step()
        assert lr == sch_h.step() == oo.scheduled_lr(it, 3, 2e-3, 2, 0.9)
        if it < nan_at:  # the host clock counts the skipped step, so the two part ways there (by design)
            for p, q in zip(params, params_h):
                assert torch.equal(p.detach(), q.detach())
        for p, q in zip(params, params_g):
            assert torch.equal(p.detach(), q.detach())
    assert opt.device_steps() == (steps, steps - 1) == opt_g.device_steps()
    for p, r in zip(params, ref):
        got = p.detach().cpu().double().numpy()
        assert np.all(np.abs(got - r) <= 2e-6 * np.abs(r) + 2e-7 * steps), float(np.abs(got - r).max())
    assert float(opt.state_dict()["state"][0]["step"]) == steps - 1


def test_flat_adamw_on_the_bucket_arena_trains_a_layer():
    """FlatAdamW adopts the gradient arena of GradBuckets (the LMUFFT kernels write their parameter gradients
    straight into it): ten steps on a small layer lower the loss, and match torch.optim.AdamW on a twin layer
    that goes through plain autograd gradients."""
    from asr_amd import LMUFFT
    from asr_amd.optim import FlatAdamW
    from asr_amd.parallel import GradBuckets

    dev = _dev()
    torch.manual_seed(3)
    a = LMUFFT(24, 32, 16, 50, 20).to(dev)
    b = LMUFFT(24, 32, 16, 50, 20).to(dev)
    b.load_state_dict(a.state_dict())
    buckets = GradBuckets([a])
    opt = FlatAdamW(buckets=buckets, lr=3e-3, weight_decay=0.01, max_grad_norm=1.0)
    topt = torch.optim.AdamW(b.parameters(), lr=3e-3, weight_decay=0.01)
    x = torch.randn(4, 50, 24, device=dev)
    y = torch.randn(4, 50, 32, device=dev)
    losses = []
    for _ in range(10):
        buckets.zero_grad()
        la = (a(x)[0] - y).square().mean()
        la.backward()
        buckets.finish()
        opt.step()
        topt.zero_grad()
        lb = (b(x)[0] - y).square().mean()
        lb.backward()
        torch.nn.utils.clip_grad_norm_(b.parameters(), 1.0)
        topt.step()
        losses.append(float(la.detach()))
        assert abs(losses[-1] - float(lb.detach())) <= 1e-4 * abs(float(lb.detach()))
    assert losses[-1] < losses[0]
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(p.detach(), q.detach(), rtol=1e-4, atol=1e-5), n


def test_errors():
    from asr_amd.optim import FlatAdamW

    _dev()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FlatAdamW([torch.nn.Parameter(torch.zeros(3))])
    with pytest.raises(ValueError):
        FlatAdamW([])


@pytest.mark.parametrize("B,L", [(1, 1), (5, 7), (64, 60), (300, 33)])
def test_pack_targets_matches_loop(B, L):
    from asr_amd.ctc import pack_targets

    dev = _dev()
    rng = np.random.default_rng(B * 100 + L)
    t = rng.integers(0, 33, size=(B, L))
    lengths = rng.integers(0, L + 1, size=B)
    lengths[0] = L
    want = oo.pack_targets(t, lengths)
    packed, offsets = pack_targets(torch.from_numpy(t).to(dev), torch.from_numpy(lengths).to(dev))
    assert packed.dtype == torch.int64 and packed.cpu().numpy().tolist() == want.tolist()
    assert offsets.cpu().numpy().tolist() == [0] + np.cumsum(lengths).tolist()
    # a caller that knows the total avoids the read-back; strided rows and int32 labels are accepted
    wide = torch.from_numpy(np.concatenate([t, t], axis=1)).to(dev).to(torch.int32)[:, :L]
    packed2, _ = pack_targets(wide, torch.from_numpy(lengths).to(dev), total=int(lengths.sum()))
    assert packed2.cpu().numpy().tolist() == want.tolist()


def test_packed_targets_feed_ctc_loss_like_the_reference_loop():
    from asr_amd.ctc import pack_targets

    dev = _dev()
    g = torch.Generator().manual_seed(8)
    B, T, V, L = 6, 50, 34, 12
    lp = torch.randn(T, B, V, generator=g).log_softmax(-1).to(dev)
    targets = torch.randint(0, V - 1, (B, L), generator=g)
    tl = torch.tensor([12, 3, 7, 1, 9, 12])
    il = torch.full((B,), T)
    packed, _ = pack_targets(targets.to(dev), tl.to(dev))
    loss = torch.nn.functional.ctc_loss(lp, packed, il, tl, blank=V - 1, zero_infinity=True)
    ref = torch.nn.functional.ctc_loss(lp.cpu(), torch.from_numpy(oo.pack_targets(targets.numpy(), tl.numpy())), il, tl,
                                       blank=V - 1, zero_infinity=True)
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))


def test_pack_targets_refuses_a_total_that_cannot_be_right():
    from asr_amd import pack_targets

    dev = _dev()
    targets = torch.arange(5 * 7, device=dev).reshape(5, 7) % 30
    lengths = torch.tensor([7, 3, 0, 9, 2])  # 9 is clamped to max_len = 7 by the kernel
    packed, offsets = pack_targets(targets, lengths, total=19)
    assert packed.numel() == 19 and int(offsets[-1]) == 19
    with pytest.raises(RuntimeError, match="sum to 19"):
        pack_targets(targets, lengths, total=21)           # the unclamped sum: a stale total
    with pytest.raises(RuntimeError, match="cannot come from"):
        pack_targets(targets, lengths.to(dev), total=36)   # device lengths: only the bound is checked
