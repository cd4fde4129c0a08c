"""The data-parallel configuration's per-GPU workload: batch 96 x 400 frames (total batch / world size,
scripts/train_flexible.py:198-203), bf16.  At 38,400 rows the row tiles of the GEMM kernels no longer
divide into whole rounds of workgroups (300 tiles of 128 rows, 192-row ranges ...), so the tile tails
differ from the batch-64 shape the other tests run: three stacked LMUFFT layers forward + backward, one
batch item of every layer against the oracle on that layer's own inputs, the structural properties of the
full-size test, and one train step of the whole model with every op on this package's kernels."""
import numpy as np
import pytest
import torch

from oracle import lmufft_oracle as orc

pytestmark = pytest.mark.gpu

B, T, D, M, HD, THETA = 96, 400, 384, 192, 384, 1000.0


def _np(t):
    return t.detach().float().cpu().numpy()


def test_c3_three_layers_bf16_batch96():
    from asr_amd import LMUFFT
    from tests.test_gpu_bf16 import _bf, _close, _termnorms

    dev = torch.device("cuda:0")
    torch.manual_seed(1234 + 3)
    layers = []
    for _ in range(3):
        l = LMUFFT(D, HD, M, T, THETA)
        torch.nn.init.xavier_uniform_(l.W_u.weight)
        torch.nn.init.xavier_uniform_(l.W_h.weight)
        torch.nn.init.zeros_(l.W_u.bias)
        torch.nn.init.zeros_(l.W_h.bias)
        layers.append(l.to(dev))
    g = torch.Generator(device=dev).manual_seed(1234 + 3)
    x = torch.randn(B, T, D, device=dev, generator=g).bfloat16().requires_grad_(True)
    gy = torch.randn(B, T, HD, device=dev, generator=g).bfloat16()
    ins, outs = [], []
    y = x
    for i, l in enumerate(layers):
        xin = y if i == 0 else y.detach().requires_grad_(True)
        y, h_n = l(xin)
        assert torch.equal(h_n, y[:, -1, :])
        ins.append(xin)
        outs.append(y)
    outs[2].backward(gy)
    outs[1].backward(ins[2].grad)
    outs[0].backward(ins[1].grad)
    gouts = [ins[1].grad, ins[2].grad, gy]
    # every layer on its own inputs: the last batch item (the tail of the last row tile) and one from the middle
    for li, l in enumerate(layers):
        sd = {k: _np(v) for k, v in l.state_dict().items() if k != "fft_H"}
        for b in (95, 50):
            xb, ghb = _np(ins[li][b:b + 1]), _np(gouts[li][b:b + 1])
            ref = orc.lmufft_backward(xb, sd["W_u.weight"], sd["W_u.bias"], _bf(sd["W_h.weight"]), sd["W_h.bias"],
                                      _bf(sd["H"]), ghb, h_mask=_np(outs[li][b:b + 1]) > 0)
            tn = _termnorms(xb, sd["W_u.weight"], sd["W_h.weight"], sd["H"], ref)
            _close(_np(outs[li][b:b + 1]), ref["h"], f"h layer{li}[{b}]", tn["h"])
            _close(_np(ins[li].grad[b:b + 1]), ref["gx"], f"gx layer{li}[{b}]", tn["gx"])
    # parameter gradients are sums over the whole batch: gb_h = column sums of gz exactly (fp32 sums of bf16
    # values, fixed order), gW_h against an fp64 evaluation from the kernel's own tensors, on the last layer
    l = layers[2]
    gz = (gy.float() * (outs[2] > 0)).double()
    gb = gz.sum(dim=(0, 1))
    assert torch.allclose(l.W_h.bias.grad.double(), gb, rtol=1e-4, atol=1e-4 * float(gb.abs().max()))
    xin = ins[2].detach().double()
    gWx = torch.einsum("btn,btd->nd", gz, xin)
    got = l.W_h.weight.grad[:, M:].double()
    scale = torch.sqrt(torch.einsum("btn,btd->nd", gz ** 2, xin ** 2))
    assert bool(((got - gWx).abs() <= 2e-2 * (gWx.abs() + 2 * scale)).all())
    # batch items are independent (a permutation of the batch permutes the outputs bit for bit) and causal
    perm = torch.randperm(B, device=dev, generator=g)
    h_p, _ = layers[0](x.detach()[perm])
    assert torch.equal(h_p, outs[0].detach()[perm])
    x2 = x.detach().clone()
    x2[:, 300:] = 0
    h_c, _ = layers[0](x2)
    assert torch.equal(h_c[:, :300], outs[0].detach()[:, :300])


def test_c3_whole_model_train_step_batch96_runs_on_hip_kernels():
    """One train step of the 3-layer 384/192 model at batch 96 under bf16 autocast inside strict_kernels():
    every op of the forward takes a HIP kernel of this package (no torch formulation is counted), the loss and
    the gradient norm are finite, and the step changes the parameters."""
    from asr_amd import FlatAdamW, harness
    from asr_amd.harness import LMUASRModel
    from asr_amd.parallel import GradBuckets

    dev = torch.device("cuda:0")
    torch.manual_seed(1234 + 3)
    model = LMUASRModel(80, HD, M, 3, THETA, T, 34, num_attention_heads=6, dropout=0.1).to(dev).train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(mod.weight)
            torch.nn.init.zeros_(mod.bias)
    buckets = GradBuckets(model.bucket_groups(), bucket_mb=4.0)
    opt = FlatAdamW(buckets=buckets, lr=1.5e-3, weight_decay=0.01, max_grad_norm=10.0)
    g = torch.Generator(device=dev).manual_seed(7)
    spec = torch.randn(B, T, 80, device=dev, generator=g)
    in_len = torch.randint(200, T + 1, (B,), device=dev, generator=g)
    in_len[0] = T
    tgt_len = torch.randint(20, 61, (B,), device=dev, generator=g)
    targets = torch.randint(0, 33, (B, 60), device=dev, generator=g)
    before = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    harness.PATHS.clear()
    buckets.zero_grad()
    with harness.strict_kernels(), torch.autocast("cuda", dtype=torch.bfloat16):
        lp = model.log_probs_for_training(spec, in_len)
    assert not [k for k in harness.PATHS if k.startswith("torch:")], harness.PATHS
    loss = model.compute_loss(lp, targets, in_len, tgt_len)
    loss.backward()
    buckets.finish()
    assert torch.isfinite(loss) and torch.isfinite(opt.grad_norm()) and float(opt.grad_norm()) > 0
    opt.step()
    after = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert bool(torch.isfinite(after).all()) and not torch.equal(before, after)
