"""bf16 forward: convolution and projection as one kernel (conv_algo "auto", where its tile shapes fit) against the
two separate kernels (conv_algo "toeplitz").  Both run the same products in the same order, so u, m, h, h_n and the
relu bit mask must agree bit for bit; the oracle parity of either is held by tests/test_gpu_bf16.py, which runs
whatever "auto" picks.  Shapes: whole and ragged frame tiles (T below, at and above multiples of 128, the maximum
512), every supported memory size (64, 128, 192), hidden sizes below and at the tile's 384 columns, one utterance
and many; and shapes the fused kernel does not take (memory 96, hidden 512, 600 frames) still run."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    (64, 400, 384, 192, 384),
    (96, 400, 384, 192, 384),
    (1, 1, 64, 64, 128),
    (3, 127, 80, 64, 128),
    (2, 128, 96, 128, 256),
    (5, 129, 64, 192, 384),
    (2, 512, 128, 192, 320),
    (7, 257, 384, 128, 384),
    (2, 300, 64, 96, 128),    # memory size not a multiple of 64: separate kernels either way
    (2, 200, 64, 64, 512),    # hidden size above the tile
    (1, 600, 64, 64, 128),    # sequence above the fused kernel's limit
    (400, 100, 64, 64, 128),  # more workgroups than 1.5 per CU: separate kernels
]


@pytest.mark.parametrize("B,T,D,M,Hd", SHAPES)
def test_fused_forward_equals_separate_kernels(B, T, D, M, Hd):
    from asr_amd import LMUFFT

    dev = torch.device("cuda:0")
    torch.manual_seed(T + M)
    outs = {}
    g = torch.Generator(device=dev).manual_seed(3)
    x0 = torch.randn(B, T, D, device=dev, generator=g).bfloat16()
    state = None
    for algo in ("auto", "toeplitz"):
        layer = LMUFFT(D, Hd, M, T, 50.0, conv_algo=algo)
        if state is None:
            torch.nn.init.xavier_uniform_(layer.W_u.weight)
            torch.nn.init.xavier_uniform_(layer.W_h.weight)
            torch.nn.init.normal_(layer.W_h.bias, std=0.1)
            state = layer.state_dict()
        else:
            layer.load_state_dict(state)
        layer = layer.to(dev)
        x = x0.clone().requires_grad_(True)
        h, h_n = layer(x)
        saved = h.grad_fn.saved_tensors
        u, m, hmask = saved[4], saved[5], saved[-1]
        outs[algo] = (h.detach().clone(), h_n.detach().clone(), u.clone(), m.clone(), None if hmask is None else hmask.clone())
        # and the backward runs from what the forward left
        (h.float().square().mean() + h_n.float().sum()).backward()
        outs[algo] += (x.grad.clone(),)
    for a, b, name in zip(outs["auto"], outs["toeplitz"], ("h", "h_n", "u", "m", "hmask", "gx")):
        if a is None or b is None:
            assert a is None and b is None, name
            continue
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))
    h = outs["auto"][0].float()
    assert torch.isfinite(h).all() and float(h.abs().max()) > 0
    assert torch.equal(outs["auto"][1], outs["auto"][0][:, -1, :])
