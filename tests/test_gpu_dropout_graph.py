"""Dropout inside a training step replayed from a HIP graph: the seeds are frozen at capture, the masks are
not -- the library adds a device-side epoch counter to the seeds when the kernels run, and the captured step
increments it (asr_amd/dropout.py, C ABI lmufft_dropout_epoch).  The reference draws a fresh mask per call
(nn.Dropout, src/models/lmu_encoder.py:45, 86, 232)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_replays_of_a_captured_dropout_step_draw_fresh_masks():
    from asr_amd.attention import masked_attention_train_packed
    from asr_amd.dropout import dropout_epoch
    from asr_amd.layernorm import add_layer_norm_train
    from asr_amd.parallel import GradBuckets

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, T, C, heads = 3, 96, 128, 2
    ln_w = torch.nn.Parameter(torch.ones(C, device=dev))
    ln_b = torch.nn.Parameter(torch.zeros(C, device=dev))
    qkv = torch.randn(B, T, 3 * C, device=dev).bfloat16().requires_grad_(True)
    res = torch.randn(B, T, C, device=dev)
    lengths = torch.tensor([96, 70, 33], device=dev, dtype=torch.int32)
    gy = torch.randn(B, T, C, device=dev)
    buckets = GradBuckets([[ln_w, ln_b]], overlap=False)
    out = {}

    def fn():
        qkv.grad = None
        ctx = masked_attention_train_packed(qkv, heads, lengths, p_drop=0.25)
        y = add_layer_norm_train(ctx, res, ln_w, ln_b, p_drop=0.25)
        y.backward(gy)
        out["y"], out["ctx"] = y.detach(), ctx.detach()

    for _ in range(2):  # eager warm-up (each call advances the Python-side call counter: different masks)
        buckets.zero_grad()
        fn()
        buckets.finish()
    out.clear()
    step = buckets.capture_graph_exchange([fn], overlap=False)
    epoch = dropout_epoch(dev)
    seen = []
    for _ in range(3):
        before = int(epoch.item())
        step()
        torch.cuda.synchronize()
        assert int(epoch.item()) == before + 1
        seen.append((out["y"].clone(), out["ctx"].clone(), qkv.grad.clone(), buckets.arena.clone()))
    # fresh masks at every replay: the dropped positions of the layer-norm output (exact zeros) differ
    z = [(y == 0) for y, _, _, _ in seen]
    assert z[0].float().mean() > 0.15 and not torch.equal(z[0], z[1]) and not torch.equal(z[1], z[2])
    assert not torch.equal(seen[0][1], seen[1][1])  # attention weights dropped differently as well
    # and one replay is self-consistent: setting the counter back reproduces that replay's forward AND backward
    epoch.fill_(int(epoch.item()) - 1)
    step()
    torch.cuda.synchronize()
    for a, b in zip(seen[2], (out["y"], out["ctx"], qkv.grad, buckets.arena)):
        assert torch.equal(a, b)
    buckets.remove()


def test_capturing_dropout_without_an_epoch_counter_raises():
    import asr_amd.dropout as dropout
    from asr_amd.layernorm import add_layer_norm_train

    dev = torch.device("cuda:0")
    x = torch.randn(4, 64, device=dev, requires_grad=True)
    w = torch.ones(64, device=dev, requires_grad=True)
    b = torch.zeros(64, device=dev, requires_grad=True)
    add_layer_norm_train(x, None, w, b, p_drop=0.1)  # eager: fine
    saved = dict(dropout._EPOCHS)
    dropout._EPOCHS.clear()
    try:
        g = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="epoch"):
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                add_layer_norm_train(x, None, w, b, p_drop=0.1)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):  # explicit seeds: the caller's business
            add_layer_norm_train(x, None, w, b, p_drop=0.1, seeds=(1, 2))
    finally:
        dropout._EPOCHS.update(saved)
