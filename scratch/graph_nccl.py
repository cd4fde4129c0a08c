"""HIP-graph capture of the step with an RCCL process group alive (world size 1 on the one GPU of a test box):
the watchdog thread and the communicator must not disturb the capture; an all-reduce on the flat buckets after the
replay must work."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from asr_amd import LMUFFT
from asr_amd.parallel import GradBuckets
torch.manual_seed(0)
layers = [LMUFFT(384, 384, 192, 400, 1000).to(dev) for _ in range(3)]
buckets = GradBuckets(list(reversed(layers)))
x = torch.randn(64, 400, 384, device=dev).bfloat16().requires_grad_(True)
gy = torch.randn(64, 400, 384, device=dev).bfloat16()
def fwd_bwd():
    y = x
    for l in layers: y, _ = l(y)
    y.backward(gy)
for _ in range(3):
    buckets.zero_grad(); x.grad = None; fwd_bwd(); buckets.finish()
t = torch.ones(4, device=dev); dist.all_reduce(t)  # communicator warm
ref = [b.clone() for b in buckets.buckets]
x.grad = None
step = buckets.capture_step(fwd_bwd)
for _ in range(5): step()
for b in buckets.buckets: dist.all_reduce(b)  # what finish() does when world > 1
torch.cuda.synchronize()
for a, b in zip(ref, buckets.buckets):
    assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * float(a.abs().max())), float((a - b).abs().max())
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("graph + process group OK,", (time.perf_counter() - t0) / 50 * 1e3, "ms/step")
dist.destroy_process_group()
