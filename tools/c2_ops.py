"""Which torch operators still launch kernels in the whole-model training step (eager, torch profiler with shapes)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from torch.profiler import ProfilerActivity, profile
from asr_amd import FlatAdamW, LearningRateScheduler
from asr_amd.harness import LMUASRModel
from asr_amd.parallel import GradBuckets

dev = torch.device("cuda:0")
B, T = 64, 400
torch.manual_seed(1234)
model = LMUASRModel(80, 384, 192, 3, 1000, T, 34, num_attention_heads=6, dropout=0.1).to(dev).train()
buckets = GradBuckets(model.bucket_groups(), bucket_mb=4.0, overlap=False)
opt = FlatAdamW(buckets=buckets, lr=1.5e-3, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=10.0, device_clock=True)
sched = LearningRateScheduler(opt, warmup_steps=1500, max_lr=1.5e-3)
spec = torch.randn(B, T, 80, device=dev)
in_len = torch.full((B,), T, dtype=torch.long, device=dev)
tgt_len = torch.randint(20, 61, (B,), device=dev).to(torch.int32)
targets = torch.randint(0, 33, (B, 60), device=dev)
in_len32 = in_len.to(torch.int32)


def step():
    buckets.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lp = model.log_probs_for_training(spec, in_len)
    loss = model.compute_loss(lp, targets, in_len32, tgt_len)
    loss.backward()
    buckets.finish()
    opt.step()
    sched.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [r for r in prof.key_averages(group_by_input_shape=True) if r.key.startswith("aten::") and r.self_device_time_total > 0]
rows.sort(key=lambda r: -r.self_device_time_total)
for r in rows:
    print(f"{r.self_device_time_total:8.1f} us x{r.count:3d}  {r.key:28s} {str(r.input_shapes)[:150]}")
