"""Where the whole model's train step (bench.py --workload c2_full) spends its time: torch profiler, GPU time by
kernel, against the wall clock of the eager step."""
import sys, time, torch
sys.argv = ["bench.py", "--workload", "c2_full"]
import bench
from torch.profiler import profile, ProfilerActivity

args = bench.parse_args()
args.steps, args.warmup = 10, 5
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
# reuse the bench's builder by running it once for timing, then profile a few steps through its closure
import types
src = open(bench.__file__).read()
out = bench.bench_full_step(args, dev, 0, 1)
print("wall ms/step", round(out["ms_per_step"], 2), "hip kernels ms/step", round(out["hip_kernel_ms_per_step"], 2))
args.steps, args.warmup = 3, 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.bench_full_step(args, dev, 0, 1)
rows = prof.key_averages()
tot = sum(r.self_device_time_total for r in rows)
print("GPU time total over the profiled call (ms):", round(tot / 1e3, 2))
for r in sorted(rows, key=lambda r: -r.self_device_time_total)[:28]:
    print(f"{r.self_device_time_total / 1e3:9.2f} ms  x{r.count:5d}  {r.key[:110]}")
