import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "dtype")})
print("roofline", d["roofline"])
print("step_hbm", d["step_hbm"])
print("cpu", d.get("cpu_baseline"))
tot = 0
for k, v in d["kernels"].items():
    tot += v["total_ms_per_step"]
    print(f"{k:22s} {v['avg_us']:8.1f} us x{v['launches_per_step']:.0f} = {v['total_ms_per_step']*1e3:8.1f} us/step")
print("sum kernels (ms/step)", tot)
