#!/bin/bash
# ablation of the projection kernel: LMUFFT_DBG bits 1=no MFMA, 2=no epilogue, 4=no A traffic, 8=no W traffic
for d in 0 1 2 3 4 8 12 7 11 15; do
  LMUFFT_DBG=$d timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/ab_$d.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("/tmp/ab_$d.json"))
print("dbg=$d proj_us=%.1f" % d["kernels"]["k_proj_fwd_bf16"]["avg_us"])
PY
done
