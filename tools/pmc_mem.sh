#!/bin/bash
# L1 / L2 counters of the bf16 step (4 TCP + 4 TCC counters per run), summary under gpurun_out/pmc_mem/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_mem
rm -rf $O && mkdir -p $O
timeout -s KILL 180 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum --output-format csv -d $O/p1 -o p -- python3 bench.py --steps 3 --warmup 1 --no-graph > $O/p1.log 2>&1
python3 tools/pmc_summary.py $O/p1 > $O/p1.txt
