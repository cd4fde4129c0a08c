#!/bin/bash
# SQ counters of the bf16 step in two passes (at most 8 SQ counters per run), summaries under gpurun_out/pmc_sq/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_sq
rm -rf $O && mkdir -p $O
timeout -s KILL 180 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -o p -- python3 bench.py --steps 3 --warmup 1 --no-graph > $O/p1.log 2>&1
timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/p2 -o p -- python3 bench.py --steps 3 --warmup 1 --no-graph > $O/p2.log 2>&1
python3 tools/pmc_summary.py $O/p1 > $O/p1.txt
python3 tools/pmc_summary.py $O/p2 > $O/p2.txt
