#!/bin/bash
# Runs on the GPU box (from the repo root): bench lines, rocprofv3 kernel statistics and PMC traffic
# for both dtypes, written under gpurun_out/refresh/.  Every GPU step has its own time limit.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
timeout -k 10 240 python3 bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout -k 10 240 python3 bench.py --dtype f32 > $O/bench_f32.json 2> $O/bench_f32.err
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o bf16 -- python3 bench.py --steps 30 --no-extras --no-cpu-baseline > $O/stats_bf16.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -o f32 -- python3 bench.py --dtype f32 --steps 30 --no-extras --no-cpu-baseline > $O/stats_f32.log 2>&1
for dt in bf16 f32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$dt/$c -o p -- python3 bench.py --dtype $dt --steps 5 --warmup 1 --no-graph --no-extras --no-cpu-baseline > $O/pmc_${dt}_$c.log 2>&1
  done
done
export PYTHONPATH=$PWD
# conv-only unit at batch 64 x 400 frames: kernel statistics and HBM traffic counters of every route
for dt in bf16 f32; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_conv_$dt -o conv -- python3 bench.py --unit conv --dtype $dt --batch 64 --frames 400 --steps 200 > $O/stats_conv_$dt.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d $O/pmc_conv_$dt/$c -o p -- python3 bench.py --unit conv --dtype $dt --batch 64 --frames 400 --steps 20 --warmup 2 > $O/pmc_conv_${dt}_$c.log 2>&1
  done
done
for shape in "64 400" "16 400" "32 3000"; do
  set -- $shape
  for dt in bf16 f32; do
    timeout -k 10 240 python3 bench.py --unit conv --dtype $dt --batch $1 --frames $2 >> $O/bench_conv.jsonl 2>> $O/bench_conv.err
  done
done
# the long-utterance configuration (batch 32, 3000 frames, one layer) on both bf16 convolution routes, graph replay
for c in scan toeplitz; do
  timeout -k 10 240 python3 bench.py --batch 32 --frames 3000 --layers 1 --conv $c --steps 100 --no-cpu-baseline > $O/bench_c4_bf16_$c.json 2>> $O/bench_c4.err
done
timeout -k 10 240 python3 bench.py --batch 32 --frames 3000 --layers 1 --dtype f32 --steps 50 --no-cpu-baseline > $O/bench_c4_f32_auto.json 2>> $O/bench_c4.err
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4_scan -o c4 -- python3 bench.py --batch 32 --frames 3000 --layers 1 --conv scan --steps 30 --no-cpu-baseline > $O/stats_c4_scan.log 2>&1
for T in 1024 1536 2048; do
  timeout -k 10 240 python3 bench.py --unit conv --dtype bf16 --batch 32 --frames $T >> $O/bench_conv_sweep.jsonl 2>> $O/bench_conv.err
done
timeout -k 10 240 python3 tools/mel_bench.py > $O/mel_bench.txt 2>&1
timeout -k 10 240 python3 tools/infer_bench.py > $O/infer_bench.json 2> $O/infer_bench.err
timeout -k 10 240 python3 tools/t3000.py > $O/t3000.txt 2>&1
timeout -k 10 300 python3 bench.py --workload c2_full --steps 50 --warmup 5 --torch-profile > $O/bench_c2_full.json 2> $O/bench_c2_full_kernels.txt
timeout -k 10 300 python3 bench.py --workload c2_full --steps 50 --warmup 5 --batch 96 > $O/bench_c3_batch96.json 2> $O/bench_c3_batch96.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2_full -o c2 -- python3 bench.py --workload c2_full --steps 30 --warmup 5 > $O/stats_c2_full.log 2>&1
timeout -k 10 240 python3 tools/attn_train_bench.py 0.1 > $O/attn_train_bench.txt 2>&1
echo done > $O/DONE
