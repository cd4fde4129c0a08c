"""Copies what tools/refresh_profiles.sh left under gpurun_out/refresh/ into profiles/<round>/ under the
names profiles/README.md lists, and regenerates profiles/traffic.json.  usage: collect_profiles.py r02"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")


def main():
    rnd = sys.argv[1]
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)

    def cp(a, b):
        p = os.path.join(SRC, a)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copyfile(p, os.path.join(dst, b))
        else:
            print("missing:", a)

    for n in ("bench_bf16.json", "bench_f32.json", "bench_c2_full.json", "bench_c2_full_kernels.txt",
              "bench_c3_batch96.json", "bench_c4_bf16_scan.json", "bench_c4_bf16_toeplitz.json",
              "bench_c4_f32_auto.json", "bench_conv_sweep.jsonl"):
        cp(n, n)
    cp("stats_bf16/bf16_kernel_stats.csv", "bf16_kernel_stats.csv")
    cp("stats_f32/f32_kernel_stats.csv", "f32_kernel_stats.csv")
    cp("stats_c2_full/c2_kernel_stats.csv", "c2_full_kernel_stats.csv")
    cp("stats_c4_scan/c4_kernel_stats.csv", "c4_bf16_scan_kernel_stats.csv")
    cp("stats_conv_bf16/conv_kernel_stats.csv", "conv_bf16_b64_t400_kernel_stats.csv")
    cp("stats_conv_f32/conv_kernel_stats.csv", "conv_f32_b64_t400_kernel_stats.csv")
    # one file per conv-only line
    p = os.path.join(SRC, "bench_conv.jsonl")
    if os.path.exists(p):
        for line in open(p):
            if not line.startswith("{"):
                continue
            d = json.loads(line)
            w = d["config"]["workload"]  # "conv-only unit: u[B,T] * H[..."
            B, T = w.split("u[")[1].split("]")[0].split(",")
            with open(os.path.join(dst, f"bench_conv_{d['dtype']}_b{B}_t{T}.json"), "w") as f:
                f.write(line)
    with open(os.path.join(dst, "infer_bench.txt"), "w") as f:
        for n in ("infer_bench.json", "infer_bench.err"):
            q = os.path.join(SRC, n)
            if os.path.exists(q):
                f.write(open(q).read())
    with open(os.path.join(dst, "microbench.txt"), "w") as f:
        for n, title in (("mel_bench.txt", "mel frontend (tools/mel_bench.py)"),
                         ("t3000.txt", "3000 frames, eager launches (tools/t3000.py)"),
                         ("attn_train_bench.txt", "training attention (tools/attn_train_bench.py 0.1)")):
            q = os.path.join(SRC, n)
            if os.path.exists(q):
                f.write(f"== {title}\n" + open(q).read() + "\n")
    args = [sys.executable, os.path.join(ROOT, "tools", "make_traffic.py")]
    for dt in ("bf16", "f32"):
        args += [os.path.join(SRC, f"pmc_{dt}"), dt]
    subprocess.run(args, check=False)
    # the conv-only unit's kernels (both routes of each dtype), per launch, in a file of their own
    args = [sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), "--out", os.path.join(dst, "conv_unit_traffic.json")]
    for dt in ("bf16", "f32"):
        args += [os.path.join(SRC, f"pmc_conv_{dt}"), dt]
    subprocess.run(args, check=False)


if __name__ == "__main__":
    main()
