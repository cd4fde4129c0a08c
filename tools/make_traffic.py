"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv files -> profiles/traffic.json.

Per kernel (bench.py's names) and per launch: HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at
half their bytes (128-byte requests tallied as 64), hence the factor 2.  Each counter comes from
its own run.  usage: make_traffic.py [--out file.json] <dir with *_counter_collection.csv> <dtype tag> [more dirs/tags]"""
import collections
import csv
import glob
import json
import os
import sys


def classify(name, seq):
    n = name
    if "k_dgrad_glds_bf16" in n or "k_dgrad_bf16" in n:
        # memory columns: 64 x 192 tiles; input columns: 256 x 192 (LDS-DMA form) or 128 x 128 (fallback)
        return "k_dgrad_m_bf16" if ("ILi64ELi192E" in n or "<64, 192," in n) else "k_dgrad_x_bf16"
    if "k_dgrad" in n:  # fp32: one kernel, launched for gm then for gx
        seq["dgrad"] += 1
        return "k_dgrad_m" if seq["dgrad"] % 2 == 1 else "k_dgrad_x"
    table = [("k_conv_proj_fwd_bf16", "k_conv_proj_fwd_bf16"), ("k_gu_q_bf16", "k_gu_q_bf16"),
             ("k_relu_bwd_q_bf16", "k_relu_bwd_bf16"), ("k_qprep_bf16", "k_qprep_bf16"), ("k_proj_fwd_glds_bf16", "k_proj_fwd_bf16"), ("k_proj_fwd_bf16", "k_proj_fwd_bf16"),
             ("k_conv_bwd2_bf16", "k_conv_bwd_bf16"), ("k_conv_bwd_bf16", "k_conv_bwd_bf16"),
             ("k_conv_fwd_bf16", "k_conv_fwd_bf16"), ("k_wgrad_pair_bf16", "k_wgrad_bf16"), ("k_wgrad_bf16", "k_wgrad_bf16"),
             ("k_relu_bwd_bf16", "k_relu_bwd_bf16"), ("k_encode_fwd_bf16", "k_encode_fwd_bf16"),
             ("k_encode_bwd_bf16", "k_encode_bwd_bf16"), ("k_prep_weights", "k_prep_weights"),
             ("k_reduce_slabs_tall", "k_reduce_slabs_tall"), ("k_reduce_slabs", "k_reduce_slabs"),
             ("k_sum_slabs_masked", "k_sum_slabs_masked"), ("k_gu_reduce", "k_gu_reduce"),
             ("k_conv_fwd_fft", "k_conv_fwd_fft"), ("k_conv_bwd_fft", "k_conv_bwd_fft"),
             ("k_fft_u", "k_fft_u"), ("k_fft_gu", "k_fft_gu"),
             ("k_conv_fwd", "k_conv_fwd"), ("k_conv_bwd", "k_conv_bwd"), ("k_proj_fwd", "k_proj_fwd"),
             ("k_wgrad", "k_wgrad"), ("k_encode_fwd", "k_encode_fwd"), ("k_encode_bwd", "k_encode_bwd"),
             ("k_mel_power", "k_mel_power"), ("k_mel_norm", "k_mel_norm")]
    for sub, out in table:
        if sub in n:
            return out
    return None


def collect(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        seq = collections.Counter()
        for r in rows:
            k = classify(r["Kernel_Name"], seq)
            if k:
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


out = {}
meta = {"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch; one rocprofv3 --pmc run per counter",
        "kernels": {}}
args = sys.argv[1:]
out_path = os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic.json")
if args and args[0] == "--out":
    out_path = args[1]
    args = args[2:]
for d, tag in zip(args[0::2], args[1::2]):
    for k, c in collect(d).items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
            w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            if k in out and tag != "bf16":
                continue  # kernels shared by both paths: keep the bf16 run's figure
            out[k] = (2 * f + w) * 1024
            meta["kernels"][k] = {"run": tag, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                                  "launches": len(c["FETCH_SIZE"])}
out["_meta"] = meta
json.dump(out, open(out_path, "w"), indent=1)
for k, v in out.items():
    if k != "_meta":
        print(f"{k:24s} {v / 1e6:9.2f} MB/launch  {meta['kernels'][k]}")
