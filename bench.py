#!/usr/bin/env python3
"""Benchmark of the LMUFFT hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = forward + backward through the encoder's stack of three LMUFFT
layers (hidden 384, memory 192, theta 1000, 400 frames: the layer shape the
training-throughput metric is quoted on) on one synthetic batch of 64 utterances per GPU, including the
data-parallel gradient exchange.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}  # dense matrix peaks, spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=384)
    ap.add_argument("--memory", type=int, default=192)
    ap.add_argument("--theta", type=float, default=1000.0)
    ap.add_argument("--conv", default="auto", choices=["auto", "toeplitz", "fft"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"],
                    help="activation / MFMA operand type (fp32 accumulation either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def kernel_work(name, B, T, D, M, Hd, s=4):
    """Algorithmic (flops, HBM bytes) of a kernel's launches in ONE layer's forward+backward.

    Bytes are what the kernel's contract makes it move at least once: its inputs read once, its
    outputs written once (s = bytes per activation element; parameters, u, gu, H are fp32)."""
    bf16 = name.endswith("_bf16")
    name = name.replace("_bf16", "")
    rows = B * T
    MD = M + D
    # the bf16 backward reads the masked upstream gradient gz (written once by k_relu_bwd);
    # the fp32 kernels re-derive it from h and gh in their loaders
    gz_in = Hd * s if bf16 else 2 * Hd * s
    table = {
        "k_proj_fwd": (2.0 * rows * MD * Hd, rows * (MD + Hd) * s + Hd * MD * s),
        "k_dgrad_m": (2.0 * rows * M * Hd, rows * (gz_in + M * s) + Hd * M * s),
        "k_dgrad_x": (2.0 * rows * D * Hd, rows * (gz_in + D * s + 4) + Hd * D * s),
        "k_wgrad": (2.0 * rows * MD * Hd, rows * (gz_in + MD * s) + Hd * MD * 4),
        "k_relu_bwd": (2.0 * rows * Hd, rows * 3 * Hd * s),
        "k_conv_fwd": (1.0 * B * T * (T + 1) * M, rows * M * s + rows * 4 + M * T * s),
        "k_conv_bwd": (1.0 * B * T * (T + 1) * M, rows * M * s + rows * 4 + M * T * s),
        "k_encode_fwd": (2.0 * rows * D, rows * D * s + rows * 4),
        "k_encode_bwd": (2.0 * rows * D, rows * D * s + rows * 4),
    }
    return table.get(name)


def cpu_baseline(args, H, seconds):
    """The reference's own route (torch.fft + autograd) timed on the host cores."""
    from oracle.lmufft_torch_port import PortLMUFFT

    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    Hc = H.cpu()
    layers = [PortLMUFFT(args.hidden, args.hidden, args.memory, Hc) for _ in range(args.layers)]
    for l in layers:
        torch.nn.init.xavier_uniform_(l.W_u.weight)
        torch.nn.init.xavier_uniform_(l.W_h.weight)
        torch.nn.init.zeros_(l.W_u.bias)
        torch.nn.init.zeros_(l.W_h.bias)
    x = torch.randn(args.batch, args.frames, args.hidden, requires_grad=True)
    gy = torch.randn(args.batch, args.frames, args.hidden)

    def step():
        y = x
        for l in layers:
            y, _ = l(y)
        y.backward(gy)
        x.grad = None
        for l in layers:
            l.zero_grad(set_to_none=True)

    step()
    times = []
    t_end = time.perf_counter() + seconds
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 50):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    # one-thread figure on a bounded sample: three steps
    torch.set_num_threads(1)
    step()
    one = []
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        one.append(time.perf_counter() - t0)
    one.sort()
    torch.set_num_threads(threads)
    return {
        "value": args.batch * args.frames / med,
        "unit": "frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"same workload (batch {args.batch} x {args.frames} frames x {args.layers} layers, fp32, "
                  f"torch.fft route + autograd), median of {len(times)} steps, {med * 1e3:.1f} ms/step",
        "one_thread": {"value": args.batch * args.frames / one[1], "ms_per_step": one[1] * 1e3, "steps": 3},
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the LMUFFT layer has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    from asr_amd import LMUFFT, _lib
    from asr_amd.parallel import GradBuckets

    B, T, D, M, Hd = args.batch, args.frames, args.hidden, args.memory, args.hidden
    torch.manual_seed(1234)  # identical parameters on every rank
    layers = []
    for _ in range(args.layers):
        l = LMUFFT(D, Hd, M, T, args.theta, conv_algo=args.conv)
        # the reference model re-initialises every nn.Linear like this (src/models/asr_model.py:271-276)
        torch.nn.init.xavier_uniform_(l.W_u.weight)
        torch.nn.init.xavier_uniform_(l.W_h.weight)
        torch.nn.init.zeros_(l.W_u.bias)
        torch.nn.init.zeros_(l.W_h.bias)
        layers.append(l.to(dev))
    buckets = GradBuckets(list(reversed(layers)))
    g = torch.Generator(device=dev).manual_seed(1234 + 2 + rank)  # each rank its own utterances
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    esize = 2 if args.dtype == "bf16" else 4
    x = torch.randn(B, T, D, device=dev, generator=g).to(tdt).requires_grad_(True)
    gy = torch.randn(B, T, Hd, device=dev, generator=g).to(tdt)

    def fwd_bwd():
        y = x
        for l in layers:
            y, _ = l(y)
        y.backward(gy)

    def step_eager():
        buckets.zero_grad()
        x.grad = None
        fwd_bwd()
        buckets.finish()

    step = step_eager

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(max(1, args.warmup if args.no_graph else 3)):
        step_eager()
    if not args.no_graph:
        # the whole forward+backward as one HIP graph: the step is ~40 short kernels, and launching
        # them one by one from Python costs about as much as running them
        x.grad = None
        try:
            step = buckets.capture_step(fwd_bwd)
        except Exception as e:  # fall back to eager launches rather than lose the measurement
            print(f"[bench] HIP graph capture failed ({type(e).__name__}: {e}); launching eagerly", file=sys.stderr)
            args.no_graph = True
            buckets.overlap = True
            torch.cuda.synchronize(dev)
        for _ in range(args.warmup):
            step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # second, untimed pass with event brackets around every kernel launch
    # (every rank runs it -- the steps contain collectives -- but only rank 0 records events)
    kern = {}
    if rank == 0:
        _lib.profile_reset()
        _lib.profile_enable(True)
    buckets.overlap = True
    for _ in range(min(args.steps, 10)):
        step_eager()
    torch.cuda.synchronize(dev)
    if rank == 0:
        _lib.profile_enable(False)
        kern = _lib.profile_read()
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * B * T / (elapsed / args.steps)
        per_kernel = {k: {"avg_us": v[0] / max(v[1], 1) * 1e3, "launches_per_step": v[1] / min(args.steps, 10),
                          "total_ms_per_step": v[0] / min(args.steps, 10)} for k, v in kern.items()}
        dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms_per_step"]) if per_kernel else None
        roofline = None
        if dom and kernel_work(dom, B, T, D, M, Hd, esize):
            flops, nbytes = kernel_work(dom, B, T, D, M, Hd, esize)
            # which roof is lower for this kernel's algorithmic work, at the spec peaks
            t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
            t_mfma = flops / (MFMA_PEAK_TFLOPS[args.dtype] * 1e12)
            bound = "hbm" if t_hbm >= t_mfma else "mfma"
            per_layer = per_kernel[dom]["launches_per_step"] / args.layers  # launches per layer pass
            flops, nbytes = flops / per_layer, nbytes / per_layer  # per launch, like the duration
            dur_s = per_kernel[dom]["avg_us"] * 1e-6
            if bound == "mfma":
                achieved, peak, unit = flops / dur_s / 1e12, MFMA_PEAK_TFLOPS[args.dtype], "TFLOP/s"
            else:
                achieved, peak, unit = nbytes / dur_s / 1e9, HBM_PEAK_GBS, "GB/s"
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(dom)
            roofline = {"kernel": dom, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                        "frac": achieved / peak, "traffic": traffic, "avg_us": per_kernel[dom]["avg_us"],
                        "algorithmic_flops": flops, "algorithmic_bytes": nbytes}
        # whole-step view against the layer's algorithmic HBM bytes: (3D+3Hd)*elsize B per frame and layer
        layer_bytes = args.layers * B * T * (3 * D + 3 * Hd) * esize
        t_kern = sum(v["total_ms_per_step"] for v in per_kernel.values()) * 1e-3
        out = {
            "metric": "LMU encoder fwd+bwd frames/sec",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{args.layers} stacked LMUFFT layers (input {D}, hidden {Hd}, memory {M}, theta "
                            f"{args.theta:g}, seq_len {T}) fwd+bwd, batch {B} per GPU, {args.dtype} activations with fp32 "
                            f"accumulation and fp32 parameters, conv={args.conv}; "
                            "training-throughput layer shape",
                "batch_per_gpu": B, "frames": T, "layers": args.layers, "parallelism": f"dp{world}",
                "launch": "eager" if args.no_graph else "hip graph replay",
            },
            "roofline": roofline,
            "step_hbm": {"algorithmic_bytes_per_step": layer_bytes,
                         "achieved_GBs_wall": layer_bytes / (elapsed / args.steps) / 1e9,
                         "achieved_GBs_kernels": layer_bytes / t_kern / 1e9 if t_kern else None,
                         "frac_of_peak_wall": layer_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            "kernels": per_kernel,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, layers[0].H, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
