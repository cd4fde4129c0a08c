#!/usr/bin/env python3
"""Benchmark of the LMUFFT hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = forward + backward through the encoder's stack of three LMUFFT
layers (hidden 384, memory 192, theta 1000, 400 frames: the layer shape the
training-throughput metric is quoted on) on one synthetic batch per GPU (64
utterances; `--workload c3`: 96, the 8-GPU data-parallel configuration), including
the data-parallel gradient exchange.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no torchrun environment starts the N ranks itself
(as child processes, before this process touches a GPU) and exits with their
status, so both launch forms measure N ranks on RCCL.

`--unit conv` measures the causal convolution and its adjoint alone (the
conv-only unit of the roofline: (2M+2) * element size bytes per frame).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec
# Dense matrix peaks, spec.  The fp32 kernels form every product from six bf16 products (three-way split of both
# factors, gemm_core_f32.h), so their ceiling in fp32-equivalent flops is the bf16 peak / 6 -- above the 157.3 TF/s of
# the fp32 MFMA instruction they used before.
MFMA_PEAK_TFLOPS = {"f32": 2500.0 / 6.0, "bf16": 2500.0}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2_lmu", choices=["c2_lmu", "c3", "c2_full"],
                    help="c2_lmu: batch 64 per GPU (single-GPU training configuration); c3: batch 96 per GPU "
                         "(the 8-GPU data-parallel configuration); c2_full: the whole model's train step at batch 64 "
                         "(LMUFFT layers, optimizer and target packing as HIP kernels, the rest torch ops)")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (overrides --workload)")
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=384)
    ap.add_argument("--memory", type=int, default=192)
    ap.add_argument("--theta", type=float, default=1000.0)
    ap.add_argument("--conv", default="auto", choices=["auto", "toeplitz", "fft", "scan"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"],
                    help="activation / MFMA operand type (fp32 accumulation either way)")
    ap.add_argument("--unit", default="layer", choices=["layer", "conv"])
    ap.add_argument("--bucket-mb", type=float, default=None,
                    help="gradient bucket size in MB (default: one bucket per layer, 0.89 MB; 'inf' = one bucket)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="exchange all gradients in one all-reduce after the backward instead of per bucket "
                         "behind the remaining backward segments")
    ap.add_argument("--overlap", action="store_true",
                    help="per-bucket exchange also for the layer workloads, whose default is one all-reduce per step "
                         "(measured on the one-rank RCCL path, three layers at batch 64: the step is 0.388 ms with one "
                         "captured all-reduce, 0.424 ms with three captured per-bucket ones, 0.476 ms with three "
                         "launched from the host -- the per-bucket form costs more than two hidden 0.85 MB exchanges save)")
    ap.add_argument("--torch-profile", action="store_true",
                    help="c2_full: also print the GPU time per kernel of a few steps (torch profiler) to stderr")
    ap.add_argument("--bf16-stream", action="store_true",
                    help="c2_full: layer norms write bf16 instead of autocast's fp32 (one more bf16 rounding per block)")
    ap.add_argument("--exchange-single-rank", action="store_true",
                    help="one rank only: still run the RCCL gradient exchange (a one-rank process group), to measure "
                         "what the exchange path costs the step on the host side")
    ap.add_argument("--host-exchange", action="store_true",
                    help="launch the all-reduces from the host between segment graphs instead of capturing them in the "
                         "step's graph")
    ap.add_argument("--async-wgrad", action="store_true",
                    help="run every layer's parameter gradients on a side stream, beside the backward of the layers below")
    ap.add_argument("--no-fuse-relu-bwd", action="store_true",
                    help="layer workload, bf16: run the ReLU backward between two stacked layers as the lower layer's "
                         "own pass instead of folding it into the upper layer's input-gradient kernel "
                         "(asr_amd.lmu.fuse_relu_backward; the gradients are bit-identical either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="one rank: skip the batch sweep (256 and 1024 utterances) and the whole-model sub-object that "
                         "the default run appends to its line after the timed region")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying HIP graphs")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 96 if args.workload == "c3" else 64
    return args


def spawn_ranks(args) -> int:
    """Start the N ranks as children through torch.distributed.run; nothing in this process has
    touched a GPU yet.  Returns the launcher's exit status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


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
        # encoder gradient straight from gz: gz against Q^T [T][Hd], causal half of the lags; reads gz and Q^T once
        "k_gu_q": (1.0 * rows * Hd * (T + 1), rows * Hd * s + T * Hd * 2 + rows * 4),
        "k_conv_fwd": (1.0 * B * T * (T + 1) * M, rows * M * s + rows * 4 + M * T * s),
        # convolution + projection in one kernel: u and x in, m and h (and the relu bit mask) out, H and W_h once
        "k_conv_proj_fwd": (1.0 * B * T * (T + 1) * M + 2.0 * rows * MD * Hd,
                            rows * (4 + M * s + D * s + Hd * s + Hd // 8) + M * T * s + Hd * MD * s),
        "k_conv_bwd": (1.0 * B * T * (T + 1) * M, rows * M * s + rows * 4 + M * T * s),
        # state-space route (bf16): two dense GEMMs of K = M + 128 against a constant matrix of (M + 128) x 128 M
        "k_scan_gemm_fwd": (2.0 * rows * M * (M + 128), rows * M * 2 + 128 * M * (M + 128) * 2),
        "k_scan_gemm_adj": (2.0 * rows * M * (M + 128), rows * M * 2 + 128 * M * (M + 128) * 2),
        "k_encode_fwd": (2.0 * rows * D, rows * D * s + rows * 4),
        "k_encode_bwd": (2.0 * rows * D, rows * D * s + rows * 4),
    }
    return table.get(name)


def host_threads():
    """Threads for the CPU baseline: the job's CPU share, not the machine's core count (a GPU box
    exports its share as OMP_NUM_THREADS; os.cpu_count() there is the whole host)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return n


def cpu_baseline(args, H, seconds):
    """The reference's own route (torch.fft + autograd) timed on the host cores."""
    import torch

    from oracle.lmufft_torch_port import PortLMUFFT

    threads = host_threads()
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
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"same workload (batch {args.batch} x {args.frames} frames x {args.layers} layers, fp32, "
                  f"torch.fft route + autograd), median of {len(times)} steps, {med * 1e3:.1f} ms/step",
        "one_thread": {"value": args.batch * args.frames / one[1], "ms_per_step": one[1] * 1e3, "steps": 3},
    }


def read_profile(_lib, steps):
    kern = _lib.profile_read()
    return {k: {"avg_us": v[0] / max(v[1], 1) * 1e3, "launches_per_step": v[1] / steps,
                "total_ms_per_step": v[0] / steps} for k, v in kern.items()}


def roofline_of(dom, per_kernel, work, dtype, per_layer_launches):
    flops, nbytes = work
    t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
    t_mfma = flops / (MFMA_PEAK_TFLOPS[dtype] * 1e12)
    bound = "hbm" if t_hbm >= t_mfma else "mfma"
    flops, nbytes = flops / per_layer_launches, nbytes / per_layer_launches  # per launch, like the duration
    dur_s = per_kernel[dom]["avg_us"] * 1e-6
    if bound == "mfma":
        achieved, peak, unit = flops / dur_s / 1e12, MFMA_PEAK_TFLOPS[dtype], "TFLOP/s"
    else:
        achieved, peak, unit = nbytes / dur_s / 1e9, HBM_PEAK_GBS, "GB/s"
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(dom)
    out = {"kernel": dom, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
           "frac": achieved / peak, "traffic": traffic, "avg_us": per_kernel[dom]["avg_us"],
           "algorithmic_flops": flops, "algorithmic_bytes": nbytes}
    if dtype == "f32" and bound == "mfma":
        out["peak_note"] = "fp32-equivalent flops; products as six bf16 MFMAs: bf16 dense peak / 6 (fp32 MFMA: 157.3)"
    return out


def bench_conv(args, dev):
    """Conv-only unit: causal convolution + adjoint through the C ABI's conv entry points."""
    import torch

    from asr_amd import _lib
    from asr_amd import functional as F
    from asr_amd.lmu import impulse_response, legendre_state_space

    B, T, M = args.batch, args.frames, args.memory
    A, Bm = legendre_state_space(M, args.theta)
    H, _ = impulse_response(A, Bm, T)
    H = H.to(dev)
    g = torch.Generator(device=dev).manual_seed(1234)
    u = torch.relu(torch.randn(B, T, device=dev, generator=g))
    esize = 2 if args.dtype == "bf16" else 4
    routes = ["toeplitz"] if args.dtype == "bf16" else (["toeplitz", "fft"] if F.fft_supported(T, M) else ["toeplitz"])
    scanprep = None
    if args.dtype == "bf16" and F.scan_supported(M):
        routes.append("scan")
        scanprep = F.prepare_scan_bf16(A, Bm, device=dev)
    out = {}
    for route in routes:
        run = F.conv_pair_runner(u, H, route, args.dtype, scanprep=scanprep)  # preallocated outputs and workspaces
        for _ in range(args.warmup):
            run()
        torch.cuda.synchronize(dev)
        # ten pairs per HIP graph, so that the wall clock sees the kernels and their boundaries, not the host's
        # launch latency (the pair is three to four launches of 5-30 us)
        reps = 10
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(reps):
                    run()
        torch.cuda.current_stream(dev).wait_stream(side)
        n_replays = max(1, args.steps // reps)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n_replays):
            graph.replay()
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t0) / (n_replays * reps)
        _lib.profile_reset()
        _lib.profile_enable(True)
        for _ in range(10):
            run()
        torch.cuda.synchronize(dev)
        _lib.profile_enable(False)
        per_kernel = read_profile(_lib, 10)
        t_k = sum(v["total_ms_per_step"] for v in per_kernel.values()) * 1e-3
        nbytes = B * T * (2 * M + 2) * esize  # u in, m out; gm in, gu out
        per_kernel = {k: v for k, v in per_kernel.items() if v["launches_per_step"] > 0}
        # priced on the wall clock of the replayed pairs; the event-bracketed kernel times (eager launches, each
        # bracket includes its dispatch) are listed beside it
        out[route] = {"wall_us": wall * 1e6, "kernels_us": t_k * 1e6, "algorithmic_bytes": nbytes,
                      "achieved_GBs": nbytes / wall / 1e9, "frac": nbytes / wall / 1e9 / HBM_PEAK_GBS,
                      "kernels": per_kernel}
        if route != "scan":
            # Supplementary, NOT the line's value: the same ten forwards and ten adjoints as two independent chains
            # on two branches of one graph (the adjoint reads a gradient tensor of its own), i.e. what the pair costs
            # when each direction's launch gaps are filled by the other -- as they are inside a training step,
            # where other kernels sit between them.
            fwd, adj = F.conv_pair_runner(u, H, route, args.dtype, scanprep=scanprep, split=True)
            fwd(), adj()
            torch.cuda.synchronize(dev)
            g2 = torch.cuda.CUDAGraph()
            s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            s1.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s1):
                with torch.cuda.graph(g2, stream=s1):
                    s2.wait_stream(s1)
                    for _ in range(reps):
                        fwd()
                    with torch.cuda.stream(s2):
                        for _ in range(reps):
                            adj()
                    s1.wait_stream(s2)
            torch.cuda.current_stream(dev).wait_stream(s1)
            for _ in range(3):
                g2.replay()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n_replays):
                g2.replay()
            torch.cuda.synchronize(dev)
            wall2 = (time.perf_counter() - t0) / (n_replays * reps)
            out[route]["two_branches"] = {"wall_us": wall2 * 1e6, "frac": nbytes / wall2 / 1e9 / HBM_PEAK_GBS,
                                          "note": "forward chain and adjoint chain on parallel graph branches"}
    best = max(out, key=lambda r: out[r]["frac"])
    return {
        "metric": "LMU causal convolution fwd+adjoint frames/sec", "value": B * T / (out[best]["wall_us"] * 1e-6),
        "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": out[best]["wall_us"] * 1e-3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"conv-only unit: u[{B},{T}] * H[{M},{T}] -> m and its adjoint, {args.dtype}",
                   "route": best},
        "roofline": {"kernel": f"conv fwd+adjoint ({best})", "bound": "hbm", "achieved": out[best]["achieved_GBs"],
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": out[best]["frac"], "traffic": None},
        "routes": out,
    }


def bench_full_step(args, dev, rank, world, steps=None, warmup=None):
    """The reference's whole training step at its single-GPU configuration: mel-shaped input -> 3-layer encoder
    (attention, LMUFFT, layer norm, dropout 0.1) -> CTC head and loss -> backward -> clip 10 -> AdamW -> scheduler
    (configs/base_config.yaml:3-15, 29-36; src/training/distributed.py:139-234), bf16 autocast.  This package's
    kernels run the LMUFFT layers, attention, every Linear layer (the vocabulary projection included), the layer
    norms with their dropout (forward and backward), the CTC loss with its gradient and the gradient norm /
    clipping / AdamW step; the log-softmax, the loss mean and a few small casts are torch ops.  This line measures
    the model with the hot path swapped in, not a roofline."""
    import torch
    import torch.distributed as dist

    from asr_amd import FlatAdamW, LearningRateScheduler, _lib
    from asr_amd.harness import LMUASRModel
    from asr_amd.parallel import GradBuckets

    B, T = args.batch, args.frames
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    torch.manual_seed(1234)
    model = LMUASRModel(80, args.hidden, args.memory, args.layers, args.theta, T, 34, num_attention_heads=6,
                        dropout=0.1).to(dev).train()
    for mod in model.modules():  # src/models/asr_model.py:271-276
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(mod.weight)
            torch.nn.init.zeros_(mod.bias)
    enc = model.encoder
    enc.bf16_stream = bool(args.bf16_stream)
    if args.async_wgrad:  # weight gradients of the Linear and LMUFFT layers on a side stream (a branch of the graph)
        import asr_amd.linear as _lin
        _lin.ASYNC_WEIGHT_GRAD = True
        for l in enc.lmu_layers:
            l.async_weight_grad = True
    buckets = GradBuckets(model.bucket_groups(), bucket_mb=args.bucket_mb if args.bucket_mb is not None else 4.0,
                          overlap=not args.no_overlap, exchange_single_rank=args.exchange_single_rank)
    # the step count and the learning rate live in device memory, so that the whole step -- optimizer and
    # scheduler included -- can be one HIP graph
    opt = FlatAdamW(buckets=buckets, lr=1.5e-3, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=10.0,
                    device_clock=True)
    sched = LearningRateScheduler(opt, warmup_steps=1500, max_lr=1.5e-3)
    g = torch.Generator(device=dev).manual_seed(1234 + 2 + rank)
    spec = torch.randn(B, T, 80, device=dev, generator=g)
    in_len = torch.full((B,), T, dtype=torch.long, device=dev)
    tgt_len = torch.randint(20, 61, (B,), device=dev, generator=g)
    targets = torch.randint(0, 33, (B, 60), device=dev, generator=g)
    # The lengths stay on the device (the CTC kernel reads them there): the step makes no host-to-device copy and
    # no read-back.
    in_len32, tgt_len32 = in_len.to(torch.int32), tgt_len.to(torch.int32)
    losses = []
    out = {}

    def forward_backward():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lp = model.log_probs_for_training(spec, in_len)
        loss = model.compute_loss(lp, targets, in_len32, tgt_len32)
        loss.backward()
        out["loss"] = loss.detach()

    def step_eager():
        buckets.zero_grad()
        forward_backward()
        buckets.finish()
        opt.step()
        sched.step()
        return out["loss"]

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    step = step_eager
    launch = "eager launches"
    for _ in range(max(3, warmup if args.no_graph else 3)):
        losses.append(step_eager().clone())
    if not args.no_graph:
        # forward, loss, backward, gradient exchange, clipping, AdamW and the scheduler's rule as ONE HIP graph:
        # a step is one replay, with no Python and no host-side launch in it
        torch.cuda.synchronize(dev)
        out.clear()
        try:
            replay = buckets.capture_graph_exchange([forward_backward], overlap=not args.no_overlap, after=opt.step)
            launch = "one hip graph (forward, CTC loss, backward, exchange, clip, AdamW, scheduler)"

            def step():
                replay()
                sched.step()  # host copy of the count, for logging; the device evaluates the rule itself
                return out["loss"]
        except Exception as e:
            print(f"[bench] capturing the whole step failed ({type(e).__name__}: {e}); launching eagerly", file=sys.stderr)
            torch.cuda.synchronize(dev)
            buckets._replay = False
            step = step_eager
        for _ in range(warmup):
            losses.append(step().clone())
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    losses.append(out["loss"].clone())
    rank_ms = [elapsed / steps * 1e3] * 2
    if world > 1:
        tt = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rank_ms = [-float(tt[1]) / steps * 1e3, float(tt[0]) / steps * 1e3]
        elapsed = float(tt[0])
    per_kernel = {}
    psteps = min(steps, 5)
    buckets._replay = False
    out.clear()  # no autograd graph of the captured step may stay alive into the eager steps (stream of its nodes)
    if rank == 0:
        _lib.profile_reset()
        _lib.profile_enable(True)
    for _ in range(psteps):
        step_eager()
    torch.cuda.synchronize(dev)
    if rank == 0:
        _lib.profile_enable(False)
        per_kernel = read_profile(_lib, psteps)
    if rank == 0 and getattr(args, "torch_profile", False):
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(psteps):
                step_eager()
            torch.cuda.synchronize(dev)
        rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
        tot = sum(r.self_device_time_total for r in rows)
        print(f"[torch profiler] GPU time {tot / psteps / 1e3:.3f} ms/step over {psteps} steps", file=sys.stderr)
        for r in rows[:45]:
            print(f"{r.self_device_time_total / psteps:9.1f} us/step  x{r.count / psteps:6.1f}  {r.key[:120]}", file=sys.stderr)
    if rank != 0:
        return None
    hip_ms = sum(v["total_ms_per_step"] for v in per_kernel.values())
    first, last = float(losses[0].detach()), float(losses[-1].detach())
    # roofline of the entry with the largest share of the step (event-bracketed entry points; "k_attention_bwd" is
    # the two backward sweeps of one attention layer together).  Algorithmic work per launch at this shape:
    heads, dk, C = 6, args.hidden // 6, args.hidden
    attn_fwd_flops = 4.0 * B * heads * T * T * dk          # scores and context
    rows = B * T
    work = {
        "k_attention_bwd": (2.5 * attn_fwd_flops, rows * C * 2 * 8),         # q k v ctx dctx in, dq dk dv out
        "k_attention_fwd_train": (attn_fwd_flops, rows * C * 2 * 4),
        "k_linear_wgrad_bf16": (2.0 * rows * C * C, rows * C * 2 * 2 + C * C * 4),
        "k_wgrad_bf16": kernel_work("k_wgrad_bf16", B, T, C, args.memory, C, 2),
        "k_linear_ws_bf16": (2.0 * rows * C * C, rows * C * 2 * 2 + C * C * 2),
        "k_layernorm_bwd": (10.0 * rows * C, rows * C * (4 + 4 + 2 + 4)),
    }
    roofline = None
    if per_kernel:
        dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms_per_step"])
        if dom in work:
            roofline = roofline_of(dom, per_kernel, work[dom], "bf16", 1.0)
            roofline["note"] = ("algorithmic work of one launch of the entry point at this shape; the Linear entries "
                                "are priced at the 384 -> 384 layer")
    return {
        "metric": "full train step frames/sec (whole model, hot path swapped in)",
        "value": world * B * T / (elapsed / steps), "unit": "frames/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"3-layer LMU ASR model (80 -> {args.hidden}/{args.memory}, 6 heads, vocab 34, dropout 0.1, "
                               f"seq_len {T}), batch {B} per GPU, bf16 autocast: forward + CTC loss + backward + clip 10 + "
                               "AdamW(1.5e-3, wd 0.01) + warm-up scheduler",
                   "launch": launch, "rank_ms_per_step_min_max": rank_ms,
                   "optimizer_steps_on_device": list(opt.device_steps()), "learning_rate_on_device": opt.device_lr(),
                   "batch_per_gpu": B, "frames": T, "parallelism": f"dp{world}", "world_size": world,
                   "hip_kernels": "LMUFFT, attention, Linear (incl. vocabulary projection), layer norm + dropout (forward "
                                  "and backward), CTC loss + gradient, gradient norm + clip + AdamW",
                   "torch_ops": "log-softmax, loss mean, input cast",
                   "activation_stream": "bf16 between blocks" if args.bf16_stream else "fp32 between blocks (autocast)",
                   "gradient_buckets": len(buckets.buckets)},
        "roofline": roofline,
        "hip_kernel_ms_per_step": hip_ms,
        "loss_first_last": [first, last],
        "kernels": per_kernel,
    }


_RESULT_FD = None


def keep_stdout_for_the_result():
    """From here on everything written to file descriptor 1 goes to stderr (RCCL prints a version banner on
    stdout when its first communicator comes up); emit() writes the one JSON line to the real stdout."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


def emit(obj) -> None:
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def bench_layer_stack(args, dev, rank, world, batch, steps, warmup, detail=True):
    """Forward + backward of the stack of LMUFFT layers with its gradient exchange, at `batch` utterances per GPU.
    Returns the result object on rank 0 (None elsewhere) and the first layer's H."""
    import torch
    import torch.distributed as dist

    from asr_amd import LMUFFT, _lib
    from asr_amd.parallel import GradBuckets

    B, T, D, M, Hd = batch, args.frames, args.hidden, args.memory, args.hidden
    torch.manual_seed(1234)  # identical parameters on every rank
    layers = []
    for _ in range(args.layers):
        l = LMUFFT(D, Hd, M, T, args.theta, conv_algo=args.conv)
        # the reference model re-initialises every nn.Linear like this (src/models/asr_model.py:271-276)
        torch.nn.init.xavier_uniform_(l.W_u.weight)
        torch.nn.init.xavier_uniform_(l.W_h.weight)
        torch.nn.init.zeros_(l.W_u.bias)
        torch.nn.init.zeros_(l.W_h.bias)
        l.async_weight_grad = bool(args.async_wgrad)
        layers.append(l.to(dev))
    fuse_relu = args.dtype == "bf16" and not args.no_fuse_relu_bwd and D % 8 == 0 and Hd % 8 == 0 and args.layers > 1
    if fuse_relu:
        from asr_amd.lmu import fuse_relu_backward

        for lo, up in zip(layers[:-1], layers[1:]):
            fuse_relu_backward(lo, up)
    overlap = args.overlap and not args.no_overlap
    buckets = GradBuckets(list(reversed(layers)), bucket_mb=args.bucket_mb, overlap=overlap,
                          exchange_single_rank=args.exchange_single_rank)
    g = torch.Generator(device=dev).manual_seed(1234 + 2 + rank)  # each rank its own utterances
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    esize = 2 if args.dtype == "bf16" else 4
    x = torch.randn(B, T, D, device=dev, generator=g).to(tdt).requires_grad_(True)
    gy = torch.randn(B, T, Hd, device=dev, generator=g).to(tdt)

    # The step as segments: [forward of every layer + backward of the last], then one backward
    # segment per remaining layer.  The layers are chained through detached copies, so each
    # backward segment is its own autograd call (and its own HIP graph); a layer's gradient
    # bucket is complete when its segment ends and is exchanged while the next segments run.
    st = {}

    def seg_first():
        ins, outs = [], []
        y = x
        for i, l in enumerate(layers):
            xin = y if i == 0 else y.detach().requires_grad_(True)
            y, _ = l(xin)
            ins.append(xin)
            outs.append(y)
        st["ins"], st["outs"] = ins, outs
        outs[-1].backward(gy)

    def seg_back(k):
        return lambda: st["outs"][k].backward(st["ins"][k + 1].grad)

    segments = [seg_first] + [seg_back(k) for k in range(len(layers) - 2, -1, -1)]

    def step_eager():
        buckets.zero_grad()
        x.grad = None
        for s in segments:
            s()
        buckets.finish()

    step = step_eager
    launch_mode = None

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(max(1, warmup if args.no_graph else 3)):
        step_eager()
    if not args.no_graph:
        # the step replayed from HIP graphs: it is ~30 short kernels, and launching them one by one
        # from Python costs about as much as running them
        # no autograd graph of the eager steps may stay alive: its gradient-accumulation nodes belong to
        # the stream they were made on and would be reused inside the capture
        st.clear()
        x.grad = None
        try:
            if not args.host_exchange:
                # one graph for the whole step, the exchange (if there is a process group) captured inside it; if
                # RCCL refuses the capture, the segments become graphs of their own and the host launches it
                try:
                    step = buckets.capture_graph_exchange(segments, overlap=overlap)
                    launch_mode = "one hip graph" + (", all-reduces captured inside it" if buckets._distributed() else "")
                except Exception as e:
                    print(f"[bench] capturing the all-reduces failed ({type(e).__name__}: {e}); launching them "
                          "from the host between segment graphs", file=sys.stderr)
                    torch.cuda.synchronize(dev)
                    st.clear()
                    x.grad = None
                    step = buckets.capture_pipeline(segments, overlap=overlap)
            else:
                step = buckets.capture_pipeline(segments, overlap=overlap)
        except Exception as e:  # fall back to eager launches rather than lose the measurement
            print(f"[bench] HIP graph capture failed ({type(e).__name__}: {e}); launching eagerly", file=sys.stderr)
            args.no_graph = True
            torch.cuda.synchronize(dev)
        for _ in range(warmup):
            step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / steps * 1e3] * 2
    if world > 1:
        tt = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rank_ms = [-float(tt[1]) / steps * 1e3, float(tt[0]) / steps * 1e3]
        elapsed = float(tt[0])

    # What the exchange costs the step where it cannot hide: the same step captured again WITHOUT its collectives
    # (every rank on its own; the gradients are not synchronised in these untimed-for-the-metric steps), timed the
    # same way.  exposed = step with exchange - slowest rank's step without.
    exchange = None
    if detail and buckets._distributed() and not args.no_graph:
        buckets.exchange_enabled = False
        try:
            torch.cuda.synchronize(dev)
            st.clear()
            x.grad = None
            solo = buckets.capture_graph_exchange(segments, overlap=overlap)
            nsolo = max(10, min(steps, 100))
            for _ in range(5):
                solo()
            sync()
            t1 = time.perf_counter()
            for _ in range(nsolo):
                solo()
            torch.cuda.synchronize(dev)
            mine = (time.perf_counter() - t1) / nsolo * 1e3
            solo_ms = [mine, mine]
            if world > 1:
                tt = torch.tensor([mine, -mine], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                solo_ms = [-float(tt[1]), float(tt[0])]
            exchange = {"ms_per_step_without_exchange_min_max_over_ranks": solo_ms,
                        "exposed_exchange_ms": elapsed / steps * 1e3 - solo_ms[1], "steps": nsolo}
        except Exception as e:
            print(f"[bench] the step without exchange could not be captured ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.synchronize(dev)
        finally:
            buckets.exchange_enabled = True

    # second, untimed pass with event brackets around every kernel launch
    # (every rank runs it -- the steps contain collectives -- but only rank 0 records events)
    per_kernel = {}
    psteps = min(steps, 10)
    # no autograd graph of the captured steps may stay alive into the eager ones: its gradient-accumulation nodes
    # belong to the capture's stream
    st.clear()
    x.grad = None
    if rank == 0:
        _lib.profile_reset()
        _lib.profile_enable(True)
    buckets._replay = False
    for _ in range(psteps):
        step_eager()
    torch.cuda.synchronize(dev)
    if rank == 0:
        _lib.profile_enable(False)
        per_kernel = read_profile(_lib, psteps)
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_step = elapsed / steps * 1e3
        value = world * B * T / (elapsed / steps)
        dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms_per_step"]) if per_kernel else None
        roofline = None
        if dom and kernel_work(dom, B, T, D, M, Hd, esize):
            roofline = roofline_of(dom, per_kernel, kernel_work(dom, B, T, D, M, Hd, esize), args.dtype,
                                   per_kernel[dom]["launches_per_step"] / args.layers)
        # whole-step view against the layer's algorithmic HBM bytes: (3D+3Hd)*elsize B per frame and layer
        layer_bytes = args.layers * B * T * (3 * D + 3 * Hd) * esize
        t_kern = sum(v["total_ms_per_step"] for v in per_kernel.values()) * 1e-3
        out = {
            "metric": "LMU encoder fwd+bwd frames/sec",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
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
                "world_size": world, "backend": "nccl (RCCL)" if world > 1 else (
                    "nccl (RCCL), one-rank group: exchange path exercised" if args.exchange_single_rank
                    else "none (single rank)"),
                "gradient_buckets": len(buckets.buckets),
                "bucket_mb": [round(b.numel() * 4 / (1 << 20), 3) for b in buckets.buckets],
                "exchange": "per bucket, behind the remaining backward segments" if overlap else "one all-reduce after the backward",
                "relu_backward_between_layers": ("folded into the upper layer's input-gradient kernel (bit-identical "
                                                 "gradients)") if fuse_relu else "the lower layer's own pass",
                "launch": "eager" if args.no_graph else (launch_mode or f"hip graph replay ({len(segments)} segments)"),
                # wall clock of the timed steps on the fastest and the slowest rank (barriers on both sides)
                "rank_ms_per_step_min_max": rank_ms,
                "exchange_cost": exchange,
            },
            "roofline": roofline,
            "step_hbm": {"algorithmic_bytes_per_step": layer_bytes,
                         "achieved_GBs_wall": layer_bytes / (elapsed / steps) / 1e9,
                         "achieved_GBs_kernels": layer_bytes / t_kern / 1e9 if t_kern else None,
                         "frac_of_peak_wall": layer_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS},
            "kernels": per_kernel,
        }
    H = layers[0].H
    buckets.remove()
    return (out if rank == 0 else None), H


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        raise SystemExit(spawn_ranks(args))
    keep_stdout_for_the_result()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the LMUFFT layer has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.unit == "conv":
        emit(bench_conv(args, dev))
        return
    if world == 1 and args.exchange_single_rank:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        # a collective that never completes (a rank lost, a capture RCCL cannot replay) ends the run with an error
        # after three minutes instead of holding the node until somebody kills it
        import datetime
        dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=180))
        if dist.get_world_size() != args.gpus or dist.get_backend() != "nccl":
            raise SystemExit(f"process group has {dist.get_world_size()} ranks on {dist.get_backend()}, "
                             f"expected {args.gpus} on nccl")

    if args.workload == "c2_full":
        out = bench_full_step(args, dev, rank, world)
        if rank == 0:
            emit(out)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    out, H = bench_layer_stack(args, dev, rank, world, args.batch, args.steps, args.warmup)
    if rank == 0:
        default_shape = (args.batch, args.frames, args.layers, args.hidden, args.memory) == (64, 400, 3, 384, 192)
        if world == 1 and not args.no_extras and default_shape and args.workload == "c2_lmu":
            # The same step at larger batches (more workgroups per CU: kernel efficiency apart from the
            # one-tile-per-CU fill regime of batch 64), and the whole model's train step; both short, after the
            # timed region
            sweep = []
            for bsz in (256, 1024):
                sub, _ = bench_layer_stack(args, dev, rank, world, bsz, 40, 5, detail=False)
                sweep.append({"batch_per_gpu": bsz, "ms_per_step": sub["ms_per_step"], "value": sub["value"],
                              "frac_of_peak_wall": sub["step_hbm"]["frac_of_peak_wall"],
                              "kernels_avg_us": {k: round(v["avg_us"], 2) for k, v in sub["kernels"].items()
                                                 if v["launches_per_step"] > 0}})
                del sub
                torch.cuda.empty_cache()
            out["batch_sweep"] = sweep
            if args.dtype == "bf16":
                full = bench_full_step(args, dev, rank, world, steps=60, warmup=10)
                out["c2_full"] = {k: full[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup",
                                                      "hip_kernel_ms_per_step", "loss_first_last")}
                out["c2_full"]["launch"] = full["config"]["launch"]
                out["c2_full"]["workload"] = full["config"]["workload"]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, H, args.cpu_seconds)
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
