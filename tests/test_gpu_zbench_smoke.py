"""bench.py as the driver runs it: fresh child processes, one JSON line on stdout, parsed here.

The variants cover what a multi-GPU run adds to the single-rank one without needing a second GPU: the RCCL
exchange path on a one-rank group (captured inside the step's graph), the per-bucket overlap form, the whole-model
step with and without the exchange.  Short runs: the numbers are not judged, the line's shape and sanity are."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]  # exactly one line on stdout (RCCL's banner goes to stderr)
    return json.loads(lines[0]), r.stderr


def _common(d, n_gpus=1):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0
    lo, hi = d["config"]["rank_ms_per_step_min_max"]
    assert 0 < lo <= hi


@pytest.mark.parametrize("overlap", [False, True])
def test_layer_stack_with_the_exchange_path_on_one_rank(overlap):
    d, err = _run("--gpus", "1", "--exchange-single-rank", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                  "--no-extras", *(["--overlap"] if overlap else []))
    _common(d)
    cfg = d["config"]
    assert "one-rank group" in cfg["backend"]
    assert cfg["launch"].startswith("one hip graph") and "all-reduces captured" in cfg["launch"], (cfg["launch"], err[-2000:])
    assert ("per bucket" in cfg["exchange"]) == overlap
    assert d["metric"] == "LMU encoder fwd+bwd frames/sec" and d["dtype"] == "bf16"
    assert d["roofline"]["kernel"] in d["kernels"] and 0 < d["roofline"]["frac"] < 1
    ex = cfg["exchange_cost"]
    assert ex is not None and ex["ms_per_step_without_exchange_min_max_over_ranks"][1] > 0
    # the default workload is the one a multi-rank launch runs on every rank
    assert cfg["batch_per_gpu"] == 64 and cfg["frames"] == 400 and cfg["layers"] == 3
    assert "AccumulateGrad" not in err, err[-3000:]


def test_default_line_carries_sweep_and_whole_model_objects():
    d, err = _run("--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    _common(d)
    assert d["config"]["backend"].startswith("none") and d["config"]["exchange_cost"] is None
    assert [s["batch_per_gpu"] for s in d["batch_sweep"]] == [256, 1024]
    for s in d["batch_sweep"]:
        assert s["ms_per_step"] > 0 and 0 < s["frac_of_peak_wall"] < 1
    # larger batches fill the chip better: frames/s do not fall
    assert d["batch_sweep"][0]["value"] > 0.9 * d["value"]
    full = d["c2_full"]
    assert full["launch"].startswith("one hip graph"), (full["launch"], err[-3000:])
    assert full["ms_per_step"] > 0 and full["loss_first_last"][1] < full["loss_first_last"][0]
    assert "AccumulateGrad" not in err, err[-3000:]


@pytest.mark.parametrize("exchange", [False, True])
def test_whole_model_step(exchange):
    d, err = _run("--workload", "c2_full", "--steps", "3", "--warmup", "2",
                  *(["--exchange-single-rank"] if exchange else []))
    _common(d)
    assert d["config"]["launch"].startswith("one hip graph"), (d["config"]["launch"], err[-3000:])
    sched_steps, opt_steps = d["config"]["optimizer_steps_on_device"]
    assert sched_steps == opt_steps and opt_steps >= 3 + 2 + 3  # eager warm-up, replays, profiled eager steps
    assert 0 < d["config"]["learning_rate_on_device"] < 1.5e-3  # inside the 1500-step warm-up
    first, last = d["loss_first_last"]
    assert last == last and first == first


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_conv_only_unit_line(dtype):
    d, err = _run("--unit", "conv", "--dtype", dtype, "--batch", "8", "--frames", "200", "--steps", "20", "--warmup", "2")
    assert d["metric"] == "LMU causal convolution fwd+adjoint frames/sec" and d["dtype"] == dtype and d["value"] > 0
    assert d["config"]["route"] in d["routes"] and 0 < d["roofline"]["frac"] < 1
    for name, r in d["routes"].items():
        assert r["wall_us"] > 0 and r["algorithmic_bytes"] == 8 * 200 * (2 * 192 + 2) * (2 if dtype == "bf16" else 4)
        if name != "scan":  # the supplementary figure: both directions as independent chains of one graph
            assert r["two_branches"]["wall_us"] > 0 and 0 < r["two_branches"]["frac"] < 1  # (a short run: no bound on the ratio)
