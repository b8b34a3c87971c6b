"""GPU: bench.py's real (not dry-run) worker for the scored config and for the training config, a few steps each: the JSON line
keeps the driver's contract, the roofline is measured from the library's own HIP events, and nothing under oracle/ is touched
when the CPU baseline is switched off."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(*extra):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "4", "--warmup", "1", "--prewarm", "4",
                        "--profile-every", "1", "--no-cpu-baseline", "--traffic", "off", *extra],
                       cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("config,launches_per_step,kernel", [("c2", 8, "k_conv_wino24"), ("c4", 16, "k_conv_wino24")])
def test_bench_line_contract_and_live_roofline(config, launches_per_step, kernel):
    """c2: BASELINE configs[1] (the scored one; 8 TriplaneConv 3x3 launches per denoising step, unet_triplane.py:27-58);
    c4: TrainLoop.run_step (train_util.py:163-247) — its 8 forward and 8 input-gradient launches are all timed."""
    d = _line("--config", config)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["name"] == config and d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and 0.05 < r["frac"] < 1.0
    assert r["launches_timed"] == 4 * launches_per_step and kernel in r["kernel"]
    assert r["conv3x3_ms_per_step"] <= d["ms_per_step"] * 1.05          # (events on every step cost a little)
    assert "cpu_baseline" not in d
    if config == "c2":
        # the same workload as two independent chains, measured in the same run and reported BESIDE the headline
        c = d["chains2"]
        assert c["chains"] == 2 and c["unit"] == "samples/s" and c["value"] > 0 and c["per_chain_latency_ms"] > d["ms_per_step"] * 0.9
        assert abs(c["ms_per_step_per_sample"] * 2 - c["per_chain_latency_ms"]) < 1e-6 * c["per_chain_latency_ms"]
        assert "16 step(s)" in d["data"]
    else:
        assert d["chains2"] is None
        w = r["wgrad3x3"]                                               # the 3x3 weight-gradient launches: eight per training step
        assert w["launches_timed"] == 4 * 8 and "k_wgrad_wino" in w["kernel"] and 0.02 < w["frac"] < 1.0
        # the side stream's overlap, from the line's own trace child: a lost hardware queue would show as kernels_in_flight_2_frac ~ 0
        o = d["side_stream_overlap"]
        assert "error" not in o, o
        assert abs(o["kernels_in_flight_0_frac"] + o["kernels_in_flight_1_frac"] + o["kernels_in_flight_2_frac"] - 1.0) < 1e-3
        assert o["hardware_queues_seen"] >= 2 and o["kernels_in_flight_2_frac"] > 0.05, o


def test_bench_chains_can_be_switched_off():
    d = _line("--config", "c2", "--chains", "0")
    assert d["chains2"] is None


def test_eager_gpu_baseline_leg():
    """The PyTorch-ROCm eager leg of the baselines: the CPU port's ATen ops with their tensors on cuda:0 (bounded, a reported figure)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    b = bench.eager_gpu_baseline(budget_s=1.0)
    assert b["unit"] == "samples/s" and b["kind"] == "port" and b["value"] > 0
    assert "default" in b["modes"] and set(b["modes"]) <= {"default", "benchmark"} and all(m["steps"] >= 5 and m["ms_per_step"] > 0 for m in b["modes"].values())
    assert abs(b["value"] - 1.0 / (1000 * b["ms_per_step"] * 1e-3)) < 1e-3 * b["value"]
