#!/usr/bin/env python3
"""Headline benchmark: DDPM-1000 triplane samples/sec at a 128^2 latent (BASELINE.json configs[1]:
128-ch UNet, (H,W,D)=(128,128,128), batch 1 per GPU), through the public sampling API.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

--config selects the BASELINE.json workload (default c2 = configs[1], the one the metric is quoted on; c3 = configs[2]'s per-GPU
share: DDIM-100, batch 8 per GPU; c5 = configs[4]: the (256,256,128) retarget, DDPM-1000) — each line carries its own roofline.

A "step" is one iteration of GaussianDiffusion.p_sample_loop_progressive (the loop src/sample.py drives) for one sample per
GPU: UNet forward + fused sampler update + the step's noise draw (device generator, as the reference does on a GPU).  value = samples/s over all ranks
= N * K / 1000 / t, where t is the max over ranks of the barrier-bracketed wall time of exactly K steps.
Prints ONE JSON line on rank 0.  Multi-GPU = independent samples per rank (no data-path collective).

`--gpus N` with N > 1 and no torchrun environment: this process touches no GPU; it starts N fresh worker processes
(one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on 127.0.0.1), relays rank 0's line and exits
non-zero if any worker fails.

Before the W warm-up steps the script always runs PREWARM untimed steps of its own: a fresh box needs more than a
handful of steps to reach steady clocks and warm caches (round 1: 1.145 ms/step after 5 warm-up steps, 1.08 after 100; the
part is power-limited under this workload, so the clock it settles at is part of the result — `roofline.clock`).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MC = 128
HWD = (128, 128, 128)
T_STEPS = 1000
PREWARM = 150
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = fp32 vector peak (at the 2400 MHz data-sheet clock)
PEAK_CLOCK_MHZ = 2400.0
TRAFFIC_PROFILE = "profiles/r05_pmc_traffic.json"
# BASELINE.json workloads a single GPU can run (per-GPU share of the multi-GPU ones): name -> (hwd, batch per GPU, sampler,
# timestep_respacing, steps per sample, metric, workload text)
CONFIGS = {
    "c2": ((128, 128, 128), 1, "ddpm", "", 1000, "DDPM-1000 triplane samples/sec @128^2 latent",
           "BASELINE configs[1]: 128^2 triplane (H,W,D)=(128,128,128), 128-ch TriplaneUNetModelSmall, DDPM-1000, batch 1 per GPU; "
           "a step = 1 denoising step"),
    "c3": ((128, 128, 128), 8, "ddim", "ddim100", 100, "DDIM-100 triplane samples/sec @128^2 latent, batch 8 per GPU",
           "BASELINE configs[2] per-GPU share: 128^2 triplane, 128-ch UNet, DDIM-100, batch 64 over 8 GPUs = 8 per GPU; "
           "a step = 1 denoising step of the batch"),
    "c5": ((256, 256, 128), 1, "ddpm", "", 1000, "DDPM-1000 triplane samples/sec @(256,256,128) retarget",
           "BASELINE configs[4] (diffusion side): retargeted (H,W,D)=(256,256,128) triplane (--resize 2 2 1), 128-ch UNet, "
           "DDPM-1000, batch 1; a step = 1 denoising step"),
    # the diffusion stage of train.py: the only config whose N > 1 path has a collective (one flat-gradient all-reduce per step)
    "c4": ((92, 128, 92), 4, "train", "", 1, "diffusion training samples/sec (towerruins-size triplane, 64-ch UNet), batch 4 per GPU",
           "BASELINE configs[3] (diffusion stage) per-GPU share: TrainLoop.run_step on a (H,W,D)=(92,128,92) triplane, 64-ch "
           "TriplaneUNetModelSmall, batch 4 per GPU, data-parallel; a step = q_sample + forward + backward + gradient all-reduce "
           "(RCCL, N > 1) + AdamW/EMA + per-step weight repack"),
}
MC_OF = {"c4": 64}                   # model_channels per config (default MC)


def f_dense_per_step(mc, H, W, D, mult=(1, 2)):
    """Conv+linear flops of one UNet forward as the reference executes it (rollout channels dense):
    SURVEY.md §8d closed form — 419.06 GF at mc=128, 128^3."""
    px = [H * W + H * D + W * D]
    for _ in mult[1:]:
        H, W, D = H // 2, W // 2, D // 2
        px.append(H * W + H * D + W * D)
    f = 0.0
    c3 = lambda cin, cout, p: 2.0 * 9 * 3 * cin * cout * p
    c1 = lambda cin, cout, p: 2.0 * cin * cout * p
    f += c1(12, mc, px[0])
    ch = mc
    chans = [ch]
    for lvl, m in enumerate(mult):
        co = m * mc
        f += c3(ch, co, px[lvl]) + c3(co, co, px[lvl]) + (c1(ch, co, px[lvl]) if ch != co else 0)
        ch = co
        chans.append(ch)
    for oi, lvl in enumerate(range(len(mult) - 1, -1, -1)):
        ich = chans.pop()
        if oi == 0:
            ich = 0
        co = mult[lvl] * mc
        f += c3(ch + ich, co, px[lvl]) + c3(co, co, px[lvl]) + (c1(ch + ich, co, px[lvl]) if ch + ich != co else 0)
        ch = co
    f += c1(mc, 12, px[0])
    ted = 4 * mc
    f += 2.0 * (mc * ted + ted * ted + ted * sum(2 * m * mc for m in mult) * 2)
    return f


def usable_cores():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs but grants 16 CPUs of time; oversubscribing oneDNN 16x makes it ~40x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(steps_budget_s=20.0):
    """BASELINE.md §3 path (ii): oracle/torch_port.py (the reference's algorithm on the reference's own CPU engine,
    oneDNN) timed on this box's host cores for a bounded number of full-size steps; samples/s = 1 / (1000 * s_per_step)."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import torch_port as tp
    from sin3dm_amd import testing as T
    cores = usable_cores()
    torch.set_num_threads(cores)
    H, W, D = HWD
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=MC), 0)
    x = torch.from_numpy(T.synthetic_noise((1, 12, H + D, W + D), 1))
    t = torch.tensor([500.0])
    with torch.no_grad():
        tp.unet_forward(sd, x, t, H, W, D, MC)                       # warm-up (oneDNN primitive creation)
        n, t0 = 0, time.perf_counter()
        while True:
            y = tp.unet_forward(sd, x, t, H, W, D, MC)
            x = (0.5 * y.clamp(-1, 1) + 0.5 * x)                     # stand-in for the sampler update (negligible)
            n += 1
            dt = time.perf_counter() - t0
            if dt > steps_budget_s or n >= 40:
                break
    s_per_step = dt / n
    return {"value": 1.0 / (T_STEPS * s_per_step), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} full-size UNet steps (128-ch, 128^3, B=1) of oracle/torch_port.py (PyTorch-CPU/oneDNN, "
                      f"{cores} threads), {s_per_step * 1e3:.0f} ms/step, extrapolated to 1000 steps"}


def eager_gpu_baseline(budget_s=10.0):
    """The same port as cpu_baseline — the reference's arithmetic as the ATen ops the reference itself calls (F.conv2d on the
    dense rollout concat, F.group_norm, F.interpolate, torch.cat ...) — with its tensors on cuda:0: what a PyTorch-ROCm install
    makes of the reference's forward on this very GPU (MIOpen / rocBLAS kernels, fp32, TF32 off, eager mode, one launch per op).
    A reported baseline like the CPU legs: run after the timed region, bounded, never the thing measured or shipped."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import torch_port as tp
    from sin3dm_amd import testing as T
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda:0")
    H, W, D = HWD
    sd = {k: v.to(dev) for k, v in T.synthetic_state_dict(T.unet_param_shapes(model_channels=MC), 0).items()}
    x = torch.from_numpy(T.synthetic_noise((1, 12, H + D, W + D), 1)).to(dev)
    t = torch.tensor([500.0], device=dev)
    out = {}
    with torch.no_grad():
        for mode in ("default", "benchmark"):                         # benchmark = MIOpen's find over its solvers for every shape
            if mode == "benchmark" and out["default"]["warmup_s"] > 20.0:  # (a box whose first pass was that slow: no second search)
                break
            torch.backends.cudnn.benchmark = mode == "benchmark"
            t_w = time.perf_counter()
            for _ in range(3):
                tp.unet_forward(sd, x, t, H, W, D, MC)
            torch.cuda.synchronize()
            warm_s = time.perf_counter() - t_w
            n, t0 = 0, time.perf_counter()
            while True:
                y = tp.unet_forward(sd, x, t, H, W, D, MC)
                x = 0.5 * y.clamp(-1, 1) + 0.5 * x
                n += 1
                if n % 5 == 0:
                    torch.cuda.synchronize()
                    if time.perf_counter() - t0 > budget_s / 2 or n >= 100:
                        break
            torch.cuda.synchronize()
            out[mode] = {"ms_per_step": round((time.perf_counter() - t0) / n * 1e3, 3), "steps": n, "warmup_s": round(warm_s, 1)}
    torch.backends.cudnn.benchmark = False
    best = min(v["ms_per_step"] for v in out.values())
    return {"value": round(1.0 / (T_STEPS * best * 1e-3), 4), "unit": "samples/s", "kind": "port", "ms_per_step": best, "modes": out,
            "sample": "full-size UNet steps (128-ch, 128^3, B=1) of oracle/torch_port.py with its tensors on cuda:0 (PyTorch-ROCm eager: "
                      "MIOpen / rocBLAS, fp32), the better of MIOpen's default and find modes, extrapolated to 1000 steps"}


def cpu_baseline_c_oracle(budget_s=25.0):
    """BASELINE.md §3 path (i): the plain-C/OpenMP restatement (oracle/sin3dm_oracle.c, literal dense rollout concat,
    no vendor library) on the same host cores.  One 128-ch step at (64,64,64) — a quarter of the full-size pixels — is
    timed first; the full-size step only if it fits the budget, otherwise the figure is that step x 4 (the work is
    proportional to the pixel count) and says so."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import numpy as np
    import oracle as orc
    from sin3dm_amd import testing as T
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    sd = orc.Params(T.synthetic_state_dict(T.unet_param_shapes(model_channels=MC), 0, as_torch=False))

    def one(hwd):
        H, W, D = hwd
        x = T.synthetic_noise((1, 12, H + D, W + D), 1)
        t0 = time.perf_counter()
        orc.unet_forward(sd, x, np.array([500.0], np.float32), H, W, D, MC)
        return time.perf_counter() - t0

    q = one((64, 64, 64))
    if 4 * q <= budget_s:
        s, what = one(HWD), "1 full-size UNet step (128-ch, 128^3, B=1)"
    else:
        s, what = 4 * q, f"1 UNet step at (64,64,64) ({q:.1f} s) x 4 for the pixel count of 128^3"
    return {"value": 1.0 / (T_STEPS * s), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{what} of oracle/sin3dm_oracle.c (plain C, OpenMP, {cores} threads), {s * 1e3:.0f} ms/step, "
                      f"extrapolated to 1000 steps"}


def s3d_switches():
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("S3D_")}


# ------------------------------------------------------------------------------------------------ launcher
def spawn_workers(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: N fresh processes, one per GPU (sin3dm_amd/launcher.py).
    The parent never initialises the GPU (no HIP call, no torch.cuda.*), never exec()s, watches ALL children — the first
    non-zero exit stops the others within seconds — and forwards rank 0's JSON line."""
    from sin3dm_amd.launcher import spawn_ranks
    return spawn_ranks(os.path.abspath(__file__), argv, args.gpus)


def csrc_sha256():
    """sha256 over the kernel sources the committed PMC traffic figure was measured with (tools/pmc_traffic.py stores it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "sin3dm_amd", "csrc", "*.hip")) + glob.glob(os.path.join(REPO, "sin3dm_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


_SAMPLER_CODE = r"""
import sys, time
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    want = sys.argv[1].lower()
    h = None
    for x in hs:
        try:
            if want and want in str(amdsmi.amdsmi_get_gpu_device_bdf(x)).lower():
                h = x
        except Exception:
            pass
    if h is None and len(hs) == 1:
        h = hs[0]
    if h is None:
        raise SystemExit(0)
    print("READY", flush=True)
    while True:
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        clk = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
        if not clk and isinstance(m.get("current_gfxclk"), (int, float)):
            clk = [m["current_gfxclk"]]
        pw = m.get("current_socket_power")
        if not isinstance(pw, (int, float)) or pw >= 60000:
            pw = m.get("average_socket_power")
        if clk:
            print("%.4f %.1f %s" % (time.time(), sum(clk) / len(clk), pw if isinstance(pw, (int, float)) and pw < 60000 else "nan"), flush=True)
        time.sleep(0.01)
except Exception:
    pass
"""


class ClockSampler:
    """gfx clock / socket power of this rank's GPU (amdsmi gpu_metrics, every 10 ms) during the benchmark's own untimed pre-warm
    steps — the same load as the timed region, never during it.  The sampling runs in a CHILD process that touches no GPU (a
    thread in this process would contend for the interpreter lock with the launch loop: it starved the GPU and cost 7 % in a
    first version); this process only notes the time window and reads the child's lines afterwards.  Everything is optional: no
    amdsmi / no permission -> no figures, never a failure."""

    def __init__(self, bdf):
        import subprocess
        self._p = None
        try:
            # (never under the profiler this process may run under: its preloaded tool library would initialise the GPU in the
            # child and write trace files of its own next to ours)
            env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCPROF", "ROCP_", "ROCTX", "LD_PRELOAD", "HSA_TOOLS_LIB"))}
            self._p = subprocess.Popen([sys.executable, "-c", _SAMPLER_CODE, bdf or ""], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
        except Exception:
            self._p = None
        self._t0 = self._t1 = None

    def begin(self):
        self._t0 = time.time()

    def end(self):
        self._t1 = time.time()

    def result(self):
        if self._p is None:
            return None
        try:
            self._p.terminate()
            out, _ = self._p.communicate(timeout=5)
        except Exception:
            return None
        rows = []
        for l in out.splitlines():
            f = l.split()
            if len(f) == 3:
                try:
                    rows.append((float(f[0]), float(f[1]), float(f[2])))
                except ValueError:
                    pass
        if self._t0 is None or self._t1 is None:
            return None
        lo = self._t0 + 0.4 * (self._t1 - self._t0)                    # the clocks settle during the first steps
        s = [(c, p) for t, c, p in rows if lo <= t <= self._t1]
        if not s:
            return None
        pw = [p for _, p in s if p == p]
        return {"gfxclk_mhz_mean": round(sum(c for c, _ in s) / len(s), 0), "gfxclk_mhz_min": round(min(c for c, _ in s), 0),
                "socket_power_w_mean": round(sum(pw) / len(pw), 0) if pw else None, "samples": len(s),
                "how": "amdsmi gpu_metrics (mean over the XCDs) every 10 ms, read by a child process during the untimed pre-warm steps of this run"}


def measure_traffic_in_run(config):
    """HBM bytes per launch of the dominant kernel, measured NOW: two fresh children of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only, the program directly after `--`),
    started before this process touches the GPU.  Returns (bytes per launch, source string) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this process is itself being profiled"
    tot = {}
    tmp = tempfile.mkdtemp(prefix="s3d_pmc_")
    t_start = time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "t", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--profile-every", "0",
                   "--prewarm", "4", "--traffic", "off", "--chains", "0", "--overlap", "off", "--config", config]
            env = dict(os.environ, TMPDIR=tmp)
            r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {r.stderr[-200:]!r}"
            agg = {}
            for f in files:                                     # (every process of the pass writes a file: take the rows, not a file)
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter or "k_conv_wino" not in row["Kernel_Name"]:
                        continue
                    a = agg.setdefault("dom", [0.0, 0])
                    a[0] += float(row["Counter_Value"]); a[1] += 1
            if "dom" not in agg:
                return None, f"no k_conv_wino launches in the {counter} pass"
            tot[counter] = agg["dom"][0] / agg["dom"][1]
    except Exception as e:                                      # a profiler hiccup must not cost the benchmark line
        return None, f"in-run PMC passes failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM section): x2; WRITE_SIZE KB as is
    bytes_per_launch = int(2 * tot["FETCH_SIZE"] * 1024 + tot["WRITE_SIZE"] * 1024)
    return bytes_per_launch, ("measured_in_run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE children of this command (separate passes, "
                              "--kernel-trace only, 6 steps each), per-launch mean over the k_conv_wino24* launches, read side x2 "
                              f"(gfx950 FETCH_SIZE correction); the two passes took {time.time() - t_start:.1f} s before the benchmark touched the GPU")


def measure_side_stream_overlap(config):
    """--config c4: does the backward pass's side stream (weight gradients beside the input-gradient chain) really run beside it?
    HIP maps streams onto a few hardware queues; a process whose side stream shares the main stream's queue loses the overlap
    silently — performance, not bits.  A short fresh child of this script under `rocprofv3 --kernel-trace`
    (no counters), started before this process touches the GPU: the fraction of the traced span with 0 / 1 / >= 2 kernels in
    flight, the hardware queues seen, and the sum of the positive gaps of the busiest queue per step."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not found"}
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return {"error": "this process is itself being profiled"}
    tmp = tempfile.mkdtemp(prefix="s3d_ovl_")
    steps = 12
    try:
        cmd = ["rocprofv3", "--kernel-trace", "-d", tmp, "-o", "t", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "4", "--no-cpu-baseline", "--profile-every", "0",
               "--prewarm", "8", "--traffic", "off", "--chains", "0", "--overlap", "off", "--config", config]
        r = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), capture_output=True, text=True, timeout=300)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return {"error": f"rocprofv3 --kernel-trace child failed (rc {r.returncode}): {r.stderr[-200:]!r}"}
        ev = sorted((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row.get("Queue_Id", "?")) for f in files for row in csv.DictReader(open(f)))
        ev = ev[len(ev) * 2 // 3:]                                  # the last third: the timed steps, not model construction / warm-up
        span = ev[-1][1] - ev[0][0]
        pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
        depth, last, hist = 0, pts[0][0], [0, 0, 0]
        for t, d in pts:
            hist[min(depth, 2)] += t - last
            depth += d; last = t
        queues = {}
        for s_, e_, q in ev:
            queues.setdefault(q, []).append((s_, e_))
        main = max(queues.values(), key=lambda v: sum(e_ - s_ for s_, e_ in v))
        gaps = sum(max(0, b[0] - a[1]) for a, b in zip(main, main[1:]))
        return {"kernels_in_flight_0_frac": round(hist[0] / span, 4), "kernels_in_flight_1_frac": round(hist[1] / span, 4),
                "kernels_in_flight_2_frac": round(hist[2] / span, 4), "hardware_queues_seen": len(queues),
                "main_queue_gap_frac": round(gaps / span, 4), "kernels": len(ev), "span_ms": round(span / 1e6, 3),
                "how": f"rocprofv3 --kernel-trace child of this command ({steps} steps, last third of its kernels), measured before this process touched the GPU; "
                       "under the tracer the host issues kernels more slowly than in the timed region: the idle fraction is an upper bound"}
    except Exception as e:                                          # never cost the benchmark line
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_chains(torch, model, diffusion, shape, kw, chains, rounds):
    """The same workload as `chains` independent batch-1 sample chains in flight at once on one GPU (one HIP stream + one
    workspace lane of the SAME model handle each, steps issued round-robin from this host thread:
    GaussianDiffusion.sample_loop_chains_progressive) — what `python -m sin3dm_amd.sample` does with several batch-1 samples per
    GPU.  Reported beside the headline, never as it: BASELINE configs[1] is ONE batch-1 chain."""
    it = diffusion.sample_loop_chains_progressive(model, shape, 10 ** 9, chains=chains, model_kwargs=kw)
    with torch.no_grad():
        for _ in range(40):
            next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            cur = next(it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert all(torch.isfinite(o["sample"]).all() for _, o in cur.values())
    it.close()
    n = len(cur)
    return {"chains": n, "rounds_timed": rounds, "value": n * shape[0] * rounds / T_STEPS / dt, "unit": "samples/s",
            "ms_per_step_per_sample": dt / (n * rounds) * 1e3, "per_chain_latency_ms": dt / rounds * 1e3,
            "note": f"{n} independent batch-1 DDPM chains on {n} HIP streams / workspace lanes of one model handle, same process, measured "
                    "right after the timed region; per-sample results are bit-identical to single-chain runs "
                    "(tests/test_hip_chains.py); the headline value / ms_per_step above are ONE chain"}


# ------------------------------------------------------------------------------------------------ worker
def worker(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    (H, W, D), BATCH, sampler_kind, respacing, steps_per_sample, metric, workload = CONFIGS[args.config]
    # --force-dist: the process group (RCCL) is created at N = 1 too, so every collective of the N > 1 path — barrier, MAX
    # all-reduce, object gather, and in --config c4 the parameter broadcast + gradient all-reduce of TrainLoop — really executes
    dist_on = world > 1 or (args.force_dist and not args.dry_run)
    clock = None
    chains2 = None

    ident = {"device_index": None, "pci_bus_id": None, "uuid": None, "name": "cpu (dry run)"}
    if args.dry_run:
        # plumbing check without a GPU (tests/test_bench_launch.py): same rendezvous / barrier / max-over-ranks / JSON
        # path over gloo, the step replaced by a sleep.  Never a measurement: the line says so.
        if args.dry_run_fail_rank == rank and args.dry_run_fail_early:
            raise SystemExit(4)                                    # a rank that dies BEFORE the rendezvous
        if world > 1:
            dist.init_process_group("gloo")
        if args.dry_run_fail_rank == rank:
            raise SystemExit(3)
        step = lambda: time.sleep(0.001)
        sync = lambda: None
        dev = torch.device("cpu")
        prof = None
    else:
        from sin3dm_amd import _lib, testing as T
        from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
        from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
        from sin3dm_amd.launcher import device_identity
        ndev = torch.cuda.device_count()                             # (counting devices does not initialise the GPU)
        if local >= ndev or world > ndev:                             # (--nnodes=1 by contract)
            raise SystemExit(f"bench.py: --gpus {args.gpus} (rank {rank}, LOCAL_RANK {local}) but this node shows {ndev} GPU(s): "
                             f"one process per GPU needs N <= {ndev}")
        _lib.require_gpu()
        torch.cuda.set_device(local)
        dev = torch.device(f"cuda:{local}")
        ident = device_identity(local)
        if rank == 0:
            clock = ClockSampler(ident.get("pci_bus_id"))            # (a child process; it is sampling by the time the model is built)
        if dist_on:
            if world == 1:                                            # --force-dist: a local rendezvous of one
                from sin3dm_amd.parallel import _free_port
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
                os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl", device_id=dev)
        mc = MC_OF.get(args.config, MC)
        model = TriplaneUNetModelSmall(12, mc, 12, num_res_blocks=1, channel_mult=(1, 2), use_scale_shift_norm=True)
        model.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0))
        kw = dict(H=H, W=W, D=D)
        from sin3dm_amd import parallel
        torch.manual_seed(parallel.sample_seed(1000, rank))          # rank r's first sample is sample index r (parallel.shard_indices): ONE seed convention
        state = {"x": None}
        if sampler_kind == "train":
            # TrainLoop.run_step (src/diffusion/train_util.py:163-247) on one fixed synthetic batch, as tools/bench_train.py drives it
            from sin3dm_amd.diffusion.train_util import TrainLoop
            model.to(dev)
            diffusion = create_gaussian_diffusion(steps=T_STEPS, predict_xstart=True)
            x0 = torch.from_numpy(T.synthetic_noise((12, H + D, W + D), 400)).clamp(-1, 1).to(dev)
            batch = x0.unsqueeze(0).expand(BATCH, -1, -1, -1)

            def data():
                while True:
                    yield batch, dict(kw)
            tl = TrainLoop(model=model, diffusion=diffusion, data=data(), batch_size=BATCH, microbatch=-1, lr=5e-4, ema_rate=0.9999,
                           log_interval=10 ** 9, save_interval=10 ** 9, resume_checkpoint=False, lr_anneal_steps=25000)

            def step():
                tl.run_step(batch, dict(kw)); tl.step += 1
                state["x"] = model.flat_parameters
        else:
            model.to(dev).eval()
            diffusion = create_gaussian_diffusion(steps=T_STEPS, noise_schedule="linear", predict_xstart=True, timestep_respacing=respacing)
            assert diffusion.num_timesteps == steps_per_sample
            loop = diffusion.ddim_sample_loop_progressive if sampler_kind == "ddim" else diffusion.p_sample_loop_progressive

            def sampler():          # the public sampling loop, sample after sample (src/sample.py:38 calls p_sample_loop)
                while True:
                    for out in loop(model, (BATCH, 12, H + D, W + D), model_kwargs=kw):
                        state["x"] = out["sample"]
                        yield
            gen = sampler()
            step = lambda: next(gen)
        sync = torch.cuda.synchronize

    def barrier():
        if dist_on:
            dist.barrier()

    with torch.no_grad():
        if clock is not None:
            clock.begin()
        for _ in range(args.prewarm + args.warmup):
            step()
        sync()
        if clock is not None:
            clock.end()
        if not args.dry_run:
            # inside the timed region only the dominant class (an event pair costs ~3 us); training: also the 3x3 weight-gradient
            # launches (class 3, eight per step on the backward pass's side stream)
            model.profile(args.profile_every, classes=9 if sampler_kind == "train" else 1)
        barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        barrier()
        dt = time.perf_counter() - t0
    if not args.dry_run:
        prof = model.profile_read()
        # the 1x1 and rank-1 launches (informational fields of the line) in a short pass of their own, outside the timed region
        model.profile(1 if args.profile_every else 0, classes=6)
        with torch.no_grad():
            for _ in range(8 if args.profile_every else 0):
                step()
        sync()
        side = model.profile_read()
        for cls in (1, 2):                                # per-step figures: scaled to the timed region's profiled forwards
            scale = prof.forwards / side.forwards if side.forwards else 0.0
            prof.ms[cls] = side.ms[cls] * scale; prof.flops[cls] = side.flops[cls] * scale
            prof.mfma_flops[cls] = side.mfma_flops[cls] * scale; prof.launches[cls] = int(side.launches[cls] * scale)
        model.profile(0)
        assert torch.isfinite(state["x"]).all()
        if args.config == "c2" and world == 1 and args.chains > 1:
            chains2 = measure_chains(torch, model, diffusion, (BATCH, 12, H + D, W + D), kw, args.chains, max(args.steps, 200))

    # every rank reports who it was and what it measured: the line shows N distinct devices, not just a world size
    mine = dict(ident, rank=rank, ms_per_step=round(dt / args.steps * 1e3, 4))
    ranks, backend_world = [mine], 1
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        backend_world = dist.get_world_size()
        backend = dist.get_backend()
        try:                                   # every rank empties its C stdout (RCCL's version banner sits there until exit) BEFORE the last
            import ctypes                      # barrier: rank 0's JSON line, printed behind it, is then the last line of the job's output
            ctypes.CDLL(None).fflush(None)
            sys.stdout.flush()
        except Exception:
            pass
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return 0
    clock = clock.result() if clock is not None else None
    # N ranks must be N physical devices: keyed on what identifies the hardware (the device INDEX is LOCAL_RANK by construction
    # and proves nothing); a build that reports neither bus id nor uuid cannot be verified and says so in the line
    keys = [(r["pci_bus_id"], r["uuid"]) for r in ranks]
    verifiable = all(k != (None, None) for k in keys)
    if not args.dry_run and verifiable and len(set(keys)) != world:
        raise SystemExit(f"bench.py: {world} ranks but only {len(set(keys))} distinct devices: {ranks}")

    ms_step = dt / args.steps * 1e3
    value = world * BATCH * args.steps / steps_per_sample / dt
    training = sampler_kind == "train"
    fd = f_dense_per_step(MC_OF.get(args.config, MC), H, W, D) * BATCH * (3 if training else 1)      # (training: forward + dgrad + wgrad as the reference executes them)
    switches = s3d_switches()
    roof = None
    if prof is not None and prof.launches[0] > 0:
        sec = prof.ms[0] * 1e-3
        avg_s = sec / prof.launches[0]
        executed = prof.mfma_flops[0] / sec / 1e12                 # what the matrix cores multiply: the hardware rate
        algorithmic = prof.flops[0] / sec / 1e12
        traffic, traffic_src = getattr(args, "traffic_measured", (None, None))
        if traffic is None and not switches and args.config == "c2":     # the committed PMC passes: default kernels, the headline config ...
            why_not = traffic_src
            try:
                tp = json.load(open(os.path.join(REPO, TRAFFIC_PROFILE)))
                if tp.get("csrc_sha256") == csrc_sha256():            # ... of exactly this tree's kernel sources
                    traffic = tp["dominant_traffic_bytes_per_launch"]
                    traffic_src = (f"from_committed_profile {TRAFFIC_PROFILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command "
                                   "(separate passes, gfx950 2x read correction), kernel sources unchanged since (sha256 match); "
                                   f"not re-measured in this run ({why_not})")
                else:
                    traffic_src = (f"dropped: {TRAFFIC_PROFILE} was measured on other kernel sources "
                                   f"(csrc sha256 {str(tp.get('csrc_sha256'))[:12]} != {csrc_sha256()[:12]}); re-run tools/refresh_profiles.sh")
            except (OSError, KeyError, ValueError):
                pass
        peak_at_clock = round(PEAK_FP32_MFMA_TFLOPS * clock["gfxclk_mhz_mean"] / PEAK_CLOCK_MHZ, 1) if clock else None
        roof = {"bound": "mfma", "achieved": round(executed, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(executed / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                # the part is power-limited under this kernel (profiles/r04_clock.txt): the peak the matrix pipe has at the clock
                # the run really held (157.3 TF is the 2400-MHz data-sheet figure and stays `peak`)
                "peak_at_measured_clock": peak_at_clock, "frac_at_measured_clock": round(executed / peak_at_clock, 4) if peak_at_clock else None,
                "clock": clock,
                "algorithmic_tflops": round(algorithmic, 2),
                "hbm_gbs": round(traffic / avg_s / 1e9, 1) if traffic else None,
                "hbm_frac_of_8TBs": round(traffic / avg_s / 8e12, 4) if traffic else None,
                "kernel": "the dense 3x3 TriplaneConv kernel (own-channel part of the rollout convolution), as reported by the "
                          "library for the timed launches" + (" (the forward AND input-gradient convolutions of the training step: the same "
                          "kernel on the operator and on its transpose)" if training else "") + ": " + model.profile_kernel(0),
                "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": int(prof.launches[0]),
                "flops_per_launch_avg": prof.flops[0] / prof.launches[0],
                "mfma_flops_per_launch_avg": prof.mfma_flops[0] / prof.launches[0],
                "conv3x3_ms_per_step": round(prof.ms[0] / max(prof.forwards, 1), 4),
                "rank1_ms_per_step": round(prof.ms[2] / max(prof.forwards, 1), 4),
                "conv1x1_ms_per_step": round(prof.ms[1] / max(prof.forwards, 1), 4),
                "whole_step_mfma_frac": round((prof.mfma_flops[0] + prof.mfma_flops[1] + prof.mfma_flops[2] + prof.mfma_flops[3]) / max(prof.forwards, 1)
                                              / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                # training only: the 3x3 weight gradient (k_wgrad_wino: Winograd F(2x2,3x3), 4/9 of the direct count on the matrix
                # cores).  HIP events on the backward pass's side stream: the launches share the chip with the main chain, so this
                # is their rate IN the step; S3D_BWD_SIDE=0 gives the kernel alone
                "wgrad3x3": ({"kernel": model.profile_kernel(3), "launches_timed": int(prof.launches[3]),
                              "avg_launch_us": round(prof.ms[3] * 1e3 / prof.launches[3], 2),
                              "achieved": round(prof.mfma_flops[3] / (prof.ms[3] * 1e-3) / 1e12, 2),
                              "frac": round(prof.mfma_flops[3] / (prof.ms[3] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                              "ms_per_step": round(prof.ms[3] / max(prof.forwards, 1), 4)} if prof.launches[3] > 0 else None),
                "note": "achieved/frac = flops the matrix cores execute for the launches (Winograd multiplies fewer than the "
                        "direct count) / HIP-event time on the launch stream inside the timed region / 157.3 TF; "
                        "algorithmic_tflops = 2*9*C*Cout per output pixel (own channels only: rank-1 rollout exploited) over the "
                        "same time; the rate on the reference-executed F_dense is effective_dense_tflops"}
    line = {"metric": metric, "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic (seeded random weights; one fixed x0 batch per rank; numpy timestep sampler, device noise)" if training else
                     f"synthetic (seeded random weights incl. zero-init convs; N(0,1) x_T; eps from the device RNG, one randn launch per {max(1, (48 << 20) // (4 * BATCH * 12 * (H + D) * (W + D)))} step(s))"),
            "config": {"workload": workload, "name": args.config,
                       "steps_per_sample": steps_per_sample, "batch_per_gpu": BATCH,
                       "parallelism": (f"dp{world}: one flat-gradient all-reduce per step" if training else
                                       f"{world} independent samples" if BATCH == 1 else f"{world} x {BATCH} independent samples"),
                       "prewarm_steps": args.prewarm},
            "f_dense_gflop_per_step": round(fd / 1e9, 2),
            "effective_dense_tflops": round(fd / (ms_step * 1e-3) / 1e12 * 1.0, 2),
            "roofline": roof, "chains2": chains2, "s3d_switches": switches,
            "side_stream_overlap": getattr(args, "overlap_measured", None),
            "ranks": ranks, "per_rank_ms": [r["ms_per_step"] for r in ranks], "devices_verified_distinct": bool(verifiable) if world > 1 else None,
            "rccl_world_size": backend_world if dist_on else None, "dist_backend": backend if dist_on else None}
    if args.dry_run:
        line["data"] = "DRY RUN (no GPU work: launcher / rendezvous plumbing check only)"
        line["value"] = 0.0
    elif not args.no_cpu_baseline and args.config == "c2" and world == 1:      # (the CPU legs time the headline workload, on rank 0 at N = 1 only)
        line["cpu_baseline"] = cpu_baseline()
        line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        try:
            line["cpu_baseline_c_openmp"] = cpu_baseline_c_oracle()
        except Exception as e:                                         # the C oracle is a checker; never fail the line for it
            line["cpu_baseline_c_openmp"] = {"error": repr(e)}
        if not args.no_eager_baseline:
            try:
                line["eager_gpu_baseline"] = eager_gpu_baseline()
                line["over_eager_gpu"] = round(value / line["eager_gpu_baseline"]["value"], 2)
            except Exception as e:
                line["eager_gpu_baseline"] = {"error": repr(e)}
    try:                                       # (RCCL writes its version banner to the C stdout, flushed at exit: keep the JSON line the LAST line)
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(line), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true", help="skip the PyTorch-ROCm eager leg of the baselines (the same port on cuda:0)")
    ap.add_argument("--profile-every", type=int, default=8, help="instrument every n-th step with HIP events (0=off)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json workload (default: configs[1], the scored one)")
    ap.add_argument("--prewarm", type=int, default=PREWARM, help="untimed steps before --warmup (a fresh box needs them to reach steady clocks)")
    ap.add_argument("--traffic", choices=["auto", "off"], default="auto",
                    help="auto: measure roofline.traffic in this run with two rocprofv3 --pmc child passes (N = 1 only)")
    ap.add_argument("--overlap", choices=["auto", "off"], default="auto",
                    help="auto (--config c4, N = 1): report whether the backward pass's side stream overlaps the main chain (field `side_stream_overlap`, "
                         "from a rocprofv3 --kernel-trace child of this command)")
    ap.add_argument("--chains", type=int, default=2, help="also report the c2 workload as this many independent chains in flight (field `chains2`; 0/1 = off)")
    ap.add_argument("--force-dist", action="store_true", help="create the RCCL process group even at --gpus 1: barrier / all-reduce / gather "
                    "(and --config c4's parameter broadcast + gradient all-reduce) execute at world size 1")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--dry-run-fail-early", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_workers(args, sys.argv[1:])
    if args.traffic == "auto" and args.gpus == 1 and not args.dry_run and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not s3d_switches():
        args.traffic_measured = measure_traffic_in_run(args.config)       # BEFORE this process touches the GPU
    if args.overlap == "auto" and args.config == "c4" and args.gpus == 1 and not args.dry_run and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        args.overlap_measured = measure_side_stream_overlap(args.config)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
