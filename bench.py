#!/usr/bin/env python3
"""Headline benchmark: DDPM-1000 triplane samples/sec at a 128^2 latent (BASELINE.json configs[1]:
128-ch UNet, (H,W,D)=(128,128,128), batch 1 per GPU), through the public sampling API.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one ancestral denoising step of one sample per GPU: UNet forward + fused sampler update +
the step's noise draw (device generator, as the reference does on a GPU).  value = samples/s over all ranks
= N * K / 1000 / t, where t is the max over ranks of the barrier-bracketed wall time of exactly K steps.
Prints ONE JSON line on rank 0.  Multi-GPU = independent samples per rank (no data-path collective).
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
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = fp32 vector peak


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
    """oracle/torch_port.py (the reference's algorithm on the reference's own CPU engine, oneDNN) timed on this
    box's host cores for a bounded number of full-size steps; samples/s = 1 / (1000 * s_per_step)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-every", type=int, default=8, help="instrument every n-th step with HIP events (0=off)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sin3dm_amd import _lib, testing as T
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    _lib.require_gpu()
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    H, W, D = HWD
    model = TriplaneUNetModelSmall(12, MC, 12, num_res_blocks=1, channel_mult=(1, 2), use_scale_shift_norm=True)
    model.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=MC), 0))
    model.to(dev).eval()
    diffusion = create_gaussian_diffusion(steps=T_STEPS, noise_schedule="linear", predict_xstart=True)
    kw = dict(H=H, W=W, D=D)
    torch.manual_seed(1000 + rank)
    x = torch.randn(1, 12, H + D, W + D, device=dev)

    all_t = torch.arange(T_STEPS, device=dev, dtype=torch.int64)[:, None].contiguous()     # as GaussianDiffusion._loop does

    def run(n, x, start):
        with torch.no_grad():
            for k in range(n):
                i = (T_STEPS - 1 - (start + k)) % T_STEPS
                x = diffusion.p_sample(model, x, all_t[i], model_kwargs=kw)["sample"]
        return x

    def barrier():
        if world > 1:
            dist.barrier()

    x = run(args.warmup, x, 0)
    torch.cuda.synchronize()
    model.profile(args.profile_every)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = run(args.steps, x, args.warmup)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = model.profile_read()
    model.profile(0)
    assert torch.isfinite(x).all()

    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * args.steps / T_STEPS / dt
        fd = f_dense_per_step(MC, H, W, D)
        roof = None
        if prof.launches[0] > 0:
            ach = prof.flops[0] / (prof.ms[0] * 1e-3) / 1e12
            traffic = None
            try:    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH/WRITE_SIZE)
                traffic = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")))["dominant_traffic_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                pass
            wino = os.environ.get("S3D_WINO", "1") != "0"
            exec_frac = 4.0 / 9.0 if wino else 1.0        # Winograd F(2x2,3x3): 16 MFMA multiplies per 4 outputs instead of 36
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "mfma_executed_tflops": round(ach * exec_frac, 2),
                    "mfma_executed_frac": round(ach * exec_frac / PEAK_FP32_MFMA_TFLOPS, 4),
                    "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 2x read correction)",
                    # north_star also asks for the HBM-roofline fraction of the same kernel: PMC bytes / live launch time
                    "hbm_gbs": round(traffic / (prof.ms[0] / prof.launches[0] * 1e-3) / 1e9, 1) if traffic else None,
                    "hbm_frac_of_8TBs": round(traffic / (prof.ms[0] / prof.launches[0] * 1e-3) / 8e12, 4) if traffic else None,
                    "kernel": (("k_conv_wino2" if os.environ.get("S3D_WINO") == "2" else "k_conv_wino4") + " (fused Winograd F(2x2,3x3)" if wino else "k_conv_mfma<3x3> (direct") +
                              ", dense own-channel part of the rollout TriplaneConv)",
                    "avg_launch_us": round(prof.ms[0] / prof.launches[0] * 1e3, 2),
                    "launches_timed": int(prof.launches[0]),
                    "flops_per_launch_avg": prof.flops[0] / prof.launches[0],
                    "conv3x3_ms_per_step": round(prof.ms[0] / max(prof.forwards, 1), 4),
                    "rank1_ms_per_step": round(prof.ms[2] / max(prof.forwards, 1), 4),
                    "conv1x1_ms_per_step": round(prof.ms[1] / max(prof.forwards, 1), 4),
                    "note": "achieved = algorithmic flops of the launches (2*9*C*Cout per output pixel, own channels only: "
                            "rank-1 rollout exploited) / HIP-event time; mfma_executed_* = what the matrix cores really "
                            "multiply (x4/9 under Winograd); the reference-executed F_dense rate is effective_dense_tflops"}
        line = {"metric": "DDPM-1000 triplane samples/sec @128^2 latent", "value": value, "unit": "samples/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic (seeded random weights incl. zero-init convs; N(0,1) x_T; device RNG per step)",
                "config": {"workload": "BASELINE configs[1]: 128^2 triplane (H,W,D)=(128,128,128), 128-ch "
                                       "TriplaneUNetModelSmall, DDPM-1000, batch 1 per GPU; a step = 1 denoising step",
                           "steps_per_sample": T_STEPS, "batch_per_gpu": 1, "parallelism": f"{world} independent samples"},
                "f_dense_gflop_per_step": round(fd / 1e9, 2),
                "effective_dense_tflops": round(fd / (ms_step * 1e-3) / 1e12 * 1.0, 2),
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
