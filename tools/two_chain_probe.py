#!/usr/bin/env python3
"""N independent batch-1 sample chains on N HIP streams of ONE process.

The reference makes several samples by batching them (src/sample.py:33-38); at batch 1 this build runs one strictly serial
chain of 42 launches per denoising step, 42 % of which are one-round latency-bound launches.  Independent samples need no
event edges between their chains, so this probe puts n chains on n streams (= hardware queues), issues their steps round-robin
from one host thread, and reports sample-steps/s against one chain — and against the same n samples as ONE batch-n chain, which
is what the reference would do.

    python tools/two_chain_probe.py [--chains 1 2 3 4] [--batches 2 4] [--steps 300] [--stagger 0.5] [--hwd 128 128 128] [--mc 128]
    GPU_MAX_HW_QUEUES=2 python tools/two_chain_probe.py ...        (the queue count is read by the HIP runtime at start-up)

Prints one JSON line per configuration.  `--trace-steps K` runs only K steps of --chains[0] chains (for rocprofv3 --kernel-trace;
tools/trace_overlap.py turns the trace into overlap figures).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, nargs="*", default=[1, 2, 3, 4])
    ap.add_argument("--batches", type=int, nargs="*", default=[2, 4])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--stagger", type=float, default=0.5, help="chain k starts k*stagger/n of a step late (a spin kernel on its stream)")
    ap.add_argument("--hwd", type=int, nargs=3, default=[128, 128, 128])
    ap.add_argument("--mc", type=int, default=128)
    ap.add_argument("--one-handle", action="store_true", help="chains share ONE model object via lanes (the product form)")
    ap.add_argument("--chains-batched", type=int, nargs="*", default=[], help="also run this many chains of each --batches size")
    ap.add_argument("--trace-steps", type=int, default=0)
    args = ap.parse_args()

    import torch
    from sin3dm_amd import testing as T
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall

    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    H, W, D = args.hwd
    kw = dict(H=H, W=W, D=D)
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=args.mc), 0)

    def new_model():
        m = TriplaneUNetModelSmall(12, args.mc, 12, num_res_blocks=1, channel_mult=(1, 2), use_scale_shift_norm=True)
        m.load_state_dict(sd)
        return m.to(dev).eval()

    models = []

    def model(i):
        while len(models) <= i:
            models.append(new_model())
        return models[i]

    def chain(i, batch):
        diffusion = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=True, timestep_respacing="")
        m = model(0 if args.one_handle else i)
        while True:
            for out in diffusion.p_sample_loop_progressive(m, (batch, 12, H + D, W + D), model_kwargs=kw):
                yield out

    import contextlib
    lane = (lambda i: model(0).lane(i)) if args.one_handle else (lambda i: contextlib.nullcontext())

    def spin(stream, us):
        if us <= 0:
            return
        with torch.cuda.stream(stream):
            torch.cuda._sleep(int(us * 1e-6 * 2.1e9))        # (cycles of the shader clock, roughly)

    def run(n, batch, steps, warmup, stagger_us=0.0):
        streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
        gens = [chain(i, batch) for i in range(n)]
        last = [None] * n
        with torch.no_grad():
            for _ in range(warmup):
                for i in range(n):
                    with torch.cuda.stream(streams[i]), lane(i):
                        last[i] = next(gens[i])
            torch.cuda.synchronize()
            for i in range(1, n):
                spin(streams[i], stagger_us * i)
            ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
            ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
            t0 = time.perf_counter()
            for i in range(n):
                ev0[i].record(streams[i])
            th0 = time.perf_counter()
            for _ in range(steps):
                for i in range(n):
                    with torch.cuda.stream(streams[i]), lane(i):
                        last[i] = next(gens[i])
            host_issue = time.perf_counter() - th0
            for i in range(n):
                ev1[i].record(streams[i])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        for i in range(n):
            assert torch.isfinite(last[i]["sample"]).all()
        per_chain = [ev0[i].elapsed_time(ev1[i]) / steps for i in range(n)]
        return {"chains": n, "batch_per_chain": batch, "steps_per_chain": steps, "wall_ms": round(dt * 1e3, 2),
                "sample_steps_per_s": round(n * batch * steps / dt, 1), "ms_per_sample_step": round(dt * 1e3 / (n * batch * steps), 4),
                "per_chain_step_latency_ms": [round(x, 4) for x in per_chain],
                "host_issue_ms_per_step": round(host_issue * 1e3 / (n * steps), 4),
                "samples_per_s_ddpm1000": round(n * batch * steps / dt / 1000, 4)}

    env = {k: v for k, v in os.environ.items() if k.startswith(("GPU_MAX_HW_QUEUES", "S3D_", "HIP_", "AMD_"))}
    if args.trace_steps:
        r = run(args.chains[0], 1, args.trace_steps, 20, 0.0)
        print(json.dumps(dict(r, env=env, trace=True)), flush=True)
        return 0
    base = None
    for n in args.chains:
        step_us = 850.0
        r = run(n, 1, args.steps, args.warmup, args.stagger * step_us / max(n, 1))
        if n == 1:
            base = r["sample_steps_per_s"]
        r["vs_one_chain"] = round(r["sample_steps_per_s"] / base, 4) if base else None
        print(json.dumps(dict(r, env=env)), flush=True)
    for b in args.batches:
        r = run(1, b, max(20, args.steps // b), max(10, args.warmup // b))
        r["vs_one_chain"] = round(r["sample_steps_per_s"] / base, 4) if base else None
        print(json.dumps(dict(r, env=env, form="one chain, batched (the reference's way, src/sample.py:33-38)")), flush=True)
        for n in args.chains_batched:
            r = run(n, b, max(20, args.steps // b), max(10, args.warmup // b), args.stagger * 850.0 * b / n)
            r["vs_one_chain"] = round(r["sample_steps_per_s"] / base, 4) if base else None
            print(json.dumps(dict(r, env=env, form=f"{n} chains of batch {b}")), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
