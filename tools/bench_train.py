#!/usr/bin/env python3
"""Step time of the diffusion TRAINING step (BASELINE config 4: 64-ch UNet on the towerruins triplane (92,128,92),
batch 4 per GPU) — not the scored bench line (bench.py measures sampling).

    python tools/bench_train.py [--mc 64] [--hwd 92 128 92] [--batch 4] [--steps 20] [--cpu-baseline]
    python tools/bench_train.py --gpus N ...                                         # data parallel: N fresh ranks (sin3dm_amd/launcher.py)
    python -m torch.distributed.run --nproc-per-node N tools/bench_train.py --gpus N ...   # the same under torchrun

A step = timestep draw + q_sample + UNet forward + per-plane MSE + UNet backward (+ gradient all-reduce) + fused
AdamW/EMA + device-side weight repack.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    # --gpus N without a torchrun environment: this process touches no GPU, starts N fresh ranks and relays rank 0's line
    _pre = argparse.ArgumentParser(add_help=False)
    _pre.add_argument("--gpus", type=int, default=1)
    _n = _pre.parse_known_args()[0].gpus
    if _n > 1:
        from sin3dm_amd.launcher import spawn_ranks
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], _n))

import torch
import torch.distributed as dist

from bench import f_dense_per_step, usable_cores
from sin3dm_amd import parallel, testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.train_util import TrainLoop
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall

ap = argparse.ArgumentParser()
ap.add_argument("--mc", type=int, default=64)
ap.add_argument("--hwd", type=int, nargs=3, default=(92, 128, 92))
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--cpu-baseline", action="store_true")
ap.add_argument("--eager-gpu-baseline", action="store_true")
ap.add_argument("--gpus", type=int, default=None, help="ranks (default: WORLD_SIZE under torchrun, else 1)")
ap.add_argument("--overlap", type=int, default=-1, choices=(-1, 0, 1),
                help="1 / 0: gradient all-reduce cut into groups overlapped with the backward pass / one all-reduce after it "
                     "(sets S3D_OVERLAP_ALLREDUCE for TrainLoop); -1: whatever the environment says (TrainLoop's default is off)")
args = ap.parse_args()

rank, local, world = parallel.env_rank_world()
if args.gpus is None:
    args.gpus = world                    # `torchrun ... tools/bench_train.py` without --gpus: the launcher's world size
if args.overlap >= 0:
    os.environ["S3D_OVERLAP_ALLREDUCE"] = str(args.overlap)
if world != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
if local >= torch.cuda.device_count():
    raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but this node shows {torch.cuda.device_count()} GPU(s)")
torch.cuda.set_device(local)
dev = torch.device(f"cuda:{local}")
parallel.init(device=dev)
H, W, D = args.hwd
sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=args.mc), 0)
model = TriplaneUNetModelSmall(12, args.mc, 12, use_scale_shift_norm=True)
model.load_state_dict(sd)
model.to(dev)
diffusion = create_gaussian_diffusion(steps=1000, predict_xstart=True)
x0 = torch.from_numpy(T.synthetic_noise((12, H + D, W + D), 400)).clamp(-1, 1).to(dev)


def data():
    batch = x0.unsqueeze(0).expand(args.batch, -1, -1, -1)
    while True:
        yield batch, {"H": H, "W": W, "D": D}


loop = TrainLoop(model=model, diffusion=diffusion, data=data(), batch_size=args.batch, microbatch=-1, lr=5e-4, ema_rate=0.9999,
                 log_interval=10 ** 9, save_interval=10 ** 9, resume_checkpoint=False, lr_anneal_steps=25000)
it = data()
for _ in range(args.warmup):
    loop.run_step(*next(it)); loop.step += 1
torch.cuda.synchronize(); parallel.barrier()
t0 = time.perf_counter()
for _ in range(args.steps):
    loop.run_step(*next(it)); loop.step += 1
torch.cuda.synchronize(); parallel.barrier()
dt = parallel.max_over_ranks(time.perf_counter() - t0, dev) / args.steps
assert torch.isfinite(model.flat_parameters).all()
if rank == 0:
    fwd = f_dense_per_step(args.mc, H, W, D) * args.batch
    line = {"what": "diffusion training step", "config": f"{args.mc}-ch UNet, (H,W,D)=({H},{W},{D}), batch {args.batch}/GPU, {world} GPU(s)",
            "ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(args.batch * world / dt, 2),
            "overlap_allreduce": bool(loop.overlap_allreduce) if world > 1 else None,
            "effective_dense_tflops_per_gpu": round(3 * fwd / dt / 1e12, 1),
            "note": "effective = 3 x F_dense(forward) per step (fwd + dgrad + wgrad as the reference executes them)"}
    if args.cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import oracle as orc
        import torch_port as tp
        torch.set_num_threads(usable_cores())
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        tabs = orc.schedule_tables_named(1000)
        xb = x0.cpu().unsqueeze(0).expand(args.batch, -1, -1, -1)
        n, c0 = 0, None
        while True:
            t = torch.randint(0, 1000, (args.batch,))
            terms, _ = tp.training_losses(p, xb, t, torch.randn_like(xb), tabs, H, W, D, model_channels=args.mc)
            terms["loss"].mean().backward()
            if c0 is None:
                c0 = time.perf_counter()        # first iteration = warm-up
                continue
            n += 1
            if time.perf_counter() - c0 > 20 or n >= 10:
                break
        cdt = (time.perf_counter() - c0) / n
        line["cpu_baseline"] = {"ms_per_step": round(cdt * 1e3, 1), "cores": usable_cores(), "kind": "port",
                                "sample": f"{n} forward+backward steps of oracle/torch_port.py (PyTorch-CPU autograd)"}
        line["gpu_over_cpu"] = round(cdt / dt, 1)
    if args.eager_gpu_baseline and world == 1:
        # the same port under PyTorch-ROCm autograd with its tensors on this GPU (MIOpen / rocBLAS, fp32, eager) + torch.optim.AdamW
        # + the EMA update: what the reference's run_step (train_util.py:163-247) costs on this box without this library
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import oracle as orc
        import torch_port as tp
        torch.backends.cudnn.allow_tf32 = False
        torch.backends.cuda.matmul.allow_tf32 = False
        torch.backends.cudnn.benchmark = True
        p = {k: v.clone().to(dev).requires_grad_(True) for k, v in sd.items()}
        ema = [v.detach().clone() for v in p.values()]
        opt_t = torch.optim.AdamW(list(p.values()), lr=5e-4, weight_decay=0.0)
        tabs = orc.schedule_tables_named(1000)
        xb = x0.to(dev).unsqueeze(0).expand(args.batch, -1, -1, -1).contiguous()
        n, c0 = 0, None
        while True:
            t = torch.randint(0, 1000, (args.batch,), device=dev)
            terms, _ = tp.training_losses(p, xb, t, torch.randn_like(xb), tabs, H, W, D, model_channels=args.mc)
            opt_t.zero_grad(set_to_none=True)
            terms["loss"].mean().backward()
            opt_t.step()
            with torch.no_grad():
                torch._foreach_mul_(ema, 0.9999)
                torch._foreach_add_(ema, [v.detach() for v in p.values()], alpha=1 - 0.9999)
            n += 1
            if n == 3:
                torch.cuda.synchronize(); c0 = time.perf_counter()    # three warm-up steps (MIOpen's find)
            if n > 3 and n % 5 == 3:
                torch.cuda.synchronize()
                if time.perf_counter() - c0 > 10 or n >= 43:
                    break
        edt = (time.perf_counter() - c0) / (n - 3)
        line["eager_gpu_baseline"] = {"ms_per_step": round(edt * 1e3, 2), "kind": "port",
                                      "sample": f"{n - 3} forward+backward+AdamW+EMA steps of oracle/torch_port.py under PyTorch-ROCm autograd on cuda:0 (MIOpen find mode, fp32)"}
        line["over_eager_gpu"] = round(edt / dt, 2)
    print(json.dumps(line), flush=True)
if world > 1:
    dist.destroy_process_group()
