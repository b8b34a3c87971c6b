#!/usr/bin/env python3
"""Step time of the AUTO-ENCODER training iteration (BASELINE config 4, first stage): 128^3 feature maps from a
256^3 x 4 input volume, 65 536 points per step, default network (4+8 latent channels, 64-wide planes, 256-wide MLPs).
Not the scored bench line.  python tools/bench_ae_train.py [--fm 128 128 128] [--points 65536] [--steps 20] [--cpu-baseline]"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from bench import usable_cores
from sin3dm_amd import _lib, testing as T
from sin3dm_amd.encoding.model import FlatGroupAdamW
from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip

ap = argparse.ArgumentParser()
ap.add_argument("--fm", type=int, nargs=3, default=(128, 128, 128))
ap.add_argument("--points", type=int, default=65536)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--cpu-baseline", action="store_true")
ap.add_argument("--eager-gpu-baseline", action="store_true")
args = ap.parse_args()
H, W, D = args.fm
N = args.points
dev = torch.device("cuda:0")
sd = T.synthetic_state_dict(T.ae_param_shapes(with_encoder=True), 5)
net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
net.load_state_dict(sd, strict=False)
net.to(dev)
g = torch.Generator(device=dev).manual_seed(0)
vol = torch.rand((1, 4, 2 * H, 2 * W, 2 * D), device=dev, generator=g)
vol[:, :1] = vol[:, :1] * 0.2 - 0.1
t0 = time.perf_counter(); net.set_volume(vol); torch.cuda.synchronize(); t_proj = time.perf_counter() - t0
cfg = _lib.AeLossCfg(1, 0, 0.05, 0.999, 1.0)
opt = FlatGroupAdamW(net, 5e-3, 0.2, lr_decay=0.1 ** (1 / 25000))
pts = torch.rand((N, 3), device=dev, generator=g) * 2 - 1
sdf = (torch.rand((N, 1), device=dev, generator=g) * 0.1 - 0.05)
tex = torch.rand((N, 3), device=dev, generator=g)
grad = torch.empty_like(net.flat_parameters)

def step():
    losses, _, gr = net.loss_and_grads(vol, pts, sdf, tex, cfg, grad_out=grad)
    opt.step(gr)
    return losses

for _ in range(args.warmup):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.steps):
    l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
assert torch.isfinite(l).all() and torch.isfinite(net.flat_parameters).all()
hw = H * W + H * D + W * D
mlp = 2 * (64 * 256 + 2 * 256 * 256 + 320 * 256 + 256 * 256) * 2 + 2 * 256 * 4           # both MLPs, flops per point
fwd = mlp * N + 2 * 25 * (32 * 64 + 64 * 64) * hw * 2                                         # as executed here (cin padded to 32)
ref_fwd = mlp * N + 2 * 25 * ((4 + 8) / 2 * 64 + 64 * 64) * hw * 2 + 2 * 64 * (4 + 16 * 2) * H * W * D * 4   # reference: real cin + Conv3d
line = {"what": "auto-encoder training iteration", "config": f"feature maps ({H},{W},{D}), {N} points/step, default AutoEncoderGroupSkip",
        "ms_per_step": round(dt * 1e3, 3), "projection_once_ms": round(t_proj * 1e3, 1),
        "effective_tflops": round(3 * ref_fwd / dt / 1e12, 1),
        "note": "effective = 3 x the reference's forward flops (incl. its per-step Conv3d, which is a one-off projection here) / time"}
if args.cpu_baseline:
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import torch_port as tp
    torch.set_num_threads(usable_cores())
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cv, cp, cs, ct = vol.cpu(), pts.cpu(), sdf.cpu(), tex.cpu()
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1])
    n, c0 = 0, None
    while True:
        ls = tp.ae_losses(tp.ae_decode(p, cp, tp.ae_encode(p, cv), aabb), cs, ct, 0.05)
        sum(ls.values()).backward()
        if c0 is None:
            c0 = time.perf_counter(); continue
        n += 1
        if time.perf_counter() - c0 > 20 or n >= 5:
            break
    cdt = (time.perf_counter() - c0) / n
    line["cpu_baseline"] = {"ms_per_step": round(cdt * 1e3, 1), "cores": usable_cores(), "kind": "port",
                            "sample": f"{n} forward+backward iterations of oracle/torch_port.py (PyTorch-CPU autograd, incl. the Conv3d encoder)"}
    line["gpu_over_cpu"] = round(cdt / dt, 1)
if args.eager_gpu_baseline:
    # the same port under PyTorch-ROCm autograd with its tensors on this GPU (Conv3d encoder, 5x5 convolutions, grid_sample, MLPs
    # through MIOpen / rocBLAS, fp32, eager) + torch.optim.AdamW: the reference's iteration (src/encoding/model.py) on this box
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import torch_port as tp
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.benchmark = True
    p = {k: v.clone().to(dev).requires_grad_(True) for k, v in sd.items()}
    opt_t = torch.optim.AdamW(list(p.values()), lr=5e-3, weight_decay=0.2)
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1], device=dev)
    n, c0 = 0, None
    while True:
        ls = tp.ae_losses(tp.ae_decode(p, pts, tp.ae_encode(p, vol), aabb), sdf, tex, 0.05)
        opt_t.zero_grad(set_to_none=True)
        sum(ls.values()).backward()
        opt_t.step()
        n += 1
        if n == 3:
            torch.cuda.synchronize(); c0 = time.perf_counter()
        if n > 3 and n % 5 == 3:
            torch.cuda.synchronize()
            if time.perf_counter() - c0 > 10 or n >= 43:
                break
    edt = (time.perf_counter() - c0) / (n - 3)
    line["eager_gpu_baseline"] = {"ms_per_step": round(edt * 1e3, 2), "kind": "port",
                                  "sample": f"{n - 3} forward+backward+AdamW iterations of oracle/torch_port.py under PyTorch-ROCm autograd on cuda:0 (MIOpen find mode, fp32, incl. the per-step Conv3d encoder)"}
    line["over_eager_gpu"] = round(edt / dt, 2)
print(json.dumps(line), flush=True)
