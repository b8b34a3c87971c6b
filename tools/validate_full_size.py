#!/usr/bin/env python3
"""One-off full-size parity run: N consecutive DDPM steps of BASELINE config 2 (128-ch UNet, 128^3) on the HIP path and on
the CPU port with identical noise, relative error (max|a-b| / max|b|) after every 10th step.  The HIP side is driven through the
public loop, p_sample_loop_progressive — one library call per step since round 4 (the output head applies the sampler update);
--single-step uses p_sample (forward + sampler kernel) instead.
    python tools/validate_full_size.py [--steps 100] [--single-step]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np, torch
import oracle as orc, torch_port as tp
from bench import usable_cores
from sin3dm_amd import testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--stride", type=int, default=10)
ap.add_argument("--single-step", action="store_true")
a = ap.parse_args()
torch.set_num_threads(usable_cores())
mc, (H, W, D) = 128, (128, 128, 128)
sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0)
model = TriplaneUNetModelSmall(12, mc, 12, use_scale_shift_norm=True); model.load_state_dict(sd); model.cuda().eval()
diffusion = create_gaussian_diffusion(steps=1000, predict_xstart=True)
tab, _ = orc.schedule_tables(None, 1000)
g = torch.Generator().manual_seed(3)
x_cpu = torch.randn((1, 12, H + D, W + D), generator=g); x_gpu = x_cpu.cuda()
# spread the steps over the whole schedule: every (1000 // steps)-th timestep would change the process; instead walk the
# FIRST `steps` steps (t = 999 ...), where the state is noise-dominated, and the LAST ones would need the whole chain
t0 = time.time()
pending = []
diffusion.noise_fn = lambda z: pending.pop(0).to(z.device)
loop = None if a.single_step else diffusion.p_sample_loop_progressive(model, tuple(x_gpu.shape), noise=x_gpu, model_kwargs=dict(H=H, W=W, D=D))
for k in range(a.steps):
    t = 999 - k
    eps = torch.randn(x_cpu.shape, generator=g)
    pending.append(eps)
    with torch.no_grad():
        if loop is None:
            x_gpu = diffusion.p_sample(model, x_gpu, torch.tensor([t], device="cuda"), model_kwargs=dict(H=H, W=W, D=D))["sample"]
        else:
            x_gpu = next(loop)["sample"]
        out = tp.unet_forward(sd, x_cpu, torch.tensor([float(t)]), H, W, D, mc)
        x_cpu, _ = tp.p_sample_update(out, x_cpu, eps, tab, t)
    if (k + 1) % a.stride == 0 or k == a.steps - 1:
        d = x_gpu.cpu() - x_cpu
        print(f"step {k + 1:4d} (t={t}): rel err {float(d.abs().max() / x_cpu.abs().max()):.3e}  rms {float(d.pow(2).mean().sqrt()):.3e}  [{time.time() - t0:.0f} s]", flush=True)
