#!/usr/bin/env python3
"""Does a process that has run independent chains (extra HIP streams) launch more slowly afterwards?  (debug aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sin3dm_amd import testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
dev = torch.device("cuda:0")
def model(mc):
    m = TriplaneUNetModelSmall(12, mc, 12, channel_mult=(1, 2), use_scale_shift_norm=True)
    m.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0)); return m.to(dev)
diff = create_gaussian_diffusion(steps=1000, predict_xstart=True)
mt = model(64)
H, W, D, B = 92, 128, 92, 4
x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1).to(dev)
t = torch.tensor([700, 3, 250, 10], device=dev); w = torch.ones(B, device=dev); kw = dict(H=H, W=W, D=D)
def train_ms(n=40):
    for _ in range(5): diff.training_losses_and_grads(mt, x0, t, w, kw)
    torch.cuda.synchronize(); t0 = time.perf_counter(); th = 0.0
    for _ in range(n):
        a = time.perf_counter(); diff.training_losses_and_grads(mt, x0, t, w, kw); th += time.perf_counter() - a
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3
ms = model(32).eval()
def sample_ms(n=200):
    g = diff.p_sample_loop_progressive(ms, (1, 12, 40, 40), model_kwargs=dict(H=20, W=20, D=20))
    for _ in range(20): next(g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): next(g)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("before: train ms/step (wall, host issue)", train_ms(), " tiny sampling step (host-bound) ms", sample_ms())
if len(sys.argv) > 1 and sys.argv[1] == "streams":
    ss = [torch.cuda.Stream() for _ in range(3)]
    for s in ss:
        with torch.cuda.stream(s): torch.zeros(16, device=dev).add_(1)
    torch.cuda.synchronize(); what = "3 idle-after-use torch streams"
else:
    diff2 = create_gaussian_diffusion(steps=1000, predict_xstart=True, timestep_respacing="50")
    diff2.sample_loop_chains(ms, (1, 12, 40, 40), 6, chains=3, model_kwargs=dict(H=20, W=20, D=20)); torch.cuda.synchronize(); what = "sample_loop_chains(3 chains)"
print("after", what, ": train", train_ms(), " tiny sampling", sample_ms())
