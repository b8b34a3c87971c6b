#!/usr/bin/env python3
"""BASELINE configs[4] end to end on one MI355X: the retargeted (256,256,128) triplane (`--resize 2 2 1` from 128^3) — DDPM-1000
with the 128-ch UNet, batch 1 — then the decoder over the `_resize_aabb` grid at `--reso 512` (aabb scaled (2,2,1): 512 x 512 x 256
= 67.1 M points; src/encoding/model.py:335-360, src/encoding/utils3d.py:13-25), iso-surface extraction on the device and the
largest component.  One JSON line: seconds per stage, decode TFLOP/s on the 1.182 MFLOP/point of the two MLPs
and its fraction of the 157.3-TF fp32 MFMA peak.  Synthetic weights (no checkpoint offline): same arithmetic.

    python tools/bench_config5.py [--steps 1000] [--mc 128]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sin3dm_amd import testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
from sin3dm_amd.encoding.isosurface import largest_component, marching_cubes
from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip
from sin3dm_amd.utils.triplane_util import decompose_featmaps

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1000, help="denoising steps of the sample (1000 = the config; fewer: a quick look)")
ap.add_argument("--mc", type=int, default=128)
ap.add_argument("--reso", type=int, default=512)
a = ap.parse_args()
dev = torch.device("cuda:0")
H, W, D = 256, 256, 128
resize = (2.0, 2.0, 1.0)
model = TriplaneUNetModelSmall(12, a.mc, 12, use_scale_shift_norm=True)
model.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=a.mc), 0)); model.to(dev).eval()
net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
net.load_state_dict(T.synthetic_state_dict(T.ae_param_shapes(), 5), strict=False); net.to(dev).eval()
diffusion = create_gaussian_diffusion(steps=1000, predict_xstart=True, timestep_respacing="" if a.steps == 1000 else str(a.steps))
base = torch.tensor([-1., -1, -1, 1, 1, 1])
# ShapeAutoEncoder._resize_aabb (src/encoding/model.py:351-360): the box grows with the feature map
aabb = torch.cat([base[:3] * torch.tensor(resize), base[3:] * torch.tensor(resize)])
flop_pt = 2.0 * 2 * (64 * 256 + 3 * 256 * 256 + 320 * 256) + 2.0 * 256 * 4


def run():
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = diffusion.p_sample_loop(model, (1, 12, H + D, W + D), model_kwargs=dict(H=H, W=W, D=D), device=dev)
    torch.cuda.synchronize(); t["sampling_s"] = time.perf_counter() - t0
    fm = [f.contiguous() for f in decompose_featmaps(x, (H, W, D))]
    t0 = time.perf_counter()
    grid = net.decode_grid(fm, a.reso, aabb=aabb)
    torch.cuda.synchronize(); t["decode_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    v, f, c = marching_cubes(grid, 0.0, 1.0, n_attr=3)
    torch.cuda.synchronize(); t["isosurface_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    v, f, c = largest_component(v, f, c)
    torch.cuda.synchronize(); t["largest_component_s"] = time.perf_counter() - t0
    n = grid.shape[0] * grid.shape[1] * grid.shape[2]
    t.update(grid=list(grid.shape[:3]), points=n, decode_tflops=n * flop_pt / t["decode_s"] / 1e12,
             decode_frac_of_157_3=n * flop_pt / t["decode_s"] / 157.3e12, vertices=int(v.shape[0]), triangles=int(f.shape[0]),
             total_s=t["sampling_s"] + t["decode_s"] + t["isosurface_s"] + t["largest_component_s"],
             sampling_ms_per_step=t["sampling_s"] / diffusion.num_timesteps * 1e3, finite=bool(torch.isfinite(grid).all()))
    return t


run()                      # warm-up (allocations, first-call packing, clocks)
r = run()
print(json.dumps({"what": f"BASELINE configs[4] end to end: DDPM-{diffusion.num_timesteps} at (256,256,128), {a.mc}-ch UNet, batch 1 -> decode_grid reso {a.reso} "
                          "with the aabb scaled (2,2,1) -> iso-surface -> largest component; one MI355X, synthetic weights",
                  **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}}))
