#!/usr/bin/env python3
"""End-to-end wall time of the demo workload the reference quotes (src/app.py:12: "30-50 s" for 4 samples on an A6000
with default settings: 64-ch UNet, DDPM-1000, marching cubes at 256, texture 2048): sampling + decode + iso-surface for
4 samples on one MI355X.  NOT like for like — the reference's figure includes xatlas UV unwrapping and nvdiffrast texture
baking, which are out of scope here (vertex colours instead), and uses trained weights (synthetic here: same arithmetic)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sin3dm_amd import testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
from sin3dm_amd.encoding.isosurface import largest_component, marching_cubes
from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip
from sin3dm_amd.utils.triplane_util import decompose_featmaps

dev = torch.device("cuda:0")
mc, (H, W, D), n_samples, reso = 64, (128, 128, 128), 4, 256
model = TriplaneUNetModelSmall(12, mc, 12, use_scale_shift_norm=True)
model.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0)); model.to(dev).eval()
net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
net.load_state_dict(T.synthetic_state_dict(T.ae_param_shapes(), 5), strict=False); net.to(dev).eval()
diffusion = create_gaussian_diffusion(steps=1000, predict_xstart=True)
aabb = torch.tensor([-1., -1, -1, 1, 1, 1])

def run():
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = diffusion.p_sample_loop(model, (n_samples, 12, H + D, W + D), model_kwargs=dict(H=H, W=W, D=D), device=dev)
    torch.cuda.synchronize(); t["sampling_s"] = time.perf_counter() - t0
    t0 = time.perf_counter(); nv = nt = 0
    for i in range(n_samples):
        fm = [f.contiguous() for f in decompose_featmaps(x[i:i + 1], (H, W, D))]
        grid = net.decode_grid(fm, reso, aabb=aabb)
        v, f, c = marching_cubes(grid, 0.0, 1.0, n_attr=3)
        v, f, c = largest_component(v, f, c)
        nv += len(v); nt += len(f)
    torch.cuda.synchronize(); t["decode_and_mesh_s"] = time.perf_counter() - t0
    t["total_s"] = t["sampling_s"] + t["decode_and_mesh_s"]; t["vertices"] = nv; t["triangles"] = nt
    return t

run()                      # warm-up (allocations, first-call packing)
print(json.dumps({"what": "4 samples end to end (batch 4 DDPM-1000 at 128^3, 64-ch) + 256^3 decode + iso-surface + largest component",
                  **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in run().items()},
                  "reference_quote": "30-50 s on an A6000 incl. UV atlas + texture baking (src/app.py:12)"}))
