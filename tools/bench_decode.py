#!/usr/bin/env python3
"""Secondary benchmark (BASELINE config 5 side): AutoEncoderGroupSkip.decode over a cell-centred grid.
Prints one JSON line: points/s, achieved TFLOP/s on the 1.182 MFLOP/point of the two MLPs."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sin3dm_amd import testing as T
from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip

ap = argparse.ArgumentParser()
ap.add_argument("--reso", type=int, default=256)
ap.add_argument("--hwd", type=int, nargs=3, default=(128, 128, 128))
ap.add_argument("--aabb-scale", type=float, nargs=3, default=(1, 1, 1))
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
net.load_state_dict(T.synthetic_state_dict(T.ae_param_shapes(), 5), strict=False)
net.to(dev).eval()
H, W, D = a.hwd
fm = [torch.from_numpy(0.8 * np.tanh(T.synthetic_noise(s, 40 + i))).to(dev) for i, s in enumerate(((1, 12, H, W), (1, 12, H, D), (1, 12, W, D)))]
aabb = torch.tensor([-s for s in a.aabb_scale] + list(a.aabb_scale), dtype=torch.float32)
torch.cuda.synchronize(); t0 = time.perf_counter()
net.prepare(fm); torch.cuda.synchronize(); t_prep = time.perf_counter() - t0
out = net.decode_grid(fm, a.reso, aabb=aabb); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    out = net.decode_grid(fm, a.reso, aabb=aabb)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
n = out.shape[0] * out.shape[1] * out.shape[2]
# iso-surface extraction of the decoded grid (marching cubes on the device, reading the sdf channel in place)
from sin3dm_amd.encoding.isosurface import marching_cubes
marching_cubes(out, 0.0, 1.0, n_attr=3); torch.cuda.synchronize()
t0 = time.perf_counter()
verts, tris, cols = marching_cubes(out, 0.0, 1.0, n_attr=3)
torch.cuda.synchronize(); t_mc = time.perf_counter() - t0
mc = {"seconds": t_mc, "vertices": int(verts.shape[0]), "triangles": int(tris.shape[0]),
      "grid_GBps": n * 16 * 3 / t_mc / 1e9, "note": "3 passes over the [X,Y,Z,4] grid (flags, vertices, triangles) incl. two scans and the count readback"}
if os.environ.get("MC_CPU_BASELINE"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle as orc
    sub = out[:128, :128, :128, 0].contiguous().cpu().numpy()
    c0 = time.perf_counter(); orc.marching_cubes(sub, 0.0, 1.0); c1 = time.perf_counter() - c0
    mc["cpu_port_128cube_s"] = c1
    mc["cpu_port_extrapolated_s"] = c1 * n / sub.size
flop_pt = 2.0 * 2 * (64 * 256 + 3 * 256 * 256 + 320 * 256) + 2.0 * 256 * 4
print(json.dumps({"metric": "decode_grid points/s", "value": n / dt, "grid": list(out.shape[:3]), "seconds": dt,
                  "prepare_first_call_s": t_prep, "tflops": n * flop_pt / dt / 1e12, "frac_of_157.3": n * flop_pt / dt / 157.3e12,
                  "finite": bool(torch.isfinite(out).all()), "marching_cubes": mc}))
