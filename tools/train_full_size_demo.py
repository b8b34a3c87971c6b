#!/usr/bin/env python3
"""Full-size training demo on synthetic data (one MI355X): a procedural textured shape -> auto-encoder stage at 128^3
feature maps with 65 536 points per iteration -> diffusion stage on the resulting latent (64-ch UNet, batch 4) -> a few
DDIM samples decoded to meshes.  Prints the loss curves and wall times; not a benchmark line, a does-it-train check at
BASELINE config 4's sizes with short schedules.
    python tools/train_full_size_demo.py [--enc-iters 400] [--diff-iters 400] [--out /tmp/s3d_demo]"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from sin3dm_amd import sample, train

ap = argparse.ArgumentParser()
ap.add_argument("--enc-iters", type=int, default=400)
ap.add_argument("--diff-iters", type=int, default=400)
ap.add_argument("--out", default="/tmp/s3d_demo")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
# a 256^3 volume of a bumpy ellipsoid with a procedural colour field, in the preprocessed-shape format of the reference
R = 256
ext = np.asarray([1.0, 1.0, 1.0], np.float32)
ax = [np.linspace(-1, 1, R, dtype=np.float32)] * 3
g = np.stack(np.meshgrid(*ax, indexing="ij"), -1)
def sdf_f(p):
    r = np.linalg.norm(p * np.asarray([1.0, 1.3, 1.6], np.float32), axis=-1)
    return ((r - 0.75) + 0.05 * np.sin(9 * p[..., 0]) * np.sin(7 * p[..., 1]) * np.sin(8 * p[..., 2])).astype(np.float32) * 0.5
def tex_f(p):
    return (0.5 + 0.5 * np.sin(4 * p + np.asarray([0.0, 2.0, 4.0], np.float32))).astype(np.float32)
rng = np.random.Generator(np.random.PCG64(0))
sd = sdf_f(g)
near_idx = np.argwhere(np.abs(sd) < 0.03)
near = (near_idx[rng.integers(0, len(near_idx), 400000)] / (R - 1) * 2 - 1).astype(np.float32) + rng.normal(0, 0.004, (400000, 3)).astype(np.float32)
data = os.path.join(a.out, "shape.npz")
np.savez(data, aabb=np.concatenate([-ext, ext]), threshold=0.05, pts_grid=g, sdf_grid=sd, tex_grid=tex_f(g), pts_near_surf=near,
         sdf_near_surf=sdf_f(near), tex_near_surf=tex_f(near), pts_on_surf=near[:20000], tex_on_surf=tex_f(near[:20000]))
tag = os.path.join(a.out, "run")
t0 = time.time()
train.main(["--tag", tag, "--data_path", data, "--fm_reso", "128", "--enc_n_iters", str(a.enc_iters), "--enc_batch_size", "65536",
            "--model_channels", "64", "--diff_batch_size", "4", "--diff_n_iters", str(a.diff_iters), "--save_interval", "1000000",
            "--log_interval", "50"], confirm=lambda _: "y")
torch.cuda.synchronize(); t_train = time.time() - t0
enc = [json.loads(l) for l in open(os.path.join(tag, "encoding", "progress.jsonl"))]
dif = [json.loads(l) for l in open(os.path.join(tag, "diffusion", "progress.jsonl"))]
print("auto-encoder  (step, sdf_loss, tex_loss):", [(e["step"], round(e["sdf_loss"], 4), round(e["tex_loss"], 4)) for e in enc])
print("eval:", json.load(open(os.path.join(tag, "encoding", "eval_stat.json"))))
print("diffusion     (step, loss, grad_norm):", [(d["step"], round(d.get("loss", float("nan")), 4), round(d.get("grad_norm", float("nan")), 3)) for d in dif])
t0 = time.time()
paths = sample.main(["--tag", tag, "--n_samples", "2", "--use_ddim", "True", "--timestep_respacing", "50", "--reso", "128"])
torch.cuda.synchronize()
print(f"train wall {t_train:.1f} s (incl. data generation/IO), sample+mesh wall {time.time() - t0:.1f} s ->", [os.path.relpath(p, a.out) for p in paths])
