#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP auto-encoder tier against the CPU oracle (oracle/torch_port.py autograd)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import torch_port as tp
from sin3dm_amd import testing as T, _lib
from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip
H, W, D, N = 12, 16, 10, 192
thr = 0.05
shapes = T.ae_param_shapes(with_encoder=True)
sd = {k: v.requires_grad_(True) for k, v in T.synthetic_state_dict(shapes, 5).items()}
vol = torch.tanh(torch.from_numpy(T.synthetic_noise((1, 4, 2 * H, 2 * W, 2 * D), 1200))); vol[:, 1:] = 0.5 * vol[:, 1:] + 0.5
aabb = torch.tensor([-0.7, -1.0, -0.45, 0.7, 1.0, 0.45])
rng = np.random.Generator(np.random.PCG64(int(os.environ.get('SEED', '1301')))); ext = np.asarray([0.7, 1.0, 0.45], np.float32)
p = torch.from_numpy(rng.uniform(-1.1, 1.1, size=(N, 3)).astype(np.float32) * ext)
s = torch.from_numpy(np.clip(rng.normal(0, 0.04, size=(N, 1)), -thr, thr).astype(np.float32))
c = torch.from_numpy(rng.uniform(0, 1, size=(N, 3)).astype(np.float32))
ls = tp.ae_losses(tp.ae_decode(sd, p, tp.ae_encode(sd, vol), aabb), s, c, thr)
sum(ls.values()).backward()
net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
net.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=False)
net.cuda(); net.reset_aabb(aabb.cuda())
losses, _, g = net.loss_and_grads(vol.cuda(), p.cuda(), s.cuda(), c.cuda(), _lib.AeLossCfg(1, 0, thr, 0.999, 1.0))
print("losses", losses.tolist(), {k: float(v) for k, v in ls.items()})
gmax = max(float(v.grad.norm()) for v in sd.values())
rows = []
for name, view in net.split_flat(g).items():
    ref = sd[name].grad
    rows.append((float((view.cpu() - ref).norm()) / max(float(ref.norm()), 1e-3 * gmax), name, float(view.norm()), float(ref.norm())))
for r in sorted(rows, reverse=True)[:25]:
    print("%.3e  %-40s hip %.4e ref %.4e" % r)
for nm in ("geo_decoder.second_layers.2.bias", "tex_decoder.second_layers.2.bias"):
    v = net.split_flat(g)[nm].cpu(); r = sd[nm].grad
    e = (v - r).abs()
    print(nm, "max abs err", float(e.max()), "ref max", float(r.abs().max()), "n(err>1e-6)", int((e > 1e-6).sum()), "of", e.numel())
    idx = torch.argsort(e, descending=True)[:6]
    print("   idx", idx.tolist(), "err", e[idx].tolist(), "hip", v[idx].tolist(), "ref", r[idx].tolist())
