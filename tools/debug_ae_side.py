#!/usr/bin/env python3
"""Which auto-encoder gradients differ between BWD_SIDE = 0 and the side stream (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from sin3dm_amd import _lib
import test_hip_ae as ta
from conftest import golden
g = golden("ae_train")
H, W, D, N = (int(v) for v in g["hwdn"])
thr = float(g["thr"])
vol = ta._volume(H, W, D)
def run(mode, reps=4):
    _lib.set_option("BWD_SIDE", mode)
    net = ta._net()
    net.reset_aabb(torch.from_numpy(g["aabb"]).cuda())
    pts, sdf, tex = (torch.from_numpy(g[k]).cuda() for k in ("pts", "sdf", "tex"))
    outs = []
    for _ in range(reps):
        losses, _, grads = net.loss_and_grads(vol, pts, sdf, tex, ta._loss_cfg(thr))
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in net.split_flat(grads).items()})
    return outs
a = run(0)
b = run(None)
for i in range(len(a)):
    bad = [(k, float((a[0][k] - b[i][k]).abs().max()), float(a[0][k].abs().max())) for k in a[0] if not torch.equal(a[0][k], b[i][k])]
    bad_a = [k for k in a[0] if not torch.equal(a[0][k], a[i][k])]
    print("rep", i, "inline self-diff:", bad_a, "| side vs inline:", bad)
