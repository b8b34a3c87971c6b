#!/usr/bin/env python3
"""Does replaying a captured graph of one denoising step beat stream launches?  
The whole step — UNet forward (~55 library launches on torch's current stream) + torch.randn_like + the fused sampler
update — is captured with torch.cuda.CUDAGraph (hipGraph underneath) on static buffers and replayed.
    python tools/graph_experiment.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sin3dm_amd import testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall

dev = torch.device("cuda:0")
for name, mc, (H, W, D), B, steps in (("C1 towerruins@64 64-ch B=1", 64, (46, 64, 46), 1, 300), ("64-ch 128^3 B=1", 64, (128, 128, 128), 1, 300),
                                      ("C2 128-ch 128^3 B=1", 128, (128, 128, 128), 1, 300)):
    model = TriplaneUNetModelSmall(12, mc, 12, use_scale_shift_norm=True)
    model.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0))
    model.to(dev).eval()
    diff = create_gaussian_diffusion(steps=1000, predict_xstart=True)
    kw = dict(H=H, W=W, D=D)
    x = torch.randn(B, 12, H + D, W + D, device=dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.int64)
    with torch.no_grad():
        for _ in range(20):
            x = diff.p_sample(model, x, t, model_kwargs=kw)["sample"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            x = diff.p_sample(model, x, t, model_kwargs=kw)["sample"]
        torch.cuda.synchronize()
        stream_ms = (time.perf_counter() - t0) / steps * 1e3
        xs = x.clone()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                y = diff.p_sample(model, xs, t, model_kwargs=kw)["sample"]
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            ys = diff.p_sample(model, xs, t, model_kwargs=kw)["sample"]
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"config": name, "stream_ms_per_step": round(stream_ms, 4), "graph_replay_ms_per_step": round(graph_ms, 4),
                      "finite": bool(torch.isfinite(ys).all())}), flush=True)
