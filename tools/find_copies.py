#!/usr/bin/env python3
"""Which torch ops / memcpys ride along with a sampling-loop step?  (kernel traces show ~1 __amd_rocclr_copyBuffer per step)
    python tools/find_copies.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from sin3dm_amd import testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
dev = torch.device("cuda:0")
mc, (H, W, D) = 128, (128, 128, 128)
model = TriplaneUNetModelSmall(12, mc, 12, use_scale_shift_norm=True)
model.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0))
model.to(dev).eval()
diff = create_gaussian_diffusion(steps=1000, predict_xstart=True)
gen = diff.p_sample_loop_progressive(model, (1, 12, H + D, W + D), model_kwargs=dict(H=H, W=W, D=D))
for _ in range(20):
    next(gen)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        next(gen)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
