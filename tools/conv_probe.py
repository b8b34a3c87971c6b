#!/usr/bin/env python3
"""Yardstick: the vendor library's fp32 3x3 convolution (torch.nn.functional.conv2d -> MIOpen, benchmark mode = MIOpen's find)
on the layer shapes of the scored step (three planes as a batch of 3; plain 'same' convolution, no rollout terms, no bias),
next to this build's per-launch times from profiles/<round>_timeline.txt.  python tools/conv_probe.py"""
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")


def bench(f, n=30):
    for _ in range(8):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, hw, ours in ((128, 128, 128, "53-56"), (256, 256, 64, "50-52"), (384, 128, 128, "128"), (128, 256, 64, "29.5")):
    for fmt in (torch.contiguous_format, torch.channels_last):
        x = torch.randn(3, cin, hw, hw, device=dev).to(memory_format=fmt)
        w = torch.randn(cout, cin, 3, 3, device=dev).to(memory_format=fmt)
        us = bench(lambda: F.conv2d(x, w, padding=1))
        flops = 2.0 * 3 * hw * hw * cin * cout * 9
        print(f"{cin:4d} -> {cout:4d} @ {hw}^2 x 3 planes  {'NCHW' if fmt == torch.contiguous_format else 'NHWC'}  {us:8.1f} us  direct-equivalent {flops / us / 1e6:6.1f} TFLOP/s   (this build: {ours} us)")
