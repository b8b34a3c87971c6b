"""Device selection helpers (reference: src/utils/dist_util.py — single process per GPU)."""
from __future__ import annotations

import io
import os

import torch as th

_gpu_id = 0


def setup_dist(device=0):
    """Select the GPU of this process (reference :19-27 sets CUDA_VISIBLE_DEVICES; one process per GPU)."""
    global _gpu_id
    _gpu_id = int(device)
    if th.cuda.is_available():
        th.cuda.set_device(_gpu_id)


def dev():
    """The MI355X of this process.  There is no CPU fallback on this path (reference :45-52)."""
    if not th.cuda.is_available():
        raise RuntimeError("sin3dm_amd needs a visible MI355X; there is no CPU fallback")
    return th.device(f"cuda:{_gpu_id}")


def load_state_dict(path, **kwargs):
    """torch.load of a checkpoint written by the reference (reference :55-59)."""
    with open(path, "rb") as f:
        data = f.read()
    return th.load(io.BytesIO(data), **kwargs)
