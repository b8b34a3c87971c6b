"""Composed-triplane layout and feat.npz I/O (reference: src/utils/triplane_util.py).

Composed map over the last two dims:  [[xy | xz], [yz^T | 0]]  with xy[H,W], xz[H,D], yz[W,D].
These are views/concats on torch tensors (plumbing); the kernels read the composed layout directly.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def compose_featmaps(feat_xy, feat_xz, feat_yz):
    """-> (composed [..., H+D, W+D], (H, W, D)); the DxD corner is zero (reference :7-17)."""
    H, W = feat_xy.shape[-2:]
    D = feat_xz.shape[-1]
    out = feat_xy.new_zeros(tuple(feat_xy.shape[:-2]) + (H + D, W + D))
    out[..., :H, :W] = feat_xy
    out[..., :H, W:] = feat_xz
    out[..., H:, :W] = feat_yz.transpose(-1, -2)
    return out, (H, W, D)


def decompose_featmaps(composed_map, sizes):
    """Views xy [..,H,W], xz [..,H,D], yz [..,W,D] of a composed map (reference :20-25)."""
    H, W, D = sizes
    return (composed_map[..., :H, :W], composed_map[..., :H, W:], composed_map[..., H:, :W].transpose(-1, -2))


def pad_composed_featmaps(composed_map, sizes, pad_sizes):
    """Zero-pad each plane; pad_sizes = [[H0,H1],[W0,W1],[D0,D1]] (reference :28-35)."""
    import torch.nn.functional as F
    xy, xz, yz = decompose_featmaps(composed_map, sizes)
    xy = F.pad(xy, list(pad_sizes[1]) + list(pad_sizes[0]))
    xz = F.pad(xz, list(pad_sizes[2]) + list(pad_sizes[0]))
    yz = F.pad(yz, list(pad_sizes[2]) + list(pad_sizes[1]))
    return compose_featmaps(xy, xz, yz)


def save_triplane_data(path, feat_xy, feat_xz, feat_yz):
    """npz with keys feat_xy / feat_xz / feat_yz (reference :38-41)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, feat_xy=feat_xy, feat_xz=feat_xz, feat_yz=feat_yz)


def load_triplane_data(path, device="cuda:0", compose=True):
    """Load a feat.npz; composed map + sizes, or the three planes (reference :44-61)."""
    data = np.load(path)
    planes = [torch.from_numpy(data[k][:]).float().to(device) for k in ("feat_xy", "feat_xz", "feat_yz")]
    if not compose:
        return tuple(planes)
    return compose_featmaps(*planes)


def get_data_iterator(featmaps_data, sizes, batch_size=1):
    """Endless iterator over the single training triplane expanded to a batch (reference :64-69)."""
    batch = featmaps_data.unsqueeze(0).expand(batch_size, -1, -1, -1)
    H, W, D = sizes
    while True:
        yield batch, {"H": H, "W": W, "D": D}
