"""Decode-side of ShapeAutoEncoder (reference: src/encoding/model.py:141-176, 319-360, 475-488)."""
from __future__ import annotations

import os

import numpy as np
import torch

from .networks import get_networks


class ShapeAutoEncoder:
    """Loads `ckpt_final.pth` written by the reference and decodes triplanes on MI355X.
    Training, encode and mesh/texture export (PyMCubes, xatlas, nvdiffrast) are out of scope."""

    def __init__(self, log_dir, cfg, device=None):
        self.log_dir = log_dir
        self.model_dir = os.path.join(log_dir, "model")
        self.device = device or torch.device(f"cuda:{getattr(cfg, 'gpu_id', 0)}")
        self.net = get_networks(cfg).to(self.device)
        self.aabb = self.net.aabb.clone()
        self.featmap_size = None

    def load_ckpt(self, name=None):
        """Reference :158-176 — dict {net, optimizer, scheduler, Ka, Kd, Ks, Ns, aabb, featmap_size}."""
        path = name if name and os.path.isabs(str(name)) else os.path.join(self.model_dir, f"ckpt_{name}.pth")
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        self.net.load_state_dict(ckpt["net"])
        self.aabb = torch.as_tensor(ckpt["aabb"], dtype=torch.float32).to(self.device)
        self.net.reset_aabb(self.aabb)
        self.featmap_size = tuple(ckpt["featmap_size"])
        self.material = {k: ckpt.get(k) for k in ("Ka", "Kd", "Ks", "Ns")}

    def _resize_aabb(self, featmap_size):
        """aabb scaled by the feature-map ratio for retargeted triplanes (reference :351-360)."""
        if self.featmap_size is None or tuple(featmap_size) == tuple(self.featmap_size):
            return self.aabb
        scale = torch.tensor([featmap_size[i] / self.featmap_size[i] for i in range(3)], device=self.aabb.device)
        new = self.aabb.clone()
        new[:3] = self.aabb[:3] * scale
        new[3:] = self.aabb[3:] * scale
        return new

    @torch.no_grad()
    def decode_batch(self, triplane_feat, points, batch_size=2 ** 14, aabb=None):
        """Reference :319-333.  Any chunking gives identical per-point results, so all points go in one launch."""
        self.net.eval()
        return self.net.decode(points, triplane_feat, aabb=self.aabb if aabb is None else aabb, clamp_color=True)

    @torch.no_grad()
    def decode_grid(self, triplane_feat, reso, batch_size=2 ** 14, aabb=None):
        """Reference :335-349 -> [Nx, Ny, Nz, 4]."""
        self.net.eval()
        return self.net.decode_grid(triplane_feat, reso, aabb=self.aabb if aabb is None else aabb)

    @torch.no_grad()
    def decode_voxel(self, save_dir, triplane_feat, reso):
        """Reference :475-488: occupancy = sdf < 0, saved as r{reso}_voxel.npz."""
        H, W = triplane_feat[0].shape[-2:]
        D = triplane_feat[1].shape[-1]
        grid = self.decode_grid(triplane_feat, reso, aabb=self._resize_aabb((H, W, D)))
        vox = (grid[..., 0] < 0).cpu().numpy()
        os.makedirs(save_dir, exist_ok=True)
        np.savez_compressed(os.path.join(save_dir, f"r{reso}_voxel.npz"), voxel=vox)
        return vox
