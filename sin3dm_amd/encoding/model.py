"""ShapeAutoEncoder (reference: src/encoding/model.py): checkpoints, decode, and the auto-encoder training loop
(:51-139, 178-258, 309-317) on the MI355X.  Mesh/texture export (PyMCubes, xatlas, nvdiffrast) and the tensorboard
figures are out of scope."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib
from .networks import get_networks

_SDF_LOSS = {"l1": 0, "weightedl1": 1}
_TEX_LOSS = {"l1": 0, "l2": 1, "huber": 2}


class FlatGroupAdamW:
    """torch.optim.AdamW over the auto-encoder's two parameter groups (geo: lr*split, tex: lr; betas 0.9/0.999, eps 1e-8,
    weight_decay 0.01 = torch's default, which the reference inherits at model.py:131-137) followed by ExponentialLR,
    on the flat parameter vector: two launches of the fused Adam kernel per step."""

    def __init__(self, net, lr, lr_split=-1.0, lr_decay=1.0, weight_decay=0.01):
        self.net = net
        flat = net.flat_parameters
        tb = net.tex_group_begin
        self.ranges = ((0, tb), (tb, flat.numel()))
        self.lrs = [lr * lr_split if lr_split > 0 else lr, lr]
        self.lr_decay, self.weight_decay = float(lr_decay), float(weight_decay)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
        self.steps = 0

    def step(self, flat_grad):
        flat = self.net.flat_parameters
        self.steps += 1
        lib = _lib.load()
        none = (C.c_void_p * 1)(None)
        rates = (C.c_float * 1)(0.0)
        with torch.cuda.device(flat.device):
            for (b, e), lr in zip(self.ranges, self.lrs):
                _lib.check(lib.s3d_train_adamw_ema(_lib.ptr(flat[b:e]), _lib.ptr(flat_grad[b:e]), _lib.ptr(self.exp_avg[b:e]),
                                                   _lib.ptr(self.exp_avg_sq[b:e]), none, rates, 0, e - b, lr, 0.9, 0.999, 1e-8,
                                                   self.weight_decay, self.steps, _lib.stream_ptr()))
        self.net.mark_parameters_changed()
        self.lrs = [lr * self.lr_decay for lr in self.lrs]            # ExponentialLR.step()

    def state_dict(self):
        return {"steps": self.steps, "lrs": list(self.lrs), "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu()}

    def load_state_dict(self, sd):
        self.steps, self.lrs = int(sd["steps"]), list(sd["lrs"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class ShapeAutoEncoder:
    """Reads and writes `<log_dir>/ckpt_{name}.pth` exactly where the reference does (src/encoding/model.py:143, 161), so an
    experiment directory trained by either side loads on the other; trains, encodes and decodes on the MI355X."""

    def __init__(self, log_dir, cfg, device=None):
        self.log_dir = log_dir
        self.device = device or torch.device(f"cuda:{getattr(cfg, 'gpu_id', 0)}")
        self.net = get_networks(cfg).to(self.device)
        self.aabb = self.net.aabb.clone()
        self.featmap_size = None
        # training hyper-parameters (reference :18-37)
        g = lambda k, d: getattr(cfg, k, d)
        self.batch_size, self.n_iters, self.vol_ratio = g("enc_batch_size", 65536), g("enc_n_iters", 25000), g("vol_ratio", 0.1)
        self.fm_reso, self.data_type = g("fm_reso", 128), g("data_type", "sdftex")
        self.sdf_loss_type, self.tex_loss_type = g("sdf_loss", "weightedl1"), g("tex_loss", "l1")
        self.tex_weight, self.tex_threshold_ratio = g("tex_weight", 1.0), g("tex_threshold_ratio", 0.999)
        self.sdf_renorm = g("sdf_renorm", 0)
        self.init_lr, self.lr_split, self.min_lr_ratio = g("enc_lr", 5e-3), g("enc_lr_split", -1), g("enc_lr_decay", 0.01)
        self.material = {"Ka": None, "Kd": None, "Ks": None, "Ns": None}
        self.sdf_threshold = None
        self.input_grid = None

    def load_ckpt(self, name=None):
        """Reference :158-176 — `<log_dir>/ckpt_{name}.pth`, dict {net, optimizer, scheduler, Ka, Kd, Ks, Ns, aabb,
        featmap_size}.  (Directories written by round 1 of this build kept it under `model/`: read as a fallback.)"""
        path = name if name and os.path.isabs(str(name)) else os.path.join(self.log_dir, f"ckpt_{name}.pth")
        if not os.path.exists(path) and os.path.exists(os.path.join(self.log_dir, "model", f"ckpt_{name}.pth")):
            path = os.path.join(self.log_dir, "model", f"ckpt_{name}.pth")
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

    @torch.no_grad()
    def decode_mesh(self, save_dir, triplane_feat, reso, name="object.obj", only_largest_cc=True, save_voxel=True):
        """The geometry half of decode_texmesh (reference :362-390): decode_grid -> voxel.npz -> iso-surface of the
        padded SDF grid at level 0 -> largest connected component -> `v / reso * box_size + box_min`, all on the device
        (sdfgrid_to_mesh, utils3d.py:196-208, used PyMCubes + point_cloud_utils on the CPU).  What follows there —
        quadric decimation, UV atlas, texture baking (open3d, xatlas, nvdiffrast) — is out of scope; the decoded colour
        is interpolated onto the vertices and written as a vertex-coloured OBJ instead."""
        from .isosurface import export_obj, largest_component, marching_cubes
        H, W = triplane_feat[0].shape[-2:]
        D = triplane_feat[1].shape[-1]
        aabb = self._resize_aabb((H, W, D))
        grid = self.decode_grid(triplane_feat, reso, aabb=aabb)                  # [Nx, Ny, Nz, 1+3], colours clamped
        os.makedirs(save_dir, exist_ok=True)
        if save_voxel:
            np.savez_compressed(os.path.join(save_dir, "voxel.npz"), vox_grid=(grid[..., 0] < 0).cpu().numpy())
        verts, tris, cols = marching_cubes(grid, 0.0, 1.0, n_attr=grid.shape[-1] - 1)
        if only_largest_cc:
            verts, tris, cols = largest_component(verts, tris, cols)
        box_min = aabb[:3]
        box_size = aabb[3:].max() - aabb[:3].min()                               # the reference's re-normalisation (:385-387)
        verts = verts / float(reso) * box_size + box_min
        export_obj(os.path.join(save_dir, name), verts, tris, cols)
        return verts, tris, cols

    # ------------------------------------------------------------------ training (reference :51-139, 178-258)
    def _load_data(self, path, sdf_renorm=False):
        """The preprocessed .npz of one shape: grid + near-surface samples (reference :51-112)."""
        if self.data_type != "sdftex":
            raise NotImplementedError("only data_type sdftex is built (scripts/run_single.sh)")
        data = np.load(path)
        dev = self.device
        self.aabb = torch.from_numpy(data["aabb"]).float().to(dev)
        self.sdf_threshold = float(data["threshold"])
        self.material = {"Ka": data["Ka"].tolist() if "Ka" in data else [0, 0, 0], "Kd": data["Kd"].tolist() if "Kd" in data else [1, 1, 1],
                         "Ks": data["Ks"].tolist() if "Ks" in data else [0.4, 0.4, 0.4], "Ns": data["Ns"].tolist() if "Ns" in data else 10}
        pts_grid, sdf_grid, tex_grid = data["pts_grid"], data["sdf_grid"], data["tex_grid"]
        fs = (torch.tensor(pts_grid.shape[:3]).float() * (self.fm_reso / max(pts_grid.shape[:3]))).long().tolist()
        self.featmap_size = [int(x // 2 * 2) for x in fs]
        grid = torch.from_numpy(np.concatenate([sdf_grid[np.newaxis], tex_grid.transpose(3, 0, 1, 2)], axis=0)).float().to(dev)
        want = [x * 2 for x in self.featmap_size]
        if list(grid.shape[1:]) != want:
            grid = F.interpolate(grid.unsqueeze(0), size=want, mode="trilinear", align_corners=False).squeeze(0)
        self.input_grid = grid.unsqueeze(0).contiguous()                                     # [1, C, 2H, 2W, 2D]
        thr = self.sdf_threshold
        self.pts_grid = torch.from_numpy(pts_grid).float().to(dev).view(-1, 3)
        self.sdf_grid = torch.from_numpy(sdf_grid).float().to(dev).view(-1, 1).clamp_(-thr, thr)
        self.pts_near_surf = torch.from_numpy(data["pts_near_surf"]).float().to(dev).view(-1, 3)
        self.sdf_near_surf = torch.from_numpy(data["sdf_near_surf"]).float().to(dev).view(-1, 1).clamp_(-thr, thr)
        tc = tex_grid.shape[-1]
        self.tex_grid = torch.from_numpy(tex_grid).float().to(dev).view(-1, tc)
        self.tex_near_surf = torch.from_numpy(data["tex_near_surf"]).float().to(dev).view(-1, tc)
        if "pts_on_surf" in data:
            self.pts_on_surf = torch.from_numpy(data["pts_on_surf"]).float().to(dev).view(-1, 3)
            self.tex_on_surf = torch.from_numpy(data["tex_on_surf"]).float().to(dev).view(-1, tc)
        if sdf_renorm:
            self.sdf_grid = self.sdf_grid / thr
            self.sdf_near_surf = self.sdf_near_surf / thr

    def _sample_batch(self, batch_size):
        """vol_ratio of the batch from the regular grid, the rest near the surface (reference :114-127)."""
        n_grid = int(batch_size * self.vol_ratio)
        gi = torch.randint(0, self.pts_grid.shape[0], (n_grid,), device=self.device)
        si = torch.randint(0, self.pts_near_surf.shape[0], (batch_size - n_grid,), device=self.device)
        return {"pts": torch.cat([self.pts_grid[gi], self.pts_near_surf[si]], dim=0),
                "sdf": torch.cat([self.sdf_grid[gi], self.sdf_near_surf[si]], dim=0),
                "tex": torch.cat([self.tex_grid[gi], self.tex_near_surf[si]], dim=0)}

    def _loss_cfg(self):
        band = 1.0 if self.sdf_renorm else self.sdf_threshold
        return _lib.AeLossCfg(_SDF_LOSS[self.sdf_loss_type], _TEX_LOSS[self.tex_loss_type], band, self.tex_threshold_ratio, self.tex_weight)

    def _set_optimizer(self, lr, min_lr_ratio=0.01):
        """AdamW with the geo group at lr*lr_split + ExponentialLR reaching min_lr_ratio after n_iters (reference :129-139)."""
        self.optimizer = FlatGroupAdamW(self.net, lr, self.lr_split, lr_decay=min_lr_ratio ** (1 / self.n_iters))

    def train_step(self, data):
        """_forward_batch + update_network (reference :178-237) -> {"sdf_loss", "tex_loss"} device scalars."""
        losses, _, grads = self.net.loss_and_grads(self.input_grid, data["pts"], data["sdf"], data["tex"], self._loss_cfg(),
                                                   aabb=self.aabb, grad_out=getattr(self, "_grad", None))
        self._grad = grads
        self.optimizer.step(grads)
        return {"sdf_loss": losses[0], "tex_loss": losses[1]}

    def train(self, data_path, log_every=100):
        self._load_data(data_path, sdf_renorm=bool(self.sdf_renorm))
        self.net.reset_aabb(self.aabb)
        self._set_optimizer(self.init_lr, self.min_lr_ratio)
        os.makedirs(self.log_dir, exist_ok=True)
        for i in range(self.n_iters):
            self.step = i
            losses = self.train_step(self._sample_batch(self.batch_size))
            if log_every and (i % log_every == 0 or i == self.n_iters - 1):
                vals = {k: float(v) for k, v in losses.items()}
                print(f"[ae {i}/{self.n_iters}] " + " ".join(f"{k} {v:.5f}" for k, v in vals.items()), flush=True)
                with open(os.path.join(self.log_dir, "progress.jsonl"), "a") as f:
                    f.write(json.dumps({"step": i, **vals}) + "\n")
        stat = self.evaluate()
        with open(os.path.join(self.log_dir, "eval_stat.json"), "w") as f:
            json.dump(stat, f, indent=2)
        self.save_ckpt("final")

    @torch.no_grad()
    def evaluate(self):
        """tsdf statistics over the regular grid (reference :273-296, 491-515)."""
        fm = self.encode()
        pred = self.decode_batch(fm, self.pts_grid)[..., :1]
        gt = self.sdf_grid
        if self.sdf_renorm:
            pred, gt = pred * self.sdf_threshold, gt * self.sdf_threshold
        l1 = (pred - gt).abs()
        stat = {"mean_tsdf_l1_error": l1.mean().item(), "mean_tsdf_rel_error": (l1 / gt.abs()).mean().item(),
                "mean_tsdf_acc": (pred * gt >= 0).float().mean().item()}
        if hasattr(self, "pts_on_surf"):
            tex = self.decode_batch(fm, self.pts_on_surf)[..., 1:]
            stat["surf_tex_l1_error"] = (tex - self.tex_on_surf).abs().mean().item()
        return stat

    @torch.no_grad()
    def encode(self, vol=None):
        """reference :309-317"""
        return self.net.encode(self.input_grid if vol is None else vol)

    def save_ckpt(self, name):
        """reference :141-156 (the optimizer entry holds this implementation's flat Adam state)."""
        os.makedirs(self.log_dir, exist_ok=True)
        sd = {k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()}
        torch.save({"net": sd, "optimizer": self.optimizer.state_dict() if hasattr(self, "optimizer") else None, "scheduler": None,
                    **self.material, "aabb": self.aabb.tolist(), "featmap_size": self.featmap_size},
                   os.path.join(self.log_dir, f"ckpt_{name}.pth"))
