"""AutoEncoderGroupSkip with the reference's constructor, parameter names and `decode` signature
(src/encoding/networks.py:124-225) — decode runs in libsin3dm_hip.so.

`decode(x, feat_maps, aabb)` keeps the reference semantics (sdf, sigmoid(rgb)); the per-plane conv blocks are
evaluated once per distinct triplane and cached (the reference recomputes them on every call with identical
results).  `encode` (Conv3d encoder) is the next tier and raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from ..diffusion.unet_triplane import _register
from ..testing import ae_param_shapes


def get_networks(cfg):
    """Reference: src/encoding/networks.py:7-18 (only the `skip` variant is on the scored path)."""
    if cfg.enc_net_type != "skip":
        raise NotImplementedError("only --enc_net_type skip is built (scripts/run_single.sh:31 uses it)")
    return AutoEncoderGroupSkip(cfg.fdim_geo, cfg.fdim_tex, cfg.fdim_up, cfg.hidden_dim, cfg.n_hidden_layers,
                                use_tex=cfg.data_type != "sdf")


class AutoEncoderGroupSkip(nn.Module):
    def __init__(self, geo_feat_channels, tex_feat_channels, feat_channel_up, mlp_hidden_channels, mlp_hidden_layers,
                 use_tex=True, tex_channels=3, posenc=0):
        super().__init__()
        if not use_tex or posenc:
            raise NotImplementedError("use_tex=False / posenc>0 are not on the run_single.sh path")
        self.use_tex = use_tex
        self.geo_feat_dim = geo_feat_channels
        self.tex_feat_dim = tex_feat_channels
        self.cfg = (geo_feat_channels, tex_feat_channels, feat_channel_up, mlp_hidden_channels, mlp_hidden_layers,
                    tex_channels)
        shapes = dict(ae_param_shapes(*self.cfg))
        # encoder parameters exist in the reference's checkpoints (ckpt_final.pth['net']); kept so they load
        enc = {"geo_encoder.weight": (geo_feat_channels, 1, 4, 4, 4), "geo_encoder.bias": (geo_feat_channels,),
               "tex_encoder.weight": (tex_feat_channels, tex_channels + 1, 4, 4, 4),
               "tex_encoder.bias": (tex_feat_channels,)}
        self._decode_names = list(shapes)
        gen = torch.Generator().manual_seed(0)
        for name, shape in {**enc, **shapes}.items():
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            if ".norm_" in name:
                t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            elif ".out_layers.1." in name:
                t = torch.zeros(shape)                      # zero_module (src/encoding/blocks.py:226-230)
            else:
                t = (torch.rand(shape, generator=gen) * 2 - 1) * (1.0 / max(fan_in, 1)) ** 0.5
            _register(self, name, t)
        self.register_buffer("aabb", torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32))
        self._handle = None
        self._synced = None
        self._prepared_for = None

    def reset_aabb(self, aabb):
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        self.aabb = aabb.to(self.aabb.device)

    def encode(self, vol):
        raise NotImplementedError("AE encode is the next tier (SURVEY.md §8f rank 3)")

    # ------------------------------------------------------------------ HIP handle
    def _ensure_handle(self):
        lib = _lib.load()
        if self._handle is None:
            h = C.c_void_p()
            cfg = _lib.DecoderCfg(*self.cfg)
            _lib.check(lib.s3d_decoder_create(C.byref(cfg), C.byref(h)))
            self._handle = h
        params = dict(self.named_parameters())
        stamp = tuple((params[n].data_ptr(), params[n]._version) for n in self._decode_names)
        if stamp != self._synced:
            for name in self._decode_names:
                host = params[name].detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * host.dim())(*host.shape)
                _lib.check(lib.s3d_decoder_set_param(self._handle, name.encode(), C.c_void_p(host.data_ptr()), shape,
                                                     host.dim()))
            self._synced = stamp
            self._prepared_for = None
        return lib

    def __del__(self):
        h = self.__dict__.pop("_handle", None)      # not via nn.Module.__setattr__: it may run at interpreter exit
        if h is not None:
            try:
                _lib.load().s3d_decoder_destroy(h)
            except Exception:
                pass

    def prepare(self, feat_maps):
        """Run geo_convs / tex_convs for this triplane (cached by tensor identity and version)."""
        lib = self._ensure_handle()
        fm = [f.contiguous().float() for f in feat_maps]
        for f in fm:
            _lib.require_gpu(f)
            assert f.dim() == 4 and f.shape[0] == 1 and f.shape[1] == self.geo_feat_dim + self.tex_feat_dim
        key = tuple((f.data_ptr(), f._version, tuple(f.shape)) for f in feat_maps)
        if key == self._prepared_for:
            return lib
        H, W = fm[0].shape[-2:]
        D = fm[1].shape[-1]
        assert fm[1].shape[-2] == H and tuple(fm[2].shape[-2:]) == (W, D)
        with torch.cuda.device(fm[0].device):
            _lib.check(lib.s3d_decoder_prepare_triplane(self._handle, _lib.ptr(fm[0]), _lib.ptr(fm[1]), _lib.ptr(fm[2]),
                                                        H, W, D, _lib.stream_ptr()))
        self._prepared_for = key
        self._keepalive = fm
        return lib

    def _aabb6(self, aabb):
        a = self.aabb if aabb is None else aabb
        a = a.detach().to("cpu", torch.float32).reshape(6)
        return (C.c_float * 6)(*[float(v) for v in a])

    def decode(self, x, feat_maps, aabb=None, clamp_color=False):
        """x [N,3] -> [N, 1+3] = (sdf, sigmoid(rgb))  (reference :192-220)."""
        lib = self.prepare(feat_maps)
        _lib.require_gpu(x)
        pts = x.contiguous().float()
        out = torch.empty((pts.shape[0], 1 + self.cfg[5]), device=pts.device, dtype=torch.float32)
        with torch.cuda.device(pts.device):
            _lib.check(lib.s3d_decoder_decode_points(self._handle, _lib.ptr(pts), pts.shape[0], self._aabb6(aabb),
                                                     int(bool(clamp_color)), _lib.ptr(out), _lib.stream_ptr()))
        return out

    def decode_grid(self, feat_maps, reso, aabb=None):
        """Cell-centred grid over aabb, colours clamped to [0,1] (decode_grid + decode_batch, model.py:319-349)."""
        lib = self.prepare(feat_maps)
        a6 = self._aabb6(aabb)
        dims = (C.c_int * 3)()
        _lib.check(lib.s3d_decoder_grid_dims(a6, int(reso), dims))
        dev = feat_maps[0].device
        out = torch.empty((dims[0], dims[1], dims[2], 1 + self.cfg[5]), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.s3d_decoder_decode_grid(self._handle, int(reso), a6, _lib.ptr(out), _lib.stream_ptr()))
        return out

    def forward(self, vol, x, aabb=None):
        return self.decode(x, self.encode(vol), aabb=aabb)
