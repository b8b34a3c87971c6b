"""AutoEncoderGroupSkip with the reference's constructor, parameter names and `decode` signature
(src/encoding/networks.py:124-225) — decode runs in libsin3dm_hip.so.

`decode(x, feat_maps, aabb)` keeps the reference semantics (sdf, sigmoid(rgb)); the per-plane conv blocks are
evaluated once per distinct triplane and cached (the reference recomputes them on every call with identical
results).

Training tier: `encode(vol)`, `forward(vol, x)` and `loss_and_grads(...)` run on the s3d_ae_*
entry points.  The parameters are then views of one flat device vector laid out [geo group | tex group] (the reference's
two AdamW parameter groups, :146-150); the input volume is reduced once to three 2-D projections (`set_volume`), which
is all the Conv3d-then-mean encoder needs.
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
        self._ae = None              # s3d_ae handle (training tier)
        self._ae_flat = None
        self._ae_layout = None
        self._ae_synced = None
        self._ae_dirty = False
        self._ae_volume = None
        self.tex_group_begin = None

    def geo_parameters(self):
        """reference :146-147"""
        return [p for n, p in self.named_parameters() if n.startswith(("geo_encoder", "geo_convs", "geo_decoder"))]

    def tex_parameters(self):
        """reference :149-150"""
        return [p for n, p in self.named_parameters() if n.startswith(("tex_encoder", "tex_convs", "tex_decoder"))]

    def reset_aabb(self, aabb):
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        self.aabb = aabb.to(self.aabb.device)

    # ------------------------------------------------------------------ training tier (s3d_ae_*)
    def _ensure_ae(self):
        lib = _lib.load()
        params = dict(self.named_parameters())
        if self._ae is None:
            h = C.c_void_p()
            cfg = _lib.DecoderCfg(*self.cfg)
            _lib.check(lib.s3d_ae_create(C.byref(cfg), C.byref(h)))
            self._ae = h
        intact = self._ae_flat is not None and all(
            params[n].data_ptr() == self._ae_flat.data_ptr() + 4 * off and params[n].device == self._ae_flat.device
            for n, off, _, _ in self._ae_layout)
        if not intact:
            dev = next(iter(params.values())).device
            _lib.require_gpu(next(iter(params.values())))
            layout = []
            for i in range(lib.s3d_ae_num_params(self._ae)):
                name, shape, nd, off = C.c_char_p(), (C.c_int64 * 5)(), C.c_int(), C.c_int64()
                _lib.check(lib.s3d_ae_param_info(self._ae, i, C.byref(name), shape, C.byref(nd), C.byref(off)))
                shp = tuple(shape[k] for k in range(nd.value))
                numel = 1
                for d in shp:
                    numel *= d
                layout.append((name.value.decode(), off.value, numel, shp))
            tb = C.c_int64()
            total = lib.s3d_ae_param_numel(self._ae, C.byref(tb))
            flat = torch.empty(total, device=dev, dtype=torch.float32)
            with torch.no_grad():
                for name, off, numel, shp in layout:
                    p = params[name]
                    assert tuple(p.shape) == shp, (name, tuple(p.shape), shp)
                    view = flat[off:off + numel].view(shp)
                    view.copy_(p.detach().to(dev, torch.float32))
                    p.data = view
            with torch.cuda.device(dev):
                _lib.check(lib.s3d_ae_attach(self._ae, _lib.ptr(flat), total))
            self._ae_flat, self._ae_layout, self.tex_group_begin = flat, layout, int(tb.value)
            self._ae_synced = None
            self._ae_volume = None
            params = dict(self.named_parameters())
        stamp = tuple((p.data_ptr(), p._version) for p in params.values())
        if stamp != self._ae_synced or self._ae_dirty:
            with torch.cuda.device(self._ae_flat.device):
                _lib.check(lib.s3d_ae_repack(self._ae, _lib.stream_ptr()))
            self._ae_synced, self._ae_dirty = stamp, False
        return lib

    @property
    def flat_parameters(self):
        self._ensure_ae()
        return self._ae_flat

    def mark_parameters_changed(self):
        """After writing to flat_parameters directly (fused optimizer): repack before the next use."""
        self._ae_dirty = True
        self._synced = None            # the decode handle mirrors the parameters from the host side
        self._prepared_for = None

    def split_flat(self, flat):
        self._ensure_ae()
        return {name: flat[off:off + numel].view(shp) for name, off, numel, shp in self._ae_layout}

    def set_volume(self, vol):
        """vol [1, 1+tex_channels, 2H, 2W, 2D]: reduce it to the encoder's three projections (done once per volume)."""
        lib = self._ensure_ae()
        _lib.require_gpu(vol)
        v = vol.contiguous().float()
        assert v.dim() == 5 and v.shape[0] == 1, tuple(v.shape)
        key = (v.data_ptr(), v._version, tuple(v.shape))
        if key != self._ae_volume:
            with torch.cuda.device(v.device):
                _lib.check(lib.s3d_ae_set_volume(self._ae, _lib.ptr(v), v.shape[1], v.shape[2], v.shape[3], v.shape[4],
                                                 _lib.stream_ptr()))
                torch.cuda.current_stream().synchronize()      # the library keeps projections, not `v`
            self._ae_volume = key
            self._feat_size = (v.shape[2] // 2, v.shape[3] // 2, v.shape[4] // 2)
        return lib

    def encode(self, vol):
        """reference :164-180 -> [xy [1,C,H,W], xz [1,C,H,D], yz [1,C,W,D]]"""
        lib = self.set_volume(vol)
        H, W, D = self._feat_size
        Cc = self.geo_feat_dim + self.tex_feat_dim
        dev = vol.device
        out = [torch.empty((1, Cc, a, b), device=dev, dtype=torch.float32) for a, b in ((H, W), (H, D), (W, D))]
        with torch.cuda.device(dev):
            _lib.check(lib.s3d_ae_encode(self._ae, _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.stream_ptr()))
        return out

    def loss_and_grads(self, vol, pts, sdf, tex, loss_cfg, aabb=None, grad_out=None, want_pred=False):
        """One ShapeAutoEncoder iteration without the optimizer (model.py:186-237 + loss.backward()): returns
        (losses [2] = (sdf_loss, tex_loss) on the device, pred or None, flat gradient of sdf_loss + tex_loss)."""
        lib = self.set_volume(vol)
        pts, sdf, tex = pts.contiguous().float(), sdf.contiguous().float(), tex.contiguous().float()
        N = pts.shape[0]
        dev = pts.device
        losses = torch.empty(2, device=dev, dtype=torch.float32)
        pred = torch.empty((N, 1 + self.cfg[5]), device=dev, dtype=torch.float32) if want_pred else None
        g = grad_out if grad_out is not None else torch.empty_like(self._ae_flat)
        with torch.cuda.device(dev):
            _lib.check(lib.s3d_ae_loss_grads(self._ae, _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(tex), N, self._aabb6(aabb),
                                             C.byref(loss_cfg), _lib.ptr(losses), _lib.ptr(pred), _lib.ptr(g), _lib.stream_ptr()))
        return losses, pred, g

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
        a = self.__dict__.pop("_ae", None)
        try:
            if h is not None:
                _lib.load().s3d_decoder_destroy(h)
            if a is not None:
                _lib.load().s3d_ae_destroy(a)
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
        """net(vol, x) (reference :222-224) through the training-tier forward (no gradient)."""
        lib = self.set_volume(vol)
        pts = x.contiguous().float()
        pred = torch.empty((pts.shape[0], 1 + self.cfg[5]), device=pts.device, dtype=torch.float32)
        with torch.cuda.device(pts.device):
            _lib.check(lib.s3d_ae_forward(self._ae, _lib.ptr(pts), pts.shape[0], self._aabb6(aabb), _lib.ptr(pred), _lib.stream_ptr()))
        return pred
