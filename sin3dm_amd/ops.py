"""Leaf operators of the triplane UNet as single calls into libsin3dm_hip.so (NCHW in, NCHW out).

These exist so each HIP kernel can be pinned against the reference op it replaces; the model itself
(diffusion/unet_triplane.py) runs the fused NHWC pipeline and never goes through here.
"""
from __future__ import annotations

import ctypes as C

import torch as th

from . import _lib


def _prep(fm):
    fm = [f.contiguous().float() for f in fm]
    for f in fm:
        _lib.require_gpu(f)
    B, Cc, H, W = fm[0].shape
    D = fm[1].shape[-1]
    assert fm[1].shape == (B, Cc, H, D) and fm[2].shape == (B, Cc, W, D)
    return fm, B, Cc, H, W, D


def _host3(ts):
    ts = [t.detach().to("cpu", th.float32).contiguous() for t in ts]
    return ts, (C.c_void_p * 3)(*[C.c_void_p(t.data_ptr()) for t in ts])


def triplane_conv(fm, weights, biases, is_rollout):
    """TriplaneConv.forward (src/diffusion/unet_triplane.py:31-60); weights OIHW per plane."""
    fm, B, Cc, H, W, D = _prep(fm)
    cout, _, k, _ = weights[0].shape
    outs = [th.empty((B, cout) + tuple(f.shape[-2:]), device=f.device) for f in fm]
    wk, wp = _host3(weights)
    bk, bp = _host3(biases)
    _lib.check(_lib.load().s3d_op_triplane_conv(_lib.ptr3(fm), _lib.ptr3(outs), B, Cc, H, W, D, cout, k,
                                                int(bool(is_rollout)), wp, bp, _lib.stream_ptr()))
    return outs


def triplane_norm_silu(fm, gammas, betas):
    """TriplaneNorm + TriplaneSiLU (src/diffusion/unet_triplane.py:63-95)."""
    fm, B, Cc, H, W, D = _prep(fm)
    outs = [th.empty_like(f) for f in fm]
    gk, gp = _host3(gammas)
    bk, bp = _host3(betas)
    _lib.check(_lib.load().s3d_op_triplane_norm_silu(_lib.ptr3(fm), _lib.ptr3(outs), B, Cc, H, W, D, gp, bp,
                                                     _lib.stream_ptr()))
    return outs


def _resample(fm, sizes, mode):
    fm = [f.contiguous().float() for f in fm]
    B, Cc = fm[0].shape[:2]
    hi = (C.c_int * 3)(*[f.shape[-2] for f in fm])
    wi = (C.c_int * 3)(*[f.shape[-1] for f in fm])
    ho = (C.c_int * 3)(*[s[0] for s in sizes])
    wo = (C.c_int * 3)(*[s[1] for s in sizes])
    outs = [th.empty((B, Cc) + tuple(s), device=f.device) for f, s in zip(fm, sizes)]
    _lib.check(_lib.load().s3d_op_triplane_resample(_lib.ptr3(fm), _lib.ptr3(outs), B, Cc, hi, wi, ho, wo, mode,
                                                    _lib.stream_ptr()))
    return outs


def triplane_downsample2x(fm):
    """TriplaneDownsample2x (src/diffusion/unet_triplane.py:127-145)."""
    return _resample(fm, [(f.shape[-2] // 2, f.shape[-1] // 2) for f in fm], 0)


def triplane_resize(fm, sizes):
    """Bilinear align_corners=False to explicit sizes (TriplaneUpsample2x :106-124, skip resize :494-499)."""
    return _resample(fm, sizes, 1)


def timestep_embedding(timesteps, dim):
    """timestep_embedding (src/diffusion/nn.py:103-121): [B] -> [B, dim] = cos | sin."""
    _lib.require_gpu(timesteps)
    t = timesteps.to(th.float32).contiguous()
    out = th.empty((t.shape[0], dim), device=t.device)
    _lib.check(_lib.load().s3d_op_timestep_embed(_lib.ptr(t), t.shape[0], dim, _lib.ptr(out), _lib.stream_ptr()))
    return out


def triplane_resblock(fm, emb, params, cout, use_scale_shift_norm=True, is_rollout=True):
    """TriplaneResBlock._forward (src/diffusion/unet_triplane.py:269-311); `params`: {state-dict name relative to the
    block: tensor}; emb: [B, emb_dim] time embedding."""
    fm, B, Cc, H, W, D = _prep(fm)
    emb = emb.contiguous().float()
    host = {k: v.detach().to("cpu", th.float32).contiguous() for k, v in params.items()}
    names = (C.c_char_p * len(host))(*[k.encode() for k in host])
    ptrs = (C.c_void_p * len(host))(*[C.c_void_p(v.data_ptr()) for v in host.values()])
    outs = [th.empty((B, cout) + tuple(f.shape[-2:]), device=f.device) for f in fm]
    _lib.check(_lib.load().s3d_op_triplane_resblock(_lib.ptr3(fm), _lib.ptr3(outs), _lib.ptr(emb), B, Cc, cout, H, W, D,
                                                    emb.shape[1], int(bool(use_scale_shift_norm)), int(bool(is_rollout)),
                                                    names, ptrs, len(host), _lib.stream_ptr()))
    return outs
