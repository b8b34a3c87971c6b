"""ctypes front-end of the C oracle (oracle/sin3dm_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under
sin3dm_amd/ imports it.  Arrays are numpy, fp32, in the reference's NCHW layout.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsin3dm_oracle.so")
_lib = None

c_fp = C.POINTER(C.c_float)
c_dp = C.POINTER(C.c_double)


class _Params(C.Structure):
    _fields_ = [("n", C.c_int), ("names", C.POINTER(C.c_char_p)), ("ptrs", C.POINTER(c_fp))]


class _UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("model_channels", C.c_int), ("out_channels", C.c_int),
                ("n_levels", C.c_int), ("channel_mult", C.c_int * 8),
                ("use_scale_shift_norm", C.c_int), ("rollout", C.c_int)]


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "sin3dm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libsin3dm_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_fp)


class Params:
    """name -> fp32 array table handed to C by pointer (kept alive by this object)."""

    def __init__(self, sd):
        self.arrs = {}
        for k, v in sd.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            self.arrs[k] = np.ascontiguousarray(v, dtype=np.float32)
        n = len(self.arrs)
        self._names = (C.c_char_p * n)(*[k.encode() for k in self.arrs])
        self._ptrs = (c_fp * n)(*[a.ctypes.data_as(c_fp) for a in self.arrs.values()])
        self.c = _Params(n, self._names, self._ptrs)

    def sub(self, prefix):
        return Params({k[len(prefix):]: v for k, v in self.arrs.items() if k.startswith(prefix)})


def _as_params(p):
    return p if isinstance(p, Params) else Params(p)


def unet_forward(sd, x, t, H, W, D, model_channels, channel_mult=(1, 2), use_scale_shift_norm=True,
                 rollout=True, in_channels=12, out_channels=12):
    pr = _as_params(sd)
    x, xp = _f(x)
    B = x.shape[0]
    t, tp = _f(np.asarray(t, dtype=np.float32))
    out = np.empty((B, out_channels, H + D, W + D), dtype=np.float32)
    cfg = _UNetCfg(in_channels, model_channels, out_channels, len(channel_mult),
                   (C.c_int * 8)(*list(channel_mult) + [0] * (8 - len(channel_mult))),
                   int(use_scale_shift_norm), int(rollout))
    lib().orc_unet_forward(C.byref(cfg), C.byref(pr.c), xp, tp, B, H, W, D, out.ctypes.data_as(c_fp))
    return out


def _tri_call(fn, pr, prefix, fm, B, C_, H, W, D, cout, *extra):
    (xy, pxy), (xz, pxz), (yz, pyz) = _f(fm[0]), _f(fm[1]), _f(fm[2])
    outs = [np.empty((B, cout, H, W), np.float32), np.empty((B, cout, H, D), np.float32),
            np.empty((B, cout, W, D), np.float32)]
    return (pxy, pxz, pyz), outs, [o.ctypes.data_as(c_fp) for o in outs]


def triplane_conv(sd, prefix, fm, cout, k, rollout):
    pr = _as_params(sd)
    B, C_, H, W = fm[0].shape
    D = fm[1].shape[-1]
    ins, outs, op = _tri_call(None, pr, prefix, fm, B, C_, H, W, D, cout)
    lib().orc_triplane_conv(C.byref(pr.c), prefix.encode(), *ins, B, C_, H, W, D, cout, k, int(rollout), *op)
    return outs


def triplane_norm_silu(sd, prefix, fm):
    pr = _as_params(sd)
    B, C_, H, W = fm[0].shape
    D = fm[1].shape[-1]
    ins, outs, op = _tri_call(None, pr, prefix, fm, B, C_, H, W, D, C_)
    lib().orc_triplane_norm_silu(C.byref(pr.c), prefix.encode(), *ins, B, C_, H, W, D, *op)
    return outs


def triplane_resblock(sd, prefix, fm, emb, cout, ssn=True, rollout=True):
    pr = _as_params(sd)
    B, C_, H, W = fm[0].shape
    D = fm[1].shape[-1]
    emb, ep = _f(emb)
    ins, outs, op = _tri_call(None, pr, prefix, fm, B, C_, H, W, D, cout)
    lib().orc_triplane_resblock(C.byref(pr.c), prefix.encode(), *ins, ep, B, C_, H, W, D, emb.shape[1], cout,
                                int(ssn), int(rollout), *op)
    return outs


def avgpool2(x):
    x, xp = _f(x)
    B, C_, H, W = x.shape
    y = np.empty((B, C_, H // 2, W // 2), np.float32)
    lib().orc_avgpool2(xp, y.ctypes.data_as(c_fp), B * C_, H, W)
    return y


def bilinear(x, ho, wo):
    x, xp = _f(x)
    B, C_, H, W = x.shape
    y = np.empty((B, C_, ho, wo), np.float32)
    lib().orc_bilinear(xp, y.ctypes.data_as(c_fp), B * C_, H, W, ho, wo)
    return y


def timestep_embedding(t, dim):
    t, tp = _f(np.asarray(t, dtype=np.float32))
    e = np.empty((t.shape[0], dim), np.float32)
    lib().orc_timestep_embedding(tp, e.ctypes.data_as(c_fp), t.shape[0], dim)
    return e


def compose(xy, xz, yz):
    (xy, a), (xz, b), (yz, c) = _f(xy), _f(xz), _f(yz)
    lead = xy.shape[:-2]
    H, W = xy.shape[-2:]
    D = xz.shape[-1]
    out = np.empty(lead + (H + D, W + D), np.float32)
    lib().orc_compose(a, b, c, out.ctypes.data_as(c_fp), int(np.prod(lead)), H, W, D)
    return out


def decompose(comp, H, W, D):
    comp, cp = _f(comp)
    lead = comp.shape[:-2]
    xy, xz, yz = (np.empty(lead + s, np.float32) for s in ((H, W), (H, D), (W, D)))
    lib().orc_decompose(cp, xy.ctypes.data_as(c_fp), xz.ctypes.data_as(c_fp), yz.ctypes.data_as(c_fp),
                        int(np.prod(lead)), H, W, D)
    return xy, xz, yz


TABLE_ROWS = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_mean_coef1",
              "posterior_mean_coef2")


def schedule_tables(keep=None, steps=1000):
    """float64 tables [8,T'] (+ timestep_map) of the linear schedule, optionally respaced to `keep`."""
    base = np.empty(steps, np.float64)
    lib().orc_linear_betas(base.ctypes.data_as(c_dp), steps)
    if keep is None:
        betas, tmap = base, np.arange(steps, dtype=np.int64)
    else:
        mask = np.zeros(steps, np.uint8)
        mask[list(keep)] = 1
        betas = np.empty(steps, np.float64)
        tmap = np.empty(steps, np.int64)
        n = lib().orc_respace_betas(base.ctypes.data_as(c_dp), steps, mask.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    betas.ctypes.data_as(c_dp), tmap.ctypes.data_as(C.POINTER(C.c_int64)))
        betas, tmap = betas[:n].copy(), tmap[:n].copy()
    T = betas.shape[0]
    tab = np.empty((8, T), np.float64)
    lib().orc_tables_from_betas(betas.ctypes.data_as(c_dp), T, tab.ctypes.data_as(c_dp))
    return tab, tmap


def schedule_tables_named(steps=1000):
    """The forward-process tables q_sample needs (gaussian_diffusion.py:141-146), float64, from the same betas."""
    tab, _ = schedule_tables(None, steps)
    return {"betas": tab[0], "alphas_cumprod": tab[1], "sqrt_alphas_cumprod": np.sqrt(tab[1]),
            "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - tab[1])}


def p_sample_update(model_out, x, eps, tab, t, clip=True, mean_eps=False, var_small=False, want_mean=False):
    """(sample, pred_xstart[, mean]) of p_sample; mean_eps: the model predicts the noise (ModelMeanType.EPSILON);
    var_small: ModelVarType.FIXED_SMALL."""
    (mo, a), (x, b), (eps, c) = _f(model_out), _f(x), _f(eps)
    s, p = np.empty_like(mo), np.empty_like(mo)
    m = np.empty_like(mo) if want_mean else None
    tab = np.ascontiguousarray(tab, np.float64)
    lib().orc_p_sample_update_ex(a, b, c, s.ctypes.data_as(c_fp), p.ctypes.data_as(c_fp),
                                 m.ctypes.data_as(c_fp) if want_mean else None, C.c_size_t(mo.size),
                                 tab.ctypes.data_as(c_dp), tab.shape[1], int(t), int(clip), int(mean_eps), int(var_small))
    return (s, p, m) if want_mean else (s, p)


def ddim_update(model_out, x, noise, tab, t, clip=True, eta=0.0, mean_eps=False):
    (mo, a), (x, b), (noise, c) = _f(model_out), _f(x), _f(noise)
    s, p = np.empty_like(mo), np.empty_like(mo)
    tab = np.ascontiguousarray(tab, np.float64)
    lib().orc_ddim_update_ex(a, b, c, s.ctypes.data_as(c_fp), p.ctypes.data_as(c_fp), C.c_size_t(mo.size),
                             tab.ctypes.data_as(c_dp), tab.shape[1], int(t), int(clip), C.c_float(eta), int(mean_eps))
    return s, p


def ae_decode(sd, pts, xy, xz, yz, aabb, geo_dim=4, tex_dim=8, up=64, hid=256, nh=4):
    pr = _as_params(sd)
    (pts, pp), (xy, a), (xz, b), (yz, c), (aabb, ab) = _f(pts), _f(xy), _f(xz), _f(yz), _f(aabb)
    H, W = xy.shape[-2:]
    D = xz.shape[-1]
    out = np.empty((pts.shape[0], 4), np.float32)
    lib().orc_ae_decode(C.byref(pr.c), pp, pts.shape[0], a, b, c, H, W, D, ab, geo_dim, tex_dim, up, hid, nh,
                        out.ctypes.data_as(c_fp))
    return out


def ae_plane_block(sd, prefix, planes, up):
    pr = _as_params(sd)
    arrs = [_f(p) for p in planes]
    cin = arrs[0][0].shape[-3]
    ph = (C.c_int * 3)(*[a.shape[-2] for a, _ in arrs])
    pw = (C.c_int * 3)(*[a.shape[-1] for a, _ in arrs])
    outs = [np.empty((1, up, a.shape[-2], a.shape[-1]), np.float32) for a, _ in arrs]
    ip = (c_fp * 3)(*[p for _, p in arrs])
    op = (c_fp * 3)(*[o.ctypes.data_as(c_fp) for o in outs])
    lib().orc_ae_plane_block(C.byref(pr.c), prefix.encode(), ip, ph, pw, cin, up, op)
    return outs


def marching_cubes(grid, iso=0.0, pad_value=1.0):
    """Checker of the device marching cubes (PyMCubes parity unpinned, see sin3dm_oracle.c).  grid: [X,Y,Z] or
    [X,Y,Z,S] (value in channel 0).  -> (verts [nv,3] float32 index coordinates, tris [nt,3] int32)."""
    g = np.ascontiguousarray(grid, np.float32)
    X, Y, Z = g.shape[:3]
    stride = g.shape[3] if g.ndim == 4 else 1
    nv, nt = C.c_int64(), C.c_int64()
    pad, pv = int(pad_value is not None), float(pad_value if pad_value is not None else 0.0)
    f = lib().orc_marching_cubes
    f.restype = C.c_int
    args = (g.ctypes.data_as(c_fp), X, Y, Z, stride, C.c_float(iso), pad, C.c_float(pv))
    f(*args, None, None, C.byref(nv), C.byref(nt))
    verts = np.empty((nv.value, 3), np.float32)
    tris = np.empty((nt.value, 3), np.int32)
    f(*args, verts.ctypes.data_as(c_fp), tris.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nv), C.byref(nt))
    return verts, tris


def mesh_components(tris, n_verts):
    """labels[v] = smallest vertex index of v's connected component (checker of s3d_mesh_components)."""
    t = np.ascontiguousarray(tris, np.int32)
    labels = np.empty(n_verts, np.int32)
    f = lib().orc_mesh_components
    f.restype = None
    f(t.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(len(t)), C.c_int64(n_verts), labels.ctypes.data_as(C.POINTER(C.c_int32)))
    return labels
