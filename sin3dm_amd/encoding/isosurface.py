"""Marching cubes on the MI355X (s3d_mc_*): the replacement for `mcubes.marching_cubes` in sdfgrid_to_mesh
(reference: src/encoding/utils3d.py:196-213).  The connected-component filter, decimation, UV atlas and texture baking
that follow it there (point_cloud_utils, open3d, xatlas, nvdiffrast) stay out of scope; vertex colours interpolated
from the decoded grid are offered instead."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from .. import _lib


_handles = {}       # device index -> s3d_mc handle (keeps its scan buffers between calls: 2 GB at 512 x 512 x 256)


def _handle(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _handles:
        h = C.c_void_p()
        _lib.check(_lib.load().s3d_mc_create(C.byref(h)))
        _handles[key] = h
    return _handles[key]


def release_workspace():
    """Free the cached scan buffers."""
    for h in _handles.values():
        _lib.load().s3d_mc_destroy(h)
    _handles.clear()


def marching_cubes(grid, iso=0.0, pad_value=1.0, n_attr=0):
    """grid: device tensor [X,Y,Z] or [X,Y,Z,1+A] (value in channel 0).  Returns (verts [nv,3] float32 in index
    coordinates, tris [nt,3] int32, attrs [nv,n_attr] or None).  pad_value=None disables the constant border that the
    reference adds so that surfaces reaching the boundary are closed."""
    _lib.require_gpu(grid)
    g = grid.contiguous().float()
    assert g.dim() in (3, 4), tuple(g.shape)
    X, Y, Z = g.shape[:3]
    stride = g.shape[3] if g.dim() == 4 else 1
    lib = _lib.load()
    h = _handle(g.device)
    nv, nt = C.c_int64(), C.c_int64()
    with torch.cuda.device(g.device):
        _lib.check(lib.s3d_mc_count(h, _lib.ptr(g), X, Y, Z, stride, float(iso), int(pad_value is not None),
                                    float(pad_value if pad_value is not None else 0.0), C.byref(nv), C.byref(nt), _lib.stream_ptr()))
        verts = torch.empty((nv.value, 3), device=g.device, dtype=torch.float32)
        tris = torch.empty((nt.value, 3), device=g.device, dtype=torch.int32)
        attrs = torch.empty((nv.value, n_attr), device=g.device, dtype=torch.float32) if n_attr else None
        _lib.check(lib.s3d_mc_extract(h, _lib.ptr(verts), _lib.ptr(attrs), int(n_attr), _lib.ptr(tris), _lib.stream_ptr()))
        torch.cuda.current_stream().synchronize()          # `g` may be a temporary: the kernels must be done with it
    return verts, tris, attrs


def mesh_components(tris, n_verts):
    """labels [n_verts] int32: the smallest vertex index of each vertex's connected component (device)."""
    _lib.require_gpu(tris)
    t = tris.contiguous().to(torch.int32)
    labels = torch.empty(n_verts, device=t.device, dtype=torch.int32)
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().s3d_mesh_components(_lib.ptr(t), t.shape[0], n_verts, _lib.ptr(labels), _lib.stream_ptr()))
    return labels


def largest_component(verts, tris, attrs=None):
    """Keep the connected component with the most faces and drop unreferenced vertices (sdfgrid_to_mesh with
    only_largest_cc=True, utils3d.py:204-208: pcu.connected_components + remove_unreferenced_mesh_vertices).
    The components come from the device kernel; the compaction below is index bookkeeping."""
    if tris.shape[0] == 0:
        return verts, tris, attrs
    labels = mesh_components(tris, verts.shape[0])
    face_label = labels[tris[:, 0].long()]
    roots, counts = torch.unique(face_label, return_counts=True)          # sorted by root id: ties -> smallest root
    best = roots[torch.argmax(counts)]
    keep_f = face_label == best
    keep_v = labels == best
    remap = torch.cumsum(keep_v.to(torch.int32), 0, dtype=torch.int32) - 1
    new_tris = remap[tris[keep_f].long()]
    return verts[keep_v], new_tris.contiguous(), (attrs[keep_v] if attrs is not None else None)


def export_obj(path, verts, tris, colors=None):
    """Wavefront OBJ with optional per-vertex colours (`v x y z r g b`)."""
    v = verts.detach().cpu().numpy()
    f = tris.detach().cpu().numpy().astype(np.int64) + 1
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as fh:
        if colors is not None:
            c = np.clip(colors.detach().cpu().numpy(), 0, 1)
            np.savetxt(fh, np.concatenate([v, c], 1), fmt="v %.6f %.6f %.6f %.4f %.4f %.4f")
        else:
            np.savetxt(fh, v, fmt="v %.6f %.6f %.6f")
        np.savetxt(fh, f, fmt="f %d %d %d")
