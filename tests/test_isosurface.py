"""Iso-surface extraction.  PyMCubes is not importable here, so parity with it is UNPINNED;
what is checked: (CPU) the C restatement + generated case table produce closed, consistently oriented meshes of the
right topology and measure on analytic fields, including fields full of ambiguous faces; (GPU) the device kernels
reproduce the restatement exactly."""
from collections import Counter

import numpy as np
import pytest

from conftest import REPO  # noqa: F401


def _field(kind, n=28):
    ax = np.linspace(-1, 1, n)
    x, y, z = np.meshgrid(ax, ax * 0.9, ax * 1.1, indexing="ij")
    if kind == "sphere":
        return (np.sqrt(x * x + y * y + z * z) - 0.7).astype(np.float32), 2
    if kind == "torus":
        return (np.sqrt((np.sqrt(x * x + y * y) - 0.6) ** 2 + z * z) - 0.22).astype(np.float32), 0
    if kind == "two":
        a = np.sqrt((x - 0.45) ** 2 + y * y + z * z) - 0.3
        b = np.sqrt((x + 0.45) ** 2 + y * y + z * z) - 0.35
        return np.minimum(a, b).astype(np.float32), 4
    raise KeyError(kind)


def _invariants(v, t):
    """closed + consistently oriented: every directed edge occurs once and so does its reverse"""
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    cnt = Counter(map(tuple, e.tolist()))
    closed = all(c == 1 and cnt.get((b, a), 0) == 1 for (a, b), c in cnt.items())
    return closed, len(v) - len(cnt) // 2 + len(t)


@pytest.mark.parametrize("kind", ["sphere", "torus", "two"])
def test_oracle_mesh_invariants(oracle, kind):
    f, euler = _field(kind)
    v, t = oracle.marching_cubes(f, 0.0, 1.0)
    closed, chi = _invariants(v, t)
    assert closed and chi == euler, (closed, chi)
    assert t.min() == 0 and t.max() == len(v) - 1 and len(np.unique(t)) == len(v)      # every vertex used
    if kind == "sphere":
        n = f.shape[0]
        p = v * (2.0 / (n - 1)) * np.asarray([1, 0.9, 1.1]) - np.asarray([1, 0.9, 1.1])
        a, b, c = p[t[:, 0]], p[t[:, 1]], p[t[:, 2]]
        area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
        vol = np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6
        assert abs(area - 4 * np.pi * 0.49) < 0.02 * 4 * np.pi * 0.49
        assert abs(vol - 4 / 3 * np.pi * 0.343) < 0.02 * 4 / 3 * np.pi * 0.343           # positive: normals point outwards


def test_oracle_ambiguous_faces_stay_watertight(oracle):
    """white noise: almost every cell has ambiguous faces; the border padding closes the surface"""
    rng = np.random.Generator(np.random.PCG64(5))
    f = rng.standard_normal((14, 11, 9)).astype(np.float32)
    v, t = oracle.marching_cubes(f, 0.0, 1.0)
    closed, _ = _invariants(v, t)
    assert closed and len(t) > 1000
    v2, t2 = oracle.marching_cubes(f, 0.0, None)                  # no padding: open at the border, still manifold inside
    e = np.concatenate([t2[:, [0, 1]], t2[:, [1, 2]], t2[:, [2, 0]]])
    cnt = Counter(map(tuple, e.tolist()))
    assert all(c == 1 for c in cnt.values())                       # no directed edge twice
    assert all((v2 >= 0).all(axis=1)) and (v2 <= np.asarray(f.shape) - 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sphere", "torus", "two", "noise"])
def test_hip_matches_restatement(oracle, kind):
    import torch
    from sin3dm_amd.encoding.isosurface import marching_cubes
    if kind == "noise":
        f = np.random.Generator(np.random.PCG64(6)).standard_normal((33, 20, 27)).astype(np.float32)
    else:
        f, _ = _field(kind, 40)
    for pad in (1.0, None):
        v_ref, t_ref = oracle.marching_cubes(f, 0.0, pad)
        v, t, _ = marching_cubes(torch.from_numpy(f).cuda(), 0.0, pad)
        assert np.array_equal(t.cpu().numpy(), t_ref)
        assert np.array_equal(v.cpu().numpy(), v_ref)


@pytest.mark.gpu
def test_hip_strided_grid_with_vertex_colours():
    """reads the sdf channel of a decode_grid-shaped [X,Y,Z,4] tensor in place and interpolates rgb onto the vertices"""
    import torch
    from sin3dm_amd.encoding.isosurface import marching_cubes
    f, _ = _field("sphere", 36)
    g = torch.from_numpy(f).cuda()
    ax = torch.linspace(0, 1, 36, device="cuda")
    rgb = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1)
    grid4 = torch.cat([g[..., None], rgb], -1).contiguous()
    v0, t0, _ = marching_cubes(g, 0.0, 1.0)
    v, t, col = marching_cubes(grid4, 0.0, 1.0, n_attr=3)
    assert torch.equal(v, v0) and torch.equal(t, t0)
    assert torch.allclose(col, v / 35.0, atol=1e-5)                # colour field = normalised position


@pytest.mark.gpu
def test_hip_full_size_grid():
    """512 x 512 x 256 (BASELINE config 5's decode grid): closed surface of the right measure, in well under a second"""
    import time
    import torch
    from sin3dm_amd.encoding.isosurface import marching_cubes
    X, Y, Z = 512, 512, 256
    ax = [torch.linspace(-1, 1, n, device="cuda") * s for n, s in ((X, 1.0), (Y, 1.0), (Z, 0.5))]
    x, y, z = torch.meshgrid(*ax, indexing="ij")
    f = (torch.sqrt(x * x + y * y + (2 * z) ** 2) - 0.8).contiguous()
    marching_cubes(f[:64, :64, :64].contiguous(), 0.0, 1.0)          # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v, t, _ = marching_cubes(f, 0.0, 1.0)
    dt = time.perf_counter() - t0
    assert dt < 1.0, dt
    h = torch.tensor([2.0 / (X - 1), 2.0 / (Y - 1), 1.0 / (Z - 1)], device="cuda")
    p = (v * h - torch.tensor([1.0, 1.0, 0.5], device="cuda")).double()
    a, b, c = p[t[:, 0].long()], p[t[:, 1].long()], p[t[:, 2].long()]
    vol = float((a * torch.cross(b, c, dim=1)).sum() / 6)
    assert abs(vol - 4 / 3 * np.pi * 0.8 * 0.8 * 0.4) < 2e-3 * 4 / 3 * np.pi * 0.8 * 0.8 * 0.4     # ellipsoid volume
    e = torch.cat([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]).long()
    key = e[:, 0] * (len(v) + 1) + e[:, 1]
    rev = e[:, 1] * (len(v) + 1) + e[:, 0]
    assert len(torch.unique(key)) == len(key) and torch.equal(torch.sort(key)[0], torch.sort(rev)[0])   # closed, oriented


def _two_blobs_mesh(oracle, n=26):
    f, _ = _field("two", n)
    return oracle.marching_cubes(f, 0.0, 1.0)


def test_oracle_components(oracle):
    v, t = _two_blobs_mesh(oracle)
    lab = oracle.mesh_components(t, len(v))
    roots = np.unique(lab)
    assert len(roots) == 2 and all(lab[r] == r for r in roots)                 # a root is its own label
    assert np.array_equal(lab[t[:, 0]], lab[t[:, 1]]) and np.array_equal(lab[t[:, 0]], lab[t[:, 2]])
    for r in roots:
        assert r == np.flatnonzero(lab == r).min()                             # label = smallest vertex index


@pytest.mark.gpu
def test_hip_components_and_largest(oracle):
    import torch
    from sin3dm_amd.encoding.isosurface import largest_component, mesh_components
    v, t = _two_blobs_mesh(oracle, 40)
    lab = mesh_components(torch.from_numpy(t).cuda(), len(v))
    assert np.array_equal(lab.cpu().numpy(), oracle.mesh_components(t, len(v)))
    # a long thin strip: the label has to travel 5000 triangles (pointer jumping keeps the iteration count low)
    n = 5000
    strip = np.stack([np.arange(n), np.arange(n) + 1, np.arange(n) + 2], 1).astype(np.int32)
    lab2 = mesh_components(torch.from_numpy(strip).cuda(), n + 2)
    assert int(lab2.max()) == 0
    vv, tt, _ = largest_component(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda())
    ref = oracle.mesh_components(t, len(v))
    roots, counts = np.unique(ref[t[:, 0]], return_counts=True)
    best = roots[np.argmax(counts)]
    assert len(vv) == int((ref == best).sum()) and len(tt) == int(counts.max())
    closed, chi = _invariants(vv.cpu().numpy(), tt.cpu().numpy())
    assert closed and chi == 2                                                   # one closed blob is left


# ------------------------------------------------------------------ table-free check (oracle/mc_independent.py)
def _independent():
    import os, sys
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import mc_independent
    return mc_independent


@pytest.mark.parametrize("kind,pad", [("sphere", 1.0), ("two", 1.0), ("noise", 1.0), ("noise", None), ("smooth_noise", 1.0)])
def test_oracle_mesh_against_table_free_definition(oracle, kind, pad):
    """The C restatement (which shares the generated case table with the HIP kernel) against what marching cubes fixes
    without any table: the vertex set and positions, one cell per triangle, the segments on every cell face
    (unambiguous faces: the one possible segment; ambiguous faces: two segments, same pairing from both sides),
    outward orientation.  What this cannot see is which of the two pairings PyMCubes' table picks on an ambiguous face."""
    mi = _independent()
    rng = np.random.Generator(np.random.PCG64(12))
    if kind == "noise":
        f = rng.standard_normal((9, 8, 7)).astype(np.float32)
    elif kind == "smooth_noise":
        c = rng.standard_normal((4, 4, 4))
        ax = [np.linspace(0, 3, n) for n in (13, 11, 12)]
        from scipy.ndimage import map_coordinates
        f = map_coordinates(c, np.meshgrid(*ax, indexing="ij"), order=3, mode="nearest").astype(np.float32)
    else:
        f, _ = _field(kind, 14)
    v, t = oracle.marching_cubes(f, 0.0, pad)
    stats = mi.check_mesh(f, v, t, 0.0, pad)
    assert stats["triangles"] == len(t) > 0
    if kind == "noise":
        assert stats["ambiguous_faces"] > 50                     # white noise exercises the ambiguous faces
    if kind == "sphere":
        n = f.shape[0]
        scale = (2.0 / (n - 1)) * np.asarray([1, 0.9, 1.1])
        area, vol = mi.surface_measure(v * scale, t)
        assert abs(vol - 4 / 3 * np.pi * 0.343) < 0.05 * 4 / 3 * np.pi * 0.343 and area > 0


@pytest.mark.gpu
@pytest.mark.parametrize("pad", [1.0, None])
def test_hip_mesh_against_table_free_definition(pad):
    """The same table-free check applied to the DEVICE mesh directly (no C restatement, no shared table in the loop)."""
    import torch
    from sin3dm_amd.encoding.isosurface import marching_cubes
    mi = _independent()
    rng = np.random.Generator(np.random.PCG64(13))
    for f in (rng.standard_normal((10, 7, 9)).astype(np.float32), _field("two", 16)[0]):
        v, t, _ = marching_cubes(torch.from_numpy(f).cuda(), 0.0, pad)
        stats = mi.check_mesh(f, v.cpu().numpy(), t.cpu().numpy(), 0.0, pad)
        assert stats["triangles"] == len(t) > 0
