"""TEST INFRASTRUCTURE — a table-free, independent check of an iso-surface mesh (marching cubes, Lorensen & Cline 1987;
the reference calls `mcubes.marching_cubes`, src/encoding/utils3d.py:196-203).

PyMCubes (the reference's pinned dependency, requirements.txt: PyMCubes) is not installed here and its source is not in
/root/reference, so the build's 256-case table (tools/gen_mc_tables.py -> s3d_mc_tables.h, shared by the HIP kernel and
by the C restatement) cannot be compared with PyMCubes' table triangle by triangle: that part of the parity stays
UNPINNED.  What this module pins WITHOUT any case table — it never imports or reads s3d_mc_tables.h — is everything the
marching-cubes definition itself fixes:

  1. the vertex set: one vertex on every grid edge whose end values straddle the iso level, at the linear interpolation
     p = p1 + (iso - v1) / (v2 - v1) * (p2 - p1)   (identical for every correct implementation, PyMCubes included);
  2. every triangle lives in one cell (its three vertices lie on edges of that cell);
  3. on every cell face with exactly TWO crossed edges (an unambiguous face) the cell's triangles leave exactly one
     boundary segment, joining those two crossings; a face with four crossed edges (ambiguous) carries exactly two
     segments that pair the four crossings, and the neighbouring cell across that face uses the same pairing;
     a face without crossings carries none.  Together with (2) this fixes the surface in every cell up to the
     triangulation of its polygons and the choice on ambiguous faces — the only freedom case tables have;
  4. orientation: every triangle's normal points from the inside (value < iso) towards the outside;
  5. `surface_measure`: area and enclosed volume, for comparison with analytic fields.

Pure numpy with python loops over cells: for grids of a few thousand cells only.
"""
from collections import Counter, defaultdict

import numpy as np


def _padded(grid, pad_value):
    g = np.asarray(grid, np.float64)
    return np.pad(g, 1, mode="constant", constant_values=pad_value) if pad_value is not None else g


def expected_vertices(grid, iso=0.0, pad_value=1.0):
    """{(axis, i, j, k): xyz} for every crossed grid edge (i,j,k) -> (i,j,k)+e_axis, coordinates in the index frame of the
    UNPADDED grid (the reference subtracts the padding, utils3d.py:203)."""
    g = _padded(grid, pad_value)
    off = 1.0 if pad_value is not None else 0.0
    out = {}
    for axis in range(3):
        a = np.moveaxis(g, axis, 0)
        v1, v2 = a[:-1], a[1:]
        cross = (v1 < iso) != (v2 < iso)
        for idx in np.argwhere(cross):
            i = [0, 0, 0]
            rest = [k for k in range(3) if k != axis]
            i[axis], i[rest[0]], i[rest[1]] = int(idx[0]), int(idx[1]), int(idx[2])
            mu = (iso - v1[tuple(idx)]) / (v2[tuple(idx)] - v1[tuple(idx)])
            p = np.array(i, np.float64)
            p[axis] += mu
            out[(axis, i[0], i[1], i[2])] = p - off
    return out


def _edge_of_vertex(p, off):
    """Grid edge a mesh vertex sits on: the one coordinate with a fractional part names the axis."""
    q = np.asarray(p, np.float64) + off
    frac = np.abs(q - np.round(q))
    axis = int(np.argmax(frac))
    base = np.round(q).astype(int)
    base[axis] = int(np.floor(q[axis]))
    return (axis, int(base[0]), int(base[1]), int(base[2]))


def check_mesh(grid, verts, tris, iso=0.0, pad_value=1.0, tol=2e-5):
    """Raises AssertionError with a description if the mesh violates any of (1)-(4); returns a dict of counts."""
    g = _padded(grid, pad_value)
    off = 1.0 if pad_value is not None else 0.0
    want = expected_vertices(grid, iso, pad_value)
    verts = np.asarray(verts, np.float64)
    tris = np.asarray(tris, np.int64)
    # (1) vertex set
    keys = [_edge_of_vertex(p, off) for p in verts]
    assert len(set(keys)) == len(keys), "two mesh vertices on one grid edge"
    assert set(keys) == set(want), (f"vertex set differs: {len(set(keys) - set(want))} unexpected, "
                                    f"{len(set(want) - set(keys))} missing")
    err = max((np.abs(verts[i] - want[k]).max() for i, k in enumerate(keys)), default=0.0)
    assert err <= tol, f"vertex position off by {err}"
    # (2) one cell per triangle.  A triangle whose three vertices lie on ONE cell face (a fan triangle of a polygon that
    # crosses an ambiguous face twice) fits both cells sharing that face: it is kept aside and attributed per face below.
    X, Y, Z = g.shape
    proper = defaultdict(list)
    tied = defaultdict(list)
    for t in tris:
        cells = None
        for v in t:
            axis, i, j, k = keys[v]
            base = [i, j, k]
            cand = set()
            rest = [a for a in range(3) if a != axis]
            for da in (0, -1):
                for db in (0, -1):
                    c = list(base)
                    c[rest[0]] += da
                    c[rest[1]] += db
                    if all(0 <= c[a] < g.shape[a] - 1 for a in range(3)):
                        cand.add(tuple(c))
            cells = cand if cells is None else cells & cand
        assert cells, f"triangle {t.tolist()} does not fit into one cell"
        assert len(cells) <= 2
        if len(cells) == 1:
            proper[min(cells)].append(t)
        else:
            tied[frozenset(cells)].append(t)

    def edges_of(t):
        return [frozenset((int(t[0]), int(t[1]))), frozenset((int(t[1]), int(t[2]))), frozenset((int(t[2]), int(t[0])))]

    # (3) per face: the segments the cells' triangles leave on it
    kid = {k: i for i, k in enumerate(keys)}
    inside = g < iso
    cell_boundary = {}
    for cell, ts in proper.items():
        und = Counter(e for t in ts for e in edges_of(t))
        assert all(c <= 2 for c in und.values()), f"cell {cell}: an edge is used by more than two triangles"
        cell_boundary[cell] = {e for e, c in und.items() if c == 1}
    n_amb = 0
    claimed = defaultdict(set)
    for axis in range(3):
        r = [a for a in range(3) if a != axis]
        n = list(g.shape)
        for fpos in range(n[axis]):
            for u in range(n[r[0]] - 1):
                for w in range(n[r[1]] - 1):
                    fe = []                          # crossed edges of this face, as vertex ids
                    for (ea, db) in ((r[0], 0), (r[0], 1), (r[1], 0), (r[1], 1)):
                        other = r[1] if ea == r[0] else r[0]
                        b = [0, 0, 0]
                        b[axis] = fpos
                        b[r[0]], b[r[1]] = u, w
                        b[other] += db
                        e2 = list(b)
                        e2[ea] += 1
                        if inside[tuple(b)] != inside[tuple(e2)]:
                            fe.append(kid[(ea, b[0], b[1], b[2])])
                    sides = []
                    for d in (-1, 0):
                        c = [0, 0, 0]
                        c[axis] = fpos + d
                        c[r[0]], c[r[1]] = u, w
                        if 0 <= c[axis] < n[axis] - 1:
                            sides.append(tuple(c))
                    vs = set(fe)
                    segs = []
                    for c in sides:
                        sg = {e for e in cell_boundary.get(c, ()) if e <= vs}
                        claimed[c] |= sg
                        segs.append(sg)
                    flat = tied.get(frozenset(sides), []) if len(sides) == 2 else []
                    flat = [t for t in flat if set(int(x) for x in t) <= vs]

                    def valid(sg):
                        if len(fe) == 0:
                            return not sg
                        if len(fe) == 2:
                            return sg == {frozenset(fe)}
                        return len(fe) == 4 and len(sg) == 2 and set().union(*sg) == vs
                    ok = False
                    for mask in range(1 << len(flat)):
                        cur = [set(s) for s in segs]
                        for i, t in enumerate(flat):
                            for e in edges_of(t):
                                cur[(mask >> i) & 1] ^= {e}
                        if all(valid(c) for c in cur) and (len(cur) < 2 or cur[0] == cur[1]):
                            ok = True
                            break
                    assert ok, (f"face axis {axis} at {fpos},{u},{w}: {len(fe)} crossings, segments {segs}, in-face triangles "
                                f"{[t.tolist() for t in flat]}: not the single possible segment / not one consistent pairing")
                    n_amb += len(fe) == 4
    for cell, bd in cell_boundary.items():
        assert claimed[cell] == bd, f"cell {cell}: boundary segments off the cell's faces"
    by_cell = proper
    # (4) orientation: normal . gradient > 0 (gradient by central differences of the trilinear field at the centroid)
    bad = 0
    for t in tris:
        p = verts[t] + off
        n = np.cross(p[1] - p[0], p[2] - p[0])
        c = p.mean(0)
        grad = np.array([_tri(g, c + d) - _tri(g, c - d) for d in 0.05 * np.eye(3)])
        if np.dot(n, grad) < 0 and np.linalg.norm(n) > 1e-9 and np.linalg.norm(grad) > 1e-9:
            bad += 1
    assert bad <= 0.02 * len(tris), f"{bad} of {len(tris)} triangles face inwards"     # flat saddle triangles may have grad . n ~ 0
    return {"vertices": len(verts), "triangles": len(tris), "cells": len(by_cell), "ambiguous_faces": n_amb}


def _tri(g, p):
    p = np.clip(p, 0, np.array(g.shape) - 1.000001)
    i = np.floor(p).astype(int)
    f = p - i
    v = 0.0
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                w = (f[0] if dx else 1 - f[0]) * (f[1] if dy else 1 - f[1]) * (f[2] if dz else 1 - f[2])
                v += w * g[i[0] + dx, i[1] + dy, i[2] + dz]
    return v


def surface_measure(verts, tris):
    """(area, enclosed volume) of a closed oriented triangle mesh."""
    v = np.asarray(verts, np.float64)
    a, b, c = v[tris[:, 0]], v[tris[:, 1]], v[tris[:, 2]]
    return (0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum(), np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
