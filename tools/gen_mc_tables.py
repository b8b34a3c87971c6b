#!/usr/bin/env python3
"""Generate the marching-cubes case table used by sin3dm_amd/csrc/s3d_mc.hip and oracle/sin3dm_oracle.c
(-> sin3dm_amd/csrc/s3d_mc_tables.h).  The table is CONSTRUCTED here, not copied: no marching-cubes source exists
in this environment (PyMCubes, scikit-image and the reference's own dependency are absent), so the 256 cases are
derived from first principles:

  * corner c of a cell sits at (c&1, (c>>1)&1, (c>>2)&1); a corner is "inside" when its value is below the iso level;
  * an edge is active when exactly one endpoint is inside; every active edge carries one surface vertex;
  * on each of the six faces the active edges are joined pairwise.  Two active edges: join them.  Four (the two
    inside corners of the face are diagonal): join the edges around each INSIDE corner (the inside corners stay
    separated).  The rule only looks at the four corner signs of the face, so both cells sharing a face cut it the
    same way and the mesh is watertight;
  * every active edge lies on two faces, so the joins form closed loops; each loop is fan-triangulated and oriented
    so that its normal points from inside to outside.

Interior ambiguities are ignored, exactly as in the classic algorithm.  Result: at most 5 triangles per case (asserted).
"""
import itertools
import os

CORNER = [((c & 1), (c >> 1) & 1, (c >> 2) & 1) for c in range(8)]
# 12 edges: pairs of corners differing in one coordinate, ordered by (axis, then the other two coordinates)
EDGES = []
for axis in range(3):
    for c in range(8):
        if CORNER[c][axis] == 0:
            d = c | (1 << axis)
            EDGES.append((c, d))
EDGE_AXIS = [0] * 4 + [1] * 4 + [2] * 4
# faces: (axis, side) -> its four corners in cyclic order
FACES = []
for axis in range(3):
    for side in (0, 1):
        cs = [c for c in range(8) if CORNER[c][axis] == side]
        a, b = [k for k in range(3) if k != axis]
        key = {(0, 0): 0, (1, 0): 1, (1, 1): 2, (0, 1): 3}
        cs.sort(key=lambda c: key[(CORNER[c][a], CORNER[c][b])])
        FACES.append(cs)


def edge_between(c, d):
    for i, (a, b) in enumerate(EDGES):
        if {a, b} == {c, d}:
            return i
    raise KeyError((c, d))


def mid(e):
    a, b = EDGES[e]
    return [(CORNER[a][k] + CORNER[b][k]) / 2 for k in range(3)]


def case_triangles(case):
    inside = [(case >> c) & 1 for c in range(8)]
    adj = {}
    for f in FACES:
        fe = [edge_between(f[i], f[(i + 1) % 4]) for i in range(4)]          # edge i joins corner i and i+1
        act = [i for i in range(4) if inside[f[i]] != inside[f[(i + 1) % 4]]]
        if len(act) == 2:
            pairs = [(fe[act[0]], fe[act[1]])]
        elif len(act) == 4:
            # join the two edges that meet at each inside corner: corner i touches edges i-1 and i
            pairs = [(fe[(i - 1) % 4], fe[i]) for i in range(4) if inside[f[i]]]
        else:
            assert not act
            pairs = []
        for a, b in pairs:
            adj.setdefault(a, []).append(b)
            adj.setdefault(b, []).append(a)
    assert all(len(v) == 2 for v in adj.values()), (case, adj)
    loops, seen = [], set()
    for start in sorted(adj):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [n for n in adj[cur] if n != prev]
            nxt = nxt[0] if nxt else adj[cur][0]
            if nxt == start:
                break
            loop.append(nxt); seen.add(nxt)
            prev, cur = cur, nxt
        loops.append(loop)
    tris = []
    for loop in loops:
        # orientation: the polygon normal must point away from the inside.  Reference point: the inside endpoint of
        # the first edge; reference direction: from that corner to the polygon centroid.
        pts = [mid(e) for e in loop]
        cen = [sum(p[k] for p in pts) / len(pts) for k in range(3)]
        nrm = [0.0, 0.0, 0.0]
        for i in range(len(pts)):                                               # Newell normal
            p, q = pts[i], pts[(i + 1) % len(pts)]
            nrm[0] += (p[1] - q[1]) * (p[2] + q[2]); nrm[1] += (p[2] - q[2]) * (p[0] + q[0]); nrm[2] += (p[0] - q[0]) * (p[1] + q[1])
        ins = [CORNER[c] for e in loop for c in EDGES[e] if inside[c]]
        ref = [sum(p[k] for p in ins) / len(ins) for k in range(3)]
        out = [cen[k] - ref[k] for k in range(3)]
        if sum(nrm[k] * out[k] for k in range(3)) < 0:
            loop = loop[::-1]
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    return tris


def main():
    table = [case_triangles(c) for c in range(256)]
    assert max(len(t) for t in table) <= 5, max(len(t) for t in table)
    assert table[0] == [] and table[255] == []
    # complementary cases cut the same edges
    for c in range(256):
        assert sorted(set(e for t in table[c] for e in t)) == sorted(set(e for t in table[255 - c] for e in t))
    import sys
    # default: the product's table; with a path argument: another copy (oracle/Makefile builds the CHECKER's own, so that the C
    # oracle never reads a file of the product tree)
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sin3dm_amd", "csrc", "s3d_mc_tables.h")
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as f:
        f.write("// Generated by tools/gen_mc_tables.py (constructed from first principles, see there). Do not edit.\n")
        f.write("// Corner c = (c&1, (c>>1)&1, (c>>2)&1); edge e joins corners MC_EDGE[e][0..1] along axis MC_EDGE_AXIS[e].\n")
        f.write("#pragma once\n")
        f.write("static const unsigned char MC_EDGE[12][2] = {" + ", ".join("{%d,%d}" % e for e in EDGES) + "};\n")
        f.write("static const unsigned char MC_EDGE_AXIS[12] = {" + ", ".join(map(str, EDGE_AXIS)) + "};\n")
        f.write("static const unsigned char MC_NTRI[256] = {" + ", ".join(str(len(t)) for t in table) + "};\n")
        f.write("static const signed char MC_TRI[256][15] = {\n")
        for t in table:
            flat = [e for tri in t for e in tri] + [-1] * (15 - 3 * len(t))
            f.write("    {" + ", ".join("%2d" % v for v in flat) + "},\n")
        f.write("};\n")
    print(out, "max triangles", max(len(t) for t in table), "total", sum(len(t) for t in table))


if __name__ == "__main__":
    main()
