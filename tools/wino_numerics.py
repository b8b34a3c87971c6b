#!/usr/bin/env python3
"""fp32 error of the Winograd variants considered for the 3x3 TriplaneConv, against an fp64 direct convolution
("F(4x4,3x3) experiment, numerics-gated").  numpy on the CPU: transforms and the channel accumulation in
fp32, weights transformed in double and rounded once (as the packers do).  relerr = max|a-b| / max|b|, the measure of the
parity tests (TOL_OP 2e-5 per leaf, TOL_FWD 1e-4 per forward).

    python tools/wino_numerics.py > profiles/r02_wino_numerics.txt
"""
import numpy as np

f32 = np.float32
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
               dtype=np.float64)


def cook_toom(points, m, r=3):
    """A^T, G for F(m, r) on the given finite points (+ infinity); B^T by solving the bilinear identity."""
    n = m + r - 1
    a = list(points)
    AT = np.array([[(a[j] ** i if j < n - 1 else (1.0 if i == m - 1 else 0.0)) for j in range(n)] for i in range(m)])
    Fd = np.array([np.prod([a[i] - a[j] for j in range(n - 1) if j != i]) for i in range(n - 1)])
    G = np.zeros((n, r))
    for i in range(n - 1):
        for k in range(r):
            G[i, k] = a[i] ** k / Fd[i]
    G[n - 1, r - 1] = 1.0
    rows, rhs = [], []
    for i in range(m):
        for k in range(r):
            for l in range(n):
                row = np.zeros((n, n))
                row[:, l] = AT[i, :] * G[:, k]
                rows.append(row.ravel()); rhs.append(1.0 if l == i + k else 0.0)
    BT = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0].reshape(n, n)
    return AT, G, BT


def conv_ref(x, w):
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
    y = np.zeros((w.shape[0], H, W))
    for kh in range(3):
        for kw in range(3):
            y += np.einsum('oc,chw->ohw', w[:, :, kh, kw], xp[:, kh:kh + H, kw:kw + W])
    return y


def conv_direct32(x, w):
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2), f32); xp[:, 1:-1, 1:-1] = x
    y = np.zeros((w.shape[0], H, W), f32)
    for c0 in range(0, C, 8):
        for kh in range(3):
            for kw in range(3):
                y += np.einsum('oc,chw->ohw', w[:, c0:c0 + 8, kh, kw], xp[c0:c0 + 8, kh:kh + H, kw:kw + W]).astype(f32)
    return y


def conv_wino32(x, w, ATr, Gr, BTr, mr, ATc, Gc, BTc, mc):
    nr, nc = mr + 2, mc + 2
    C, H, W = x.shape
    Co = w.shape[0]
    th, tw = (H + mr - 1) // mr, (W + mc - 1) // mc
    xp = np.zeros((C, th * mr + 2, tw * mc + 2), f32); xp[:, 1:H + 1, 1:W + 1] = x
    U = np.einsum('ik,ockl,jl->ocij', Gr, w.astype(np.float64), Gc).astype(f32)
    d = np.zeros((C, th, tw, nr, nc), f32)
    for i in range(th):
        for j in range(tw):
            d[:, i, j] = xp[:, i * mr:i * mr + nr, j * mc:j * mc + nc]
    V = np.einsum('ik,ctukl->ctuil', BTr.astype(f32), d).astype(f32)
    V = np.einsum('ctuil,jl->ctuij', V, BTc.astype(f32)).astype(f32)
    M = np.zeros((Co, th, tw, nr, nc), f32)
    for c0 in range(0, C, 8):
        M += np.einsum('ocij,ctuij->otuij', U[:, c0:c0 + 8], V[c0:c0 + 8]).astype(f32)
    Y = np.einsum('ik,otukl->otuil', ATr.astype(f32), M).astype(f32)
    Y = np.einsum('otuil,jl->otuij', Y, ATc.astype(f32)).astype(f32)
    y = np.zeros((Co, th * mr, tw * mc), f32)
    for i in range(th):
        for j in range(tw):
            y[:, i * mr:(i + 1) * mr, j * mc:(j + 1) * mc] = Y[:, i, j]
    return y[:, :H, :W]


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


if __name__ == "__main__":
    AT4m, G4m, BT4m = cook_toom([0, 1, -1, 2, -0.5], 4)
    schemes = [("direct fp32", None),
               ("F(2x2)  [k_conv_wino4/2]", (AT2, G2, BT2, 2, AT2, G2, BT2, 2)),
               ("F(2x4) points 0,+-1,+-2  [k_conv_wino24s/24: the default]", (AT2, G2, BT2, 2, AT4, G4, BT4, 4)),
               ("F(2x4) points 0,+-1,2,-1/2", (AT2, G2, BT2, 2, AT4m, G4m, BT4m, 4)),
               ("F(4x4) points 0,+-1,+-2", (AT4, G4, BT4, 4, AT4, G4, BT4, 4)),
               ("F(4x4) points 0,+-1,2,-1/2", (AT4m, G4m, BT4m, 4, AT4m, G4m, BT4m, 4))]
    print(__doc__)
    np.random.seed(1)
    for (C, Co, S) in ((128, 128, 48), (384, 128, 48), (256, 256, 32)):
        x = np.random.randn(C, S, S).astype(f32)
        x = (x / (1 + np.exp(-x))).astype(f32)                   # SiLU of a normal: what the convolutions see
        w = (np.random.randn(Co, C, 3, 3) * 0.02).astype(f32)
        ref = conv_ref(x.astype(np.float64), w.astype(np.float64))
        print(f"layer {C:3d} -> {Co:3d} channels, {S}x{S} pixels")
        for name, sch in schemes:
            y = conv_direct32(x, w) if sch is None else conv_wino32(x, w, *sch)
            print(f"    {name:62s} {relerr(y, ref):.2e}")
    print("""
Reading: F(2x4) with the standard points costs 3-4x the rounding error of F(2x2) (1e-6 against 3e-7) for 25 % fewer
multiplications; it is what the forward now runs, and every GPU gate held unchanged (leaf 2e-5, forward 1e-4, trajectory 2e-4,
full-size chain below).  F(4x4) (44 % fewer multiplications than F(2x2)) would still pass the gates on this evidence
(4-5e-6 per layer) but needs 36 frequencies: with 32x32 accumulator tiles that is six waves per tile or 144 accumulator
registers per wave — neither fits the three-blocks-per-CU structure that keeps the matrix pipe fed at batch 1 (DESIGN.md
section 3.0); not built.  The mixed-point variants halve the error again at the price of dense transform matrices.""")
