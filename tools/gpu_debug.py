#!/usr/bin/env python3
"""Prints relative errors of every leaf kernel and UNet case against the golden vectors (no asserts).
S3D_CONV_IMPL=naive switches the convolutions to the plain direct kernel for triangulation."""
import os, sys, traceback
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import golden, relerr
from sin3dm_amd import testing as T, ops
import test_hip_parity as P

def run(name, fn):
    try:
        print(f"{name:40s} {fn()}")
    except Exception as e:
        print(f"{name:40s} EXC {type(e).__name__}: {e}")
        traceback.print_exc()

print("conv impl:", os.environ.get("S3D_CONV_IMPL", "mfma"), torch.cuda.get_device_name(0))
g = golden("leaves")
for tag, C in (("a", 32), ("b", 64)):
    fm = [P.cu(g[f"{tag}.in_{p}"]) for p in T.PLANES]
    gam = [torch.from_numpy(T.synthetic_tensor(f"0.norm_{p}.weight", (C,), 1)) for p in T.PLANES]
    bet = [torch.from_numpy(T.synthetic_tensor(f"0.norm_{p}.bias", (C,), 1)) for p in T.PLANES]
    run(f"{tag} normsilu", lambda: [f"{relerr(y.cpu().numpy(), g[f'{tag}.normsilu_{p}']):.2e}" for p, y in zip(T.PLANES, ops.triplane_norm_silu(fm, gam, bet))])
    for name, k, roll, cout in (("conv3", 3, False, 48), ("conv1", 1, False, 40), ("conv3r", 3, True, 48)):
        ws = [torch.from_numpy(T.synthetic_tensor(f"conv_{p}.weight", (cout, C * 3 if roll else C, k, k), 2)) for p in T.PLANES]
        bs = [torch.from_numpy(T.synthetic_tensor(f"conv_{p}.bias", (cout,), 2)) for p in T.PLANES]
        run(f"{tag} {name}", lambda: [f"{relerr(y.cpu().numpy(), g[f'{tag}.{name}_{p}']):.2e}" for p, y in zip(T.PLANES, ops.triplane_conv(fm, ws, bs, roll))])
    run(f"{tag} down", lambda: [f"{relerr(y.cpu().numpy(), g[f'{tag}.down_{p}']):.2e}" for p, y in zip(T.PLANES, ops.triplane_downsample2x(fm))])
    run(f"{tag} up", lambda: [f"{relerr(y.cpu().numpy(), g[f'{tag}.up_{p}']):.2e}" for p, y in zip(T.PLANES, ops.triplane_resize(fm, [(2*f.shape[-2], 2*f.shape[-1]) for f in fm]))])
    run(f"{tag} resize", lambda: [f"{relerr(y.cpu().numpy(), g[f'{tag}.resize_{p}']):.2e}" for p, y in zip(T.PLANES, ops.triplane_resize(fm, [(2*f.shape[-2]+1, 2*f.shape[-1]+1) for f in fm]))])
gu = golden("unet_fwd")
for tag, mc, raw, ssn, cm in P.UNET_CASES:
    def f():
        H, W, D = (int(v) for v in gu[f"{tag}.hwd"])
        m = P.make_model(mc, raw, ssn, cm)
        with torch.no_grad():
            y = m(P.cu(gu[f"{tag}.x"]), P.cu(gu[f"{tag}.t"]), H=H, W=W, D=D).cpu().numpy()
        return f"{relerr(y, gu[f'{tag}.y']):.2e} corner0={bool(np.all(y[..., H:, W:] == 0))}"
    run(f"unet {tag}", f)
