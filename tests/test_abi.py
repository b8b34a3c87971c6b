"""CPU-only: the C-ABI library loads and exports every symbol include/sin3dm_hip.h declares; the host
logic (schedules, respacing, parameter manifest, layout helpers) matches the golden vectors."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO, golden
from sin3dm_amd import _lib
from sin3dm_amd import testing as T


def _declared_symbols():
    txt = open(os.path.join(REPO, "include", "sin3dm_hip.h")).read()
    return sorted(set(re.findall(r"S3D_API\s+[\w\s\*]+?\b(s3d_\w+)\s*\(", txt)))


def test_header_symbols_exported():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sin3dm_hip.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert lib.s3d_abi_version() == _lib.ABI_VERSION


def test_param_manifest_from_library():
    """The library's own parameter registry equals the reference's state_dict manifest."""
    import ctypes as C
    lib = _lib.load()
    g = golden("unet_manifest")
    for mc in (32, 64, 128):
        cfg = _lib.UNetCfg(12, mc, 12, 1, 2, (C.c_int32 * 8)(1, 2), 1, 1)
        h = C.c_void_p()
        assert lib.s3d_unet_create(C.byref(cfg), C.byref(h)) == 0
        got = {}
        for i in range(lib.s3d_unet_num_params(h)):
            name, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            assert lib.s3d_unet_param_info(h, i, C.byref(name), shape, C.byref(nd)) == 0
            got[name.value.decode()] = tuple(shape[k] for k in range(nd.value))
        lib.s3d_unet_destroy(h)
        ref = {k.split("/", 1)[1]: tuple(int(i) for i in g[k]) for k in g.files if k.startswith(f"mc{mc}/")}
        assert got == ref


def test_create_rejects_unconstructible_configs():
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.UNetCfg(12, 64, 12, 2, 2, (C.c_int32 * 8)(1, 2), 1, 1)       # num_res_blocks=2
    assert lib.s3d_unet_create(C.byref(cfg), C.byref(h)) == _lib.ERR_UNSUPPORTED
    assert b"num_res_blocks" in lib.s3d_last_error()
    cfg = _lib.UNetCfg(12, 48, 12, 1, 2, (C.c_int32 * 8)(1, 2), 1, 1)       # GroupNorm(32, 48) impossible
    assert lib.s3d_unet_create(C.byref(cfg), C.byref(h)) == _lib.ERR_INVALID


def test_schedules_match_reference():
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    g = golden("schedules")
    for tag, resp in (("full", ""), ("r100", "100"), ("r10", "10"), ("ddim50", "ddim50"), ("r20", "20")):
        d = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=True, timestep_respacing=resp)
        assert np.array_equal(np.asarray(d.timestep_map), g[f"{tag}.timestep_map"])
        for f in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod"):
            np.testing.assert_allclose(getattr(d, f), g[f"{tag}.{f}"], rtol=1e-13, atol=0, err_msg=f"{tag}.{f}")


def test_compose_decompose():
    from sin3dm_amd.utils.triplane_util import compose_featmaps, decompose_featmaps
    g = golden("compose")
    comp, hwd = compose_featmaps(*(torch.from_numpy(g[k]) for k in ("xy", "xz", "yz")))
    assert tuple(hwd) == tuple(int(v) for v in g["hwd"])
    assert np.array_equal(comp.numpy(), g["composed"])
    for a, k in zip(decompose_featmaps(comp, hwd), ("xy", "xz", "yz")):
        assert np.array_equal(a.numpy(), g[k])


def test_module_state_dict_names_and_cpu_refusal():
    from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
    m = TriplaneUNetModelSmall(12, 32, 12, use_scale_shift_norm=True)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(T.unet_param_shapes(model_channels=32))
    m.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 0))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):          # no CPU fallback: fail loudly
            m(torch.zeros(1, 12, 16, 20), torch.zeros(1), H=10, W=14, D=6)


def test_sample_args_roundtrip(tmp_path):
    import json
    from sin3dm_amd.utils import parser_util as pu
    tag = tmp_path / "exp"
    args = pu.train_args(["--tag", str(tag), "--data_path", "x.npz", "--model_channels", "128"])
    assert args.in_channels == 12 and args.out_channels == 12
    saved = json.load(open(tag / "diffusion" / "args.json"))
    assert saved["model_channels"] == 128 and saved["channel_mult"] == "1,2" and saved["use_scale_shift_norm"] is True
    s = pu.sample_args(["--tag", str(tag), "--timestep_respacing", "100", "--use_ddim", "True"])
    assert s.model_channels == 128 and s.timestep_respacing == "100" and s.use_ddim is True and s.fdim_geo == 4


@pytest.mark.parametrize("orig,respacing", [(1000, "100"), (200, ""), (300, "50"), (4000, "ddim25"), (777, "")])
def test_host_mapped_timesteps_equal_the_tensor_path(orig, respacing):
    """_WrappedModel maps host-known timesteps on the host; the value must be the float the tensor path (and the reference,
    respace.py:123-128: fp32 tensor x python scalar) produces, also when 1000/original_num_steps is not exact."""
    from sin3dm_amd.diffusion.gaussian_diffusion import HostTimesteps
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    d = create_gaussian_diffusion(steps=orig, timestep_respacing=respacing, rescale_timesteps=True)
    seen = {}

    class Probe:
        def __call__(self, x, ts, **kw):
            seen["ts"] = ts
            return x
    wrapped = d._wrap_model(Probe())
    n = d.num_timesteps
    idx = torch.arange(n)
    wrapped(torch.zeros(n), idx)                                   # tensor path
    via_tensor = seen["ts"].clone()
    for i in (0, 1, n // 3, n - 1):
        wrapped(torch.zeros(2), HostTimesteps(torch.tensor([i, i]), (i, i)))
        assert torch.equal(torch.as_tensor(seen["ts"]).float(), via_tensor[i].expand(2)), (i, seen["ts"], via_tensor[i])
        assert seen["ts"].host_values == (float(via_tensor[i]),) * 2


def test_options_through_the_abi_and_environment_fallback():
    """s3d_set_option / s3d_get_option: the documented way to select kernel forms (the S3D_* environment variables are only the
    fallback until the first call for an option); unknown names are rejected with a message."""
    import subprocess
    import sys
    code = ("import os, sys\n"
            f"sys.path.insert(0, {REPO!r})\n"
            "os.environ['S3D_WINO'] = '4'; os.environ['S3D_CONV_IMPL'] = 'naive'\n"
            "from sin3dm_amd import _lib\n"
            "assert _lib.get_option('WINO') == 4 and _lib.get_option('S3D_CONV_IMPL') == 1        # environment, read at first use\n"
            "assert _lib.get_option('WINO24W') is None and _lib.get_option('BWD_SIDE') is None     # unset: the library chooses\n"
            "_lib.set_option('S3D_WINO', 24); _lib.set_option('WINO24W', '1'); _lib.set_option('CONV_IMPL', 'mfma')\n"
            "assert _lib.get_option('WINO') == 24 and _lib.get_option('WINO24W') == 1 and _lib.get_option('CONV_IMPL') == 0\n"
            "_lib.set_option('WINO24W', None)\n"
            "assert _lib.get_option('WINO24W') is None\n"
            "os.environ['S3D_GN_FUSED'] = '1'; _lib.set_option('GN_FUSED', 0)                      # a call beats the environment\n"
            "assert _lib.get_option('GN_FUSED') == 0\n"
            "try:\n"
            "    _lib.set_option('NO_SUCH_OPTION', 1)\n"
            "    raise SystemExit('accepted an unknown option')\n"
            "except AssertionError as e:\n"
            "    assert 'NO_SUCH_OPTION' in str(e)\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if not k.startswith("S3D_")})
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    names = re.findall(r"^ \*   ([A-Z][A-Z0-9_]+)  ", open(os.path.join(REPO, "include", "sin3dm_hip.h")).read(), flags=re.M)
    assert len(names) == 13                                  # every documented option exists
    for n in names:
        _lib.get_option(n)
