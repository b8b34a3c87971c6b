"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(a) golden vectors captured from the reference and (b) the C oracle on seeded inputs.

Tolerance: north_star demands <= 1e-3 relative fp32 against the reference's PyTorch-CPU path; the gates
below are 1e-4 per forward / trajectory (max|a-b| / max|b|), i.e. 10x tighter.
"""
import numpy as np
import pytest
import torch

from conftest import golden, relerr
from sin3dm_amd import testing as T

pytestmark = pytest.mark.gpu

TOL_OP = 2e-5
TOL_FWD = 1e-4


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def make_model(mc, raw=False, ssn=True, cm=(1, 2)):
    from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall, TriplaneUNetModelSmallRaw
    cls = TriplaneUNetModelSmallRaw if raw else TriplaneUNetModelSmall
    m = cls(12, mc, 12, channel_mult=cm, use_scale_shift_norm=ssn)
    m.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc, rollout=not raw,
                                                                 use_scale_shift_norm=ssn, channel_mult=cm), 0))
    return m.to(dev()).eval()


def make_diffusion(resp):
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    return create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=True, timestep_respacing=resp)


# ------------------------------------------------------------------ leaf kernels vs the reference ops
@pytest.mark.parametrize("tag,C", [("a", 32), ("b", 64)])
def test_leaf_ops(tag, C):
    from sin3dm_amd import ops
    g = golden("leaves")
    fm = [cu(g[f"{tag}.in_{p}"]) for p in T.PLANES]
    gam = [torch.from_numpy(T.synthetic_tensor(f"0.norm_{p}.weight", (C,), 1)) for p in T.PLANES]
    bet = [torch.from_numpy(T.synthetic_tensor(f"0.norm_{p}.bias", (C,), 1)) for p in T.PLANES]
    for p, y in zip(T.PLANES, ops.triplane_norm_silu(fm, gam, bet)):
        assert relerr(y.cpu().numpy(), g[f"{tag}.normsilu_{p}"]) < TOL_OP, p
    for name, k, roll, cout in (("conv3r", 3, True, 48), ("conv3", 3, False, 48), ("conv1", 1, False, 40)):
        ws = [torch.from_numpy(T.synthetic_tensor(f"conv_{p}.weight", (cout, C * 3 if roll else C, k, k), 2)) for p in T.PLANES]
        bs = [torch.from_numpy(T.synthetic_tensor(f"conv_{p}.bias", (cout,), 2)) for p in T.PLANES]
        for p, y in zip(T.PLANES, ops.triplane_conv(fm, ws, bs, roll)):
            assert relerr(y.cpu().numpy(), g[f"{tag}.{name}_{p}"]) < TOL_OP, (name, p)
    for p, y in zip(T.PLANES, ops.triplane_downsample2x(fm)):
        assert relerr(y.cpu().numpy(), g[f"{tag}.down_{p}"]) < 1e-6
    for p, y in zip(T.PLANES, ops.triplane_resize(fm, [(2 * f.shape[-2], 2 * f.shape[-1]) for f in fm])):
        assert relerr(y.cpu().numpy(), g[f"{tag}.up_{p}"]) < 1e-6
    for p, y in zip(T.PLANES, ops.triplane_resize(fm, [(2 * f.shape[-2] + 1, 2 * f.shape[-1] + 1) for f in fm])):
        assert relerr(y.cpu().numpy(), g[f"{tag}.resize_{p}"]) < 2e-6


def test_conv_edge_shapes(oracle):
    """ragged tiles, 1-pixel planes (edge variant 3), Cout not a multiple of the N tile, and a Cout that is not a
    multiple of 4 (the Winograd epilogue moves channel quads: such a layer takes the direct kernel); TriplaneConv with rollout
    (unet_triplane.py:31-60) against the C oracle's literal dense concat."""
    from sin3dm_amd import ops
    # (the 128- and 256-channel cases go through k_rank1b — whole 128-channel chunks of own channels: a Cout that is not a multiple
    #  of its 32-channel groups, positions that are not a multiple of its 32-position tiles, batch 2 / 3, one and two chunks)
    for (B, C, H, W, D, cout) in ((1, 32, 1, 1, 1, 32), (2, 32, 1, 37, 2, 24), (1, 64, 17, 3, 33, 72), (1, 32, 40, 9, 1, 64),
                                  (1, 32, 9, 12, 5, 30), (2, 128, 9, 12, 5, 40), (3, 256, 33, 20, 17, 96), (1, 128, 1, 35, 2, 32)):
        fm = [T.synthetic_noise(s, 11 + i) for i, s in enumerate(((B, C, H, W), (B, C, H, D), (B, C, W, D)))]
        sd = {}
        for p in T.PLANES:
            sd[f"c.conv_{p}.weight"] = T.synthetic_tensor(f"conv_{p}.weight", (cout, 3 * C, 3, 3), 7)
            sd[f"c.conv_{p}.bias"] = T.synthetic_tensor(f"conv_{p}.bias", (cout,), 7)
        want = oracle.triplane_conv(sd, "c", fm, cout, 3, True)
        got = ops.triplane_conv([cu(f) for f in fm], [torch.from_numpy(sd[f"c.conv_{p}.weight"]) for p in T.PLANES],
                                [torch.from_numpy(sd[f"c.conv_{p}.bias"]) for p in T.PLANES], True)
        for p, a, b in zip(T.PLANES, got, want):
            assert relerr(a.cpu().numpy(), b) < TOL_OP, ((B, C, H, W, D, cout), p)


# ------------------------------------------------------------------ UNet forward vs the reference
UNET_CASES = [("mc32_a", 32, False, True, (1, 2)), ("mc32_odd", 32, False, True, (1, 2)),
              ("mc64_b", 64, False, True, (1, 2)), ("mc32_raw", 32, True, True, (1, 2)),
              ("mc32_add", 32, False, False, (1, 2)), ("mc32_3lev", 32, False, True, (1, 2, 2))]


@pytest.mark.parametrize("tag,mc,raw,ssn,cm", UNET_CASES)
def test_unet_forward_golden(tag, mc, raw, ssn, cm):
    g = golden("unet_fwd")
    H, W, D = (int(v) for v in g[f"{tag}.hwd"])
    model = make_model(mc, raw, ssn, cm)
    with torch.no_grad():
        y = model(cu(g[f"{tag}.x"]), cu(g[f"{tag}.t"]), H=H, W=W, D=D)
    y = y.cpu().numpy()
    assert relerr(y, g[f"{tag}.y"]) < TOL_FWD
    assert np.all(y[..., H:, W:] == 0)


@pytest.mark.parametrize("variant", ["24w", "4", "2", "0", "novcat", "gnsplit", "1x1t", "naive"])
def test_unet_forward_golden_other_conv_kernels(variant):
    """Every 3x3 kernel on the golden planes (24w: S3D_WINO24W=1, the 64-output-channel block k_conv_wino24w of the mixed Winograd
    F(2x4,3x3) kernel forced onto every launch whose widths allow it — by default it only takes launches of several rounds of
    blocks; 4 / 2: F(2x2) with one / two frequency rows per wave; 0: direct MFMA convolution) against the same golden vectors,
    leaf convolutions and ragged shapes included; novcat: S3D_VCAT=0, the upsample + concat materialised instead of the virtual
    concat of Fwd::resblock_cat; gnsplit: S3D_GN_FUSED=0, every GroupNorm statistic from a launch of its own (k_gn_finalize_as) instead of
    being added inside k_gn_act / the output head; 1x1t: S3D_CONV1X1_T=1, the transposed-accumulator epilogue of the 1x1 convolutions (16-byte
    accesses; by default only launches of four rounds of blocks take it) on every plain 1x1 launch; naive: S3D_CONV_IMPL=naive, the
    one-thread-per-output convolutions used for triangulation).  The choices are read once per process, hence the subprocess."""
    import os, subprocess, sys
    code = (
        "import numpy as np, torch, sys\n"
        "sys.path.insert(0, 'tests')\n"
        "from conftest import golden, relerr\n"
        "import conftest, test_hip_parity as tp\n"
        "tp.test_leaf_ops('a', 32); tp.test_leaf_ops('b', 64)\n"
        "sys.path.insert(0, 'oracle'); import oracle as orc; orc.lib(); tp.test_conv_edge_shapes(orc)\n"
        "from test_hip_parity import make_model, cu, UNET_CASES\n"
        "g = golden('unet_fwd')\n"
        "for tag, mc, raw, ssn, cm in UNET_CASES:\n"
        "    H, W, D = (int(v) for v in g[f'{tag}.hwd'])\n"
        "    with torch.no_grad():\n"
        "        y = make_model(mc, raw, ssn, cm)(cu(g[f'{tag}.x']), cu(g[f'{tag}.t']), H=H, W=W, D=D).cpu().numpy()\n"
        "    e = relerr(y, g[f'{tag}.y'])\n"
        "    assert e < 1e-4, (tag, e)\n"
        "print('ok')\n")
    env = {"24w": dict(S3D_WINO24W="1"), "novcat": dict(S3D_VCAT="0"), "1x1t": dict(S3D_CONV1X1_T="1"), "gnsplit": dict(S3D_GN_FUSED="0"),
           "naive": dict(S3D_CONV_IMPL="naive")}.get(variant, dict(S3D_WINO=variant))
    env = dict(os.environ, **env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


R1_CASES = [(32, 2, (10, 14, 6)), (64, 3, (40, 24, 56)), (128, 2, (64, 48, 32)), (128, 1, (128, 128, 128)),
            (128, 3, (96, 80, 64))]      # the last two have more 8x16-pixel tiles than co-resident blocks


def _r1_forward(mc, B, hwd, seed, rank1_name=False):
    H, W, D = hwd
    model = make_model(mc)
    x = cu(T.synthetic_noise((B, 12, H + D, W + D), seed))
    t = torch.arange(B, device=dev(), dtype=torch.float32) * 37.0 + 5.0
    model.profile(1)
    with torch.no_grad():
        ys = [model(x, t, H=H, W=W, D=D) for _ in range(3)]
    model.profile_read()
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    if rank1_name:
        return ys[0].cpu().numpy(), model.profile_kernel(0), model.profile_kernel(2)
    return ys[0].cpu().numpy(), model.profile_kernel(0)


@pytest.mark.parametrize("switch,value,other,default", [("S3D_WINO24W", "1", "k_conv_wino24w", "k_conv_wino24s"),
                                                        ("S3D_WINO24G", "1", "k_conv_wino24g", "k_conv_wino24")])
def test_switched_conv_forms_are_bit_identical_and_reported(tmp_path, switch, value, other, default):
    """The two blockings of the mixed Winograd 3x3 kernel (TriplaneConv, unet_triplane.py:27-58) do the same arithmetic in the
    same order: k_conv_wino24w (8x16 pixels x 64 output channels per block, two n32 sub-blocks sharing one halo + input transform;
    S3D_WINO24W=1 forces it onto every launch whose cout is a multiple of 64) against k_conv_wino24s (x 32; S3D_WINO24W=0), each
    in its own process (the switch is read once).  Whole-UNet outputs are bit-identical — the convolution outputs AND the
    GroupNorm partial sums their epilogues leave — repeated calls agree, and the library reports which kernel ran.
    S3D_WINO24G=1: the LDS-DMA form (k_conv_wino24g: halo by buffer_load ... lds into a swizzled unpadded ring, persistent blocks when a
    launch has more tiles than co-resident blocks — the last two cases) against the library's default choice, same bar."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(tag, env, expect):
        code = (
            "import numpy as np, sys\n"
            "sys.path.insert(0, 'tests')\n"
            "import test_hip_parity as tp\n"
            "for i, (mc, B, hwd) in enumerate(tp.R1_CASES):\n"
            "    y, name = tp._r1_forward(mc, B, hwd, 70 + i)\n"
            f"    assert i < 3 or {expect!r} in name, name\n"
            f"    np.save(r'{tmp_path}/{tag}_' + str(i) + '.npy', y)\n"
            "print('ok')\n")
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

    run("plain", {switch: "0"}, default)
    run("form", {switch: value}, other)
    for i, (mc, B, hwd) in enumerate(R1_CASES):
        a, b = np.load(f"{tmp_path}/plain_{i}.npy"), np.load(f"{tmp_path}/form_{i}.npy")
        assert np.array_equal(a, b), (mc, B, hwd, float(np.abs(a - b).max()))


def test_forms_selected_through_the_abi_in_one_process():
    """s3d_set_option (the documented boundary; the environment variables are its fallback) switches a form between two
    launches of ONE process: both blockings of the mixed Winograd kernel and both GroupNorm-statistics forms give the same bits,
    the library names the kernel that ran, and clearing the option returns the choice to the library."""
    from sin3dm_amd import _lib
    mc, B, hwd = R1_CASES[2]
    try:
        _lib.set_option("WINO24W", 0)
        a, name_a = _r1_forward(mc, B, hwd, 72)
        _lib.set_option("WINO24W", 1)
        b, name_b = _r1_forward(mc, B, hwd, 72)
        assert "k_conv_wino24s" in name_a and "k_conv_wino24w" in name_b and "k_conv_wino24s" not in name_b
        assert np.array_equal(a, b)
        _lib.set_option("GN_FUSED", 0)
        c, _ = _r1_forward(mc, B, hwd, 72)
        _lib.set_option("GN_FUSED", 1)
        d, _ = _r1_forward(mc, B, hwd, 72)
        assert np.array_equal(a, c) and np.array_equal(a, d)
    finally:
        _lib.set_option("WINO24W", None)
        _lib.set_option("GN_FUSED", None)
    assert _lib.get_option("WINO24W") is None
    e, _ = _r1_forward(mc, B, hwd, 72)
    assert np.array_equal(a, e)


def test_groupnorm_statistics_forms_are_bit_identical(tmp_path):
    """GroupNorm32 statistics (nn.py:17-19, 93-100) of a convolution output are the sum of its epilogue's partial records.  Small
    launches add them in the CONSUMER's own blocks (k_gn_act, k_out_head_px: one dependent launch less), launches of many rounds
    of blocks (batch >= 2, big planes) get them from k_gn_finalize_as, which runs the consumer's own summation (same lanes per
    group, same trip and meeting order) once ahead of it.  S3D_GN_FUSED=1 / =0 force either form everywhere, the default chooses
    by launch size: whole-UNet outputs agree bit for bit — so a sample's result does not depend on the batch it is computed in."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tag, val in (("in", "1"), ("ahead", "0")):
        code = (
            "import numpy as np, sys\n"
            "sys.path.insert(0, 'tests')\n"
            "import test_hip_parity as tp\n"
            "for i, (mc, B, hwd) in enumerate(tp.R1_CASES):\n"
            "    y, _ = tp._r1_forward(mc, B, hwd, 90 + i)\n"
            f"    np.save(r'{tmp_path}/{tag}_' + str(i) + '.npy', y)\n"
            "print('ok')\n")
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, S3D_GN_FUSED=val), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    for i, (mc, B, hwd) in enumerate(R1_CASES):
        y, _ = _r1_forward(mc, B, hwd, 90 + i)
        a, b = np.load(f"{tmp_path}/in_{i}.npy"), np.load(f"{tmp_path}/ahead_{i}.npy")
        assert np.array_equal(a, b), (mc, B, hwd, float(np.abs(a - b).max()))
        assert np.array_equal(y, a), (mc, B, hwd)


def test_rank1_forms_are_bit_identical(tmp_path):
    """The rollout tables (unet_triplane.py:37-58) come from k_rank1b when the own channels are whole 128-channel chunks
    (fragment-order weights straight into a register ring, 32 positions x 32 output channels x 3 taps per block, XCD-local weight
    reuse; s3d_conv.hip), otherwise from k_rank1 — with TWO samples per staged weight tile from batch 2 on (s3d_rank1.h, NS = 2;
    an odd batch leaves a one-sample block).  Every form builds each sample's sums in the order of the plain kernel: the default
    against S3D_RANK1_BATCH=0 (k_rank1, one sample per block, everywhere) in a separate process, K slices on in both, bit for bit;
    the library reports which kernels built the tables."""
    import os, subprocess, sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, 'tests')\n"
        "import test_hip_parity as tp\n"
        "for i, (mc, B, hwd) in enumerate(tp.R1_CASES):\n"
        "    y, _, name = tp._r1_forward(mc, B, hwd, 110 + i, rank1_name=True)\n"
        "    assert 'k_rank1<' in name, name\n"
        f"    np.save(r'{tmp_path}/one_' + str(i) + '.npy', y)\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, S3D_RANK1_BATCH="0"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    for i, (mc, B, hwd) in enumerate(R1_CASES):
        y, _, name = _r1_forward(mc, B, hwd, 110 + i, rank1_name=True)
        # (a 64-channel UNet has one 128-channel level: both kernels; 32 channels: 32 / 64 / 96 own channels, k_rank1 only)
        assert ("k_rank1b" in name) == (mc >= 64) and ("k_rank1<" in name) == (mc < 128), (mc, name)
        assert np.array_equal(y, np.load(f"{tmp_path}/one_{i}.npy")), (mc, B, hwd)


def test_unet_forward_vs_oracle_towerruins64(oracle):
    """BASELINE config 1 shape: 64-ch, (H,W,D)=(46,64,46) — non-square planes, ragged tiles."""
    mc, (H, W, D) = 64, (46, 64, 46)
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0, as_torch=False)
    x = T.synthetic_noise((1, 12, H + D, W + D), 5)
    t = np.array([321], np.float32)
    want = oracle.unet_forward(sd, x, t, H, W, D, mc)
    model = make_model(mc)
    with torch.no_grad():
        got = model(cu(x), cu(t), H=H, W=W, D=D).cpu().numpy()
    assert relerr(got, want) < TOL_FWD


def test_unet_properties_full_size():
    """BASELINE config 2 size (128-ch, 128^3): size-independent properties — bit-repeatable, batch elements
    independent (sample b of a batch == the same sample alone), zero DxD corner, finite."""
    mc, (H, W, D) = 128, (128, 128, 128)
    model = make_model(mc)
    x = cu(T.synthetic_noise((2, 12, H + D, W + D), 9))
    t = cu(np.array([999, 3], np.float32))
    with torch.no_grad():
        y2 = model(x, t, H=H, W=W, D=D)
        y2b = model(x, t, H=H, W=W, D=D)
        y1 = model(x[1:2].contiguous(), t[1:2].contiguous(), H=H, W=W, D=D)
    assert torch.equal(y2, y2b), "forward must be bit-repeatable"
    assert torch.equal(y2[1:2], y1), "batch elements must not interact"
    assert torch.isfinite(y2).all()
    assert float(y2[..., H:, W:].abs().max()) == 0.0


# ------------------------------------------------------------------ sampler steps and trajectories vs the reference
def test_sampler_steps_golden():
    g = golden("sampler_steps")
    H, W, D = (int(v) for v in g["hwd"])
    kw = dict(H=H, W=W, D=D)
    model = make_model(32)
    for tag, resp in (("full", ""), ("r20", "20")):
        diff = make_diffusion(resp)
        Tn = diff.num_timesteps
        for ti in (Tn - 1, 1, 0):
            pre = f"{tag}.t{ti}"
            x, eps = cu(g[pre + ".x"]), cu(g[pre + ".eps"])
            t = torch.full((x.shape[0],), ti, device=dev(), dtype=torch.int64)
            diff.noise_fn = lambda z: eps.clone()
            with torch.no_grad():
                o1 = diff.p_sample(model, x, t, model_kwargs=kw)
                o2 = diff.ddim_sample(model, x, t, model_kwargs=kw)
                o3 = diff.ddim_sample(model, x, t, model_kwargs=kw, eta=0.7)
                pm = diff.p_mean_variance(model, x, t, model_kwargs=kw)
            assert relerr(o1["sample"].cpu().numpy(), g[pre + ".p_sample"]) < TOL_FWD
            assert relerr(o1["pred_xstart"].cpu().numpy(), g[pre + ".p_xstart"]) < TOL_FWD
            assert relerr(o2["sample"].cpu().numpy(), g[pre + ".ddim_sample"]) < TOL_FWD
            assert relerr(o2["pred_xstart"].cpu().numpy(), g[pre + ".ddim_xstart"]) < TOL_FWD
            assert relerr(o3["sample"].cpu().numpy(), g[pre + ".ddim_eta_sample"]) < TOL_FWD
            assert relerr(pm["pred_xstart"].cpu().numpy(), g[pre + ".p_xstart"]) < TOL_FWD
            assert pm["mean"].shape == pm["variance"].shape == pm["log_variance"].shape == x.shape


BRANCH_TAGS = ["noclip", "eps", "eps_noclip", "eps_r20", "small", "small_r20_noclip", "eps_small"]


@pytest.mark.parametrize("tag", BRANCH_TAGS)
def test_sampler_branches_golden(tag):
    """Reachable branches without a default-path fixture: clip_denoised=False (train_util.py:181), ModelMeanType.EPSILON
    (gaussian_diffusion.py:306-315, 329-335) and ModelVarType.FIXED_SMALL (:286-289), at t in {T-1, 1, 0}.
    (1) s3d_sampler_step on the reference's own model output: the update arithmetic alone, 2e-6.  (2) p_sample / ddim_sample /
    p_mean_variance through the model at TOL_FWD — times sqrt(1/alphas_cumprod - 1) (157 at t = 999) where an eps-derived x0
    multiplies the UNet's round-off."""
    from sin3dm_amd import _lib
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    g = golden("sampler_branches")
    H, W, D = (int(v) for v in g["hwd"])
    kw = dict(H=H, W=W, D=D)
    px, small, resp, clip = (int(v) for v in g[f"{tag}.cfg"])
    diff = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=bool(px), sigma_small=bool(small),
                                     timestep_respacing=str(resp) if resp else "")
    model = make_model(32)
    Tn = diff.num_timesteps
    for ti in (Tn - 1, 1, 0):
        pre = f"{tag}.t{ti}"
        x, eps, mo = cu(g[pre + ".x"]), cu(g[pre + ".eps"]), cu(g[pre + ".model_out"])
        t = torch.full((x.shape[0],), ti, device=dev(), dtype=torch.int64)
        # (1) the kernel on the reference's model output
        for mode, eta, want in ((_lib.STEP_DDPM, 0.0, "p_sample"), (_lib.STEP_DDIM, 0.0, "ddim_sample"), (_lib.STEP_DDIM, 0.7, "ddim_eta_sample")):
            sample, pred, mean = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            a = _lib.SamplerArgs(mode=mode, mean_type=diff._mean_type_code(), clip_denoised=clip, is_mask_t0=0, eta=eta, T=Tn,
                                 batch=x.shape[0], per_sample=x[0].numel(), model_out=mo.data_ptr(), x=x.data_ptr(),
                                 noise=eps.data_ptr(), t=t.data_ptr(), tables=diff._tables(x.device).data_ptr(), y0=None, mask=None,
                                 sample=sample.data_ptr(), pred_xstart=pred.data_ptr(), mean=mean.data_ptr() if mode == _lib.STEP_DDPM else None)
            _lib.check(_lib.load().s3d_sampler_step(a, _lib.stream_ptr()))
            assert relerr(sample.cpu().numpy(), g[pre + "." + want]) < 2e-6, (ti, want)
            assert relerr(pred.cpu().numpy(), g[pre + ".p_xstart"]) < 2e-6, (ti, want)
            if mode == _lib.STEP_DDPM:
                assert relerr(mean.cpu().numpy(), g[pre + ".mean"]) < 2e-6
        # (2) through the public API
        srm1 = float(diff.sqrt_recipm1_alphas_cumprod[ti])
        amp = 1.0 if px else max(1.0, srm1 * float(np.abs(g[pre + ".model_out"]).max()) / float(np.abs(g[pre + ".p_xstart"]).max()))
        diff.noise_fn = lambda z: eps.clone()
        with torch.no_grad():
            o1 = diff.p_sample(model, x, t, clip_denoised=bool(clip), model_kwargs=kw)
            o2 = diff.ddim_sample(model, x, t, clip_denoised=bool(clip), model_kwargs=kw)
            o3 = diff.ddim_sample(model, x, t, clip_denoised=bool(clip), model_kwargs=kw, eta=0.7)
            pm = diff.p_mean_variance(model, x, t, clip_denoised=bool(clip), model_kwargs=kw)
        tol = TOL_FWD * amp
        assert relerr(o1["sample"].cpu().numpy(), g[pre + ".p_sample"]) < tol and relerr(o1["pred_xstart"].cpu().numpy(), g[pre + ".p_xstart"]) < tol
        assert relerr(o2["sample"].cpu().numpy(), g[pre + ".ddim_sample"]) < tol and relerr(o3["sample"].cpu().numpy(), g[pre + ".ddim_eta_sample"]) < tol
        assert relerr(pm["mean"].cpu().numpy(), g[pre + ".mean"]) < tol and relerr(pm["pred_xstart"].cpu().numpy(), g[pre + ".p_xstart"]) < tol
        assert abs(float(pm["variance"].flatten()[0]) - float(g[pre + ".variance"][0])) <= 1e-6 * abs(float(g[pre + ".variance"][0]))
        assert abs(float(pm["log_variance"].flatten()[0]) - float(g[pre + ".log_variance"][0])) <= 1e-6 * abs(float(g[pre + ".log_variance"][0]))


@pytest.mark.parametrize("tag,px,resp,ddim,clip", [("noclip_ddpm20", True, "20", False, False), ("eps_ddim10", False, "10", True, True)])
def test_trajectories_branches_golden(tag, px, resp, ddim, clip):
    """An unclipped ancestral run as TrainLoop._sample_and_visualize does it (train_util.py:181) and an epsilon-prediction DDIM
    run, whole loops against the reference with its eps stream."""
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    g = golden("trajectories_branches")
    H, W, D = (int(v) for v in g["hwd"])
    model = make_model(32)
    diff = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=px, timestep_respacing=resp)
    eps = iter(cu(g[f"{tag}.eps"]))
    diff.noise_fn = lambda z: next(eps).clone()
    xT = cu(g[f"{tag}.xT"])
    fn = diff.ddim_sample_loop if ddim else diff.p_sample_loop
    final = fn(model, tuple(xT.shape), noise=xT.clone(), clip_denoised=clip, model_kwargs=dict(H=H, W=W, D=D)).cpu().numpy()
    assert relerr(final, g[f"{tag}.final"]) < 2e-4
    # the D x D corner: the model's output is zero there.  As an x0 prediction that ends the corner at exactly 0; as an eps
    # prediction x0 = x_t / sqrt(alphas_cumprod) there (clamped): the reference's corner is NOT zero, and ours equals it
    corner, ref_corner = final[..., H:, W:], g[f"{tag}.final"][..., H:, W:]
    if px:
        assert np.all(corner == 0)
    else:
        assert np.abs(ref_corner).max() > 0.5 and relerr(corner, ref_corner) < 2e-4


@pytest.mark.parametrize("mc,B", [(32, 2), (64, 1), (64, 2)])
def test_fused_denoise_step_equals_forward_plus_sampler_kernel(mc, B):
    """The sampling loops' step (s3d_unet_step_film: ONE library call; SURVEY section 2b K8 + K9) against the single-step API
    (forward, then s3d_sampler_step), bit for bit: DDPM, DDIM with eta, the in-painting branch, EPSILON and unclipped, t = 0
    included.  mc = 64, batch 1: the pixel-chunk output head applies the update in its own launch and the model output is never
    stored; batch 2: the same head writes to the workspace and the sampler kernel follows (bandwidth-bound regime); mc = 32: the
    generic head.  And the loops draw their eps ahead from the device generator."""
    from sin3dm_amd import _lib
    from sin3dm_amd.diffusion.gaussian_diffusion import HostTimesteps
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    H, W, D = 12, 9, 7
    kw = dict(H=H, W=W, D=D)
    model = make_model(mc)
    shape = (B, 12, H + D, W + D)
    x, eps = cu(T.synthetic_noise(shape, 91)), cu(T.synthetic_noise(shape, 92))
    y0 = cu(T.synthetic_noise(shape, 93))
    mask = (cu(T.synthetic_noise(shape, 94)) > 0).float()
    for px, resp in ((True, "10"), (False, "")):
        diff = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=px, timestep_respacing=resp)
        for ti in (diff.num_timesteps - 1, 0):
            t = torch.full((B,), ti, device=dev(), dtype=torch.int64)
            ht = HostTimesteps(t, (ti,) * B)
            for mode, extra in ((_lib.STEP_DDPM, {}), (_lib.STEP_DDIM, dict(eta=0.7)), (_lib.STEP_DDIM, dict(y0=y0, mask=mask)),
                                (_lib.STEP_DDIM, dict(y0=y0, mask=mask, is_mask_t0=True))):
                for clip in (True, False):
                    with torch.no_grad():
                        a = diff._step(mode, model, x, ht, clip, None, kw, fuse=False, noise=eps, **extra)
                        b = diff._step(mode, model, x, ht, clip, None, kw, fuse=True, noise=eps, **extra)
                    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (mc, B, px, ti, mode, extra.keys(), clip)
    # device-generator loops: reproducible under a seed, finite, corner at zero
    diff = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=True, timestep_respacing="10")
    outs = []
    for _ in range(2):
        torch.manual_seed(5)
        outs.append(diff.p_sample_loop(model, shape, model_kwargs=kw))
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all() and float(outs[0][..., H:, W:].abs().max()) == 0.0
    diff._NOISE_AHEAD_BYTES = 3 * 4 * outs[0].numel()            # three steps per randn launch: the chunk boundary inside a loop
    torch.manual_seed(5)
    z = diff.p_sample_loop(model, shape, model_kwargs=kw)
    assert torch.isfinite(z).all() and float(z[..., H:, W:].abs().max()) == 0.0


@pytest.mark.parametrize("mc,B,hwd,ddim", [(64, 1, (40, 24, 56), False), (64, 2, (12, 9, 7), True), (128, 1, (64, 48, 32), False), (32, 2, (10, 14, 6), False)])
def test_next_steps_in_conv_carried_by_the_output_head_is_the_same_bits(mc, B, hwd, ddim):
    """in_conv is TriplaneConv(in, ch, 1, padding=0, is_rollout=False) (src/diffusion/unet_triplane.py:378, 482): a pointwise map of x_t
    with no timestep in it, and in a sampling loop x_t IS the previous step's sample (gaussian_diffusion.py:533-534).  The loops let the
    output head of step n also evaluate step n + 1's in_conv and its GroupNorm partial sums on the x_{t-1} it has just formed
    (s3d_unet_step_film_carry: SURVEY.md section 2b, K8 + K9 + K2) — k_in_conv_lds's products in its order and thread
    mapping.  Every step's sample and pred_xstart equal the loop that launches in_conv each step, bit for bit: ragged planes whose
    rows are not multiples of the head's 64-pixel or in_conv's 32-pixel segments, batch 2, DDPM and DDIM, the widths of the
    pixel-chunk head (64 / 128; at 32 channels the generic head takes the step and nothing is carried).  A consumer of the
    progressive loop that edits out["sample"] in place switches the carry off for that step (version counter): still equal."""
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    H, W, D = hwd
    kw = dict(H=H, W=W, D=D)
    shape = (B, 12, H + D, W + D)
    diff = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=True, timestep_respacing="ddim6" if ddim else "6")
    loop = diff.ddim_sample_loop_progressive if ddim else diff.p_sample_loop_progressive

    def run(carry, edit_at=None):
        model = make_model(mc)
        model.carries_in_conv = carry
        torch.manual_seed(17)
        outs = []
        for n, out in enumerate(loop(model, shape, model_kwargs=kw)):
            outs.append((out["sample"].clone(), out["pred_xstart"].clone()))
            if n == edit_at:
                out["sample"].mul_(0.5)                    # the reference's loop continues from whatever the consumer left in out["sample"]
        return outs

    a, b = run(False), run(True)
    assert len(a) == 6
    for n, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), (mc, B, hwd, ddim, n)
    assert torch.isfinite(a[-1][0]).all()
    a, b = run(False, edit_at=2), run(True, edit_at=2)
    for n, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), ("edited", mc, n)


@pytest.mark.parametrize("tag,resp,ddim", [("ddim10", "10", True), ("ddpm20", "20", False)])
def test_trajectories_golden(tag, resp, ddim):
    g = golden("trajectories")
    H, W, D = (int(v) for v in g["hwd"])
    model = make_model(32)
    diff = make_diffusion(resp)
    eps = iter(cu(g[f"{tag}.eps"]))
    diff.noise_fn = lambda z: next(eps).clone()
    xT = cu(g[f"{tag}.xT"])
    fn = diff.ddim_sample_loop_progressive if ddim else diff.p_sample_loop_progressive
    inter = []
    Tn = diff.num_timesteps
    for k, o in enumerate(fn(model, tuple(xT.shape), noise=xT.clone(), model_kwargs=dict(H=H, W=W, D=D))):
        if k % 5 == 4 or k == Tn - 1:
            inter.append(o["sample"].cpu().numpy())
        final = o["sample"]
    final = final.cpu().numpy()
    assert relerr(final, g[f"{tag}.final"]) < 2e-4
    assert relerr(np.stack(inter), g[f"{tag}.inter"]) < 2e-4
    assert np.all(final[..., H:, W:] == 0), "DxD corner must end at exactly 0"


def test_sample_loop_api_device_noise():
    """p_sample_loop / ddim_sample_loop with the default (device) noise source: shape, finiteness, corner."""
    H, W, D = 12, 8, 10
    model = make_model(32)
    diff = make_diffusion("10")
    shape = (2, 12, H + D, W + D)
    torch.manual_seed(0)
    a = diff.p_sample_loop(model, shape, model_kwargs=dict(H=H, W=W, D=D))
    b = diff.ddim_sample_loop(model, shape, model_kwargs=dict(H=H, W=W, D=D))
    for s in (a, b):
        assert tuple(s.shape) == shape and torch.isfinite(s).all()
        assert float(s[..., H:, W:].abs().max()) == 0.0


# ------------------------------------------------------------------ decoder vs the reference
def make_decoder(up, hid):
    from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip
    net = AutoEncoderGroupSkip(4, 8, up, hid, 4)
    missing, unexpected = net.load_state_dict(T.synthetic_state_dict(T.ae_param_shapes(4, 8, up, hid, 4), 5), strict=False)
    assert not unexpected and all(k.startswith(("geo_encoder", "tex_encoder", "aabb")) for k in missing)
    return net.to(dev()).eval()


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_decoder_golden(tag):
    g = golden("decoder")
    up, hid, H, W, D = (int(v) for v in g[f"{tag}.cfg"])
    net = make_decoder(up, hid)
    fm = [cu(g[f"{tag}.{p}"]) for p in T.PLANES]
    out = net.decode(cu(g[f"{tag}.pts"]), fm, aabb=torch.from_numpy(g[f"{tag}.aabb"]))
    assert relerr(out.cpu().numpy(), g[f"{tag}.out"]) < TOL_FWD
    out = net.decode(cu(g[f"{tag}.pts"][:33]), fm)                      # module buffer aabb
    assert relerr(out.cpu().numpy(), g[f"{tag}.out_default_aabb"]) < TOL_FWD


def test_decoder_grid_vs_oracle(oracle):
    """decode_grid over a non-cubic aabb: grid coordinates restated from src/encoding/utils3d.py:13-25,
    values checked against the oracle's decode on those points; full-width decoder (up 64, hidden 256)."""
    up, hid, (H, W, D), reso = 64, 256, (20, 12, 16), 24
    net = make_decoder(up, hid)
    fm = [0.8 * np.tanh(T.synthetic_noise(s, 40 + i)) for i, s in enumerate(((1, 12, H, W), (1, 12, H, D), (1, 12, W, D)))]
    aabb = torch.tensor([-1.0, -0.6, -0.8, 1.0, 0.6, 0.8])
    grid = net.decode_grid([cu(f) for f in fm], reso, aabb=aabb).cpu().numpy()
    size = aabb[3:] - aabb[:3]
    res = (reso * size / size.max()).long()
    assert grid.shape == (int(res[0]), int(res[1]), int(res[2]), 4) == (24, 14, 19, 4)
    axes = [torch.linspace(0.5, float(r) - 0.5, int(r)) / r * size[i] + aabb[i] for i, r in enumerate(res)]
    pts = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3).numpy()
    sd = oracle.Params(T.synthetic_state_dict(T.ae_param_shapes(4, 8, up, hid, 4), 5, as_torch=False))
    want = oracle.ae_decode(sd, pts, *fm, aabb.numpy(), 4, 8, up, hid, 4)
    want[:, 1:] = np.clip(want[:, 1:], 0, 1)
    assert relerr(grid.reshape(-1, 4), want) < TOL_FWD
    pts_out = net.decode(cu(pts), [cu(f) for f in fm], aabb=aabb, clamp_color=True).cpu().numpy()
    # the in-kernel grid coordinates may differ from torch's CPU linspace/div by one ulp
    assert relerr(pts_out, grid.reshape(-1, 4)) < 1e-5, "grid mode and point mode must agree"


@pytest.mark.gpu
def test_full_size_trajectory_vs_cpu_port(oracle):
    """BASELINE config 2 itself (128-ch UNet, 128^3 triplane, DDPM-1000 schedule): eight consecutive ancestral steps
    from t = 999 with identical noise on the HIP path and on the CPU port (oracle/torch_port.py, the reference's
    algorithm on ATen/oneDNN).  ~2.5 s of CPU time per step bounds the length; tools/validate_full_size.py runs longer
    chains (profiles/r01_full_size_parity.txt)."""
    import torch
    import torch_port as tp
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
    mc, (H, W, D), steps = 128, (128, 128, 128), 8
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0)
    model = TriplaneUNetModelSmall(12, mc, 12, use_scale_shift_norm=True)
    model.load_state_dict(sd)
    model.cuda().eval()
    diffusion = create_gaussian_diffusion(steps=1000, predict_xstart=True)
    tab, _ = oracle.schedule_tables(None, 1000)
    g = torch.Generator().manual_seed(3)
    x_cpu = torch.randn((1, 12, H + D, W + D), generator=g)
    x_gpu = x_cpu.cuda()
    worst = 0.0
    for k in range(steps):
        t = 999 - k
        eps = torch.randn(x_cpu.shape, generator=g)
        diffusion.noise_fn = lambda z, e=eps: e.to(z.device)
        with torch.no_grad():
            x_gpu = diffusion.p_sample(model, x_gpu, torch.tensor([t], device="cuda"), model_kwargs=dict(H=H, W=W, D=D))["sample"]
            out = tp.unet_forward(sd, x_cpu, torch.tensor([float(t)]), H, W, D, mc)
            x_cpu, _ = tp.p_sample_update(out, x_cpu, eps, tab, t)
        worst = max(worst, relerr(x_gpu.cpu().numpy(), x_cpu.numpy()))
    assert worst < 1e-4, worst


# ------------------------------------------------------------------ a5 / a7 leaves directly from the committed goldens
@pytest.mark.parametrize("tag,C,Cout,ssn", [("same", 32, 32, True), ("skip", 32, 64, True), ("add", 32, 32, False)])
def test_leaf_resblock_golden(tag, C, Cout, ssn):
    """TriplaneResBlock._forward (unet_triplane.py:269-311) alone: FiLM on/off, identity / 1x1 skip."""
    from sin3dm_amd import ops
    g = golden("resblock")
    full = T.unet_param_shapes(model_channels=32, use_scale_shift_norm=ssn)
    pre = "input_blocks.0.0." if C == Cout else "input_blocks.1.1."
    shapes = {k[len(pre):]: v for k, v in full.items() if k.startswith(pre)}
    params = {k: torch.from_numpy(T.synthetic_tensor(k, v, 3)) for k, v in shapes.items()}
    fm = [cu(g[f"{tag}.in_{p}"]) for p in T.PLANES]
    out = ops.triplane_resblock(fm, cu(g[f"{tag}.emb"]), params, Cout, use_scale_shift_norm=ssn)
    for p, y in zip(T.PLANES, out):
        assert relerr(y.cpu().numpy(), g[f"{tag}.out_{p}"]) < TOL_OP, (tag, p)


def test_leaf_timestep_embedding_golden():
    """timestep_embedding (nn.py:103-121): cos | sin of t * exp(-ln(1e4) i / half).  Arguments reach 999 rad, where one
    fp32 ulp of the argument is 6e-5: absolute tolerance, as in the CPU oracle's test of the same fixture."""
    from sin3dm_amd import ops
    g = golden("temb")
    for mc in (32, 64):
        e = ops.timestep_embedding(cu(g["t"].astype(np.float32)), mc).cpu().numpy()
        assert e.shape == g[f"emb{mc}"].shape
        np.testing.assert_allclose(e, g[f"emb{mc}"], atol=2e-4, rtol=0)
        small = g["t"] <= 1                                           # t in {0, 1}: no argument-rounding excuse
        np.testing.assert_allclose(e[small], g[f"emb{mc}"][small], atol=2e-6, rtol=0)


def test_ddim_inpainting_branch(oracle):
    """ddim_sample's y0 / mask branch (gaussian_diffusion.py:568-577): pred_xstart is replaced by
    mask * y0 + (1 - mask) * pred_xstart — at every step when is_mask_t0, otherwise only while t != 0."""
    H, W, D = 10, 14, 6
    kw = dict(H=H, W=W, D=D)
    model = make_model(32)
    diff = make_diffusion("10")
    shape = (2, 12, H + D, W + D)
    x, eps = T.synthetic_noise(shape, 21), T.synthetic_noise(shape, 22)
    y0 = np.tanh(T.synthetic_noise(shape, 23))
    mask = (T.synthetic_noise(shape, 24) > 0).astype(np.float32)
    tab, tmap = oracle.schedule_tables(sorted(diff.use_timesteps))
    sd = oracle.Params(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 0, as_torch=False))
    diff.noise_fn = lambda z: cu(eps)
    for ti in (7, 0):
        mo = oracle.unet_forward(sd, x, [tmap[ti]] * 2, H, W, D, 32)
        plain = np.clip(mo, -1, 1)
        mixed = mask * y0 + (1 - mask) * plain
        t = torch.full((2,), ti, device=dev(), dtype=torch.int64)
        for is_t0 in (False, True):
            for eta in (0.0, 0.6):
                with torch.no_grad():
                    o = diff.ddim_sample(model, cu(x), t, model_kwargs=kw, eta=eta, y0=cu(y0), mask=cu(mask), is_mask_t0=is_t0)
                x0 = mixed if (is_t0 or ti != 0) else plain
                # the update of ddim_update() with its x0 replaced: eps is re-derived from the mixed x0 (:579)
                sr, srm1, ab, abp = (float(tab[r][ti]) for r in (3, 4, 1, 2))     # oracle.TABLE_ROWS
                e = (np.float32(sr) * x - x0) / np.float32(srm1)
                sigma = eta * np.sqrt((1 - abp) / (1 - ab)) * np.sqrt(1 - ab / abp)
                want = x0 * np.float32(np.sqrt(abp)) + np.float32(np.sqrt(1 - abp - sigma ** 2)) * e + (ti != 0) * np.float32(sigma) * eps
                assert relerr(o["pred_xstart"].cpu().numpy(), x0) < TOL_FWD, (ti, is_t0, eta)
                assert relerr(o["sample"].cpu().numpy(), want) < TOL_FWD, (ti, is_t0, eta)
    # and without y0/mask the branch is inert
    with torch.no_grad():
        a = diff.ddim_sample(model, cu(x), t, model_kwargs=kw)
        b = diff.ddim_sample(model, cu(x), t, model_kwargs=kw, y0=None, mask=None, is_mask_t0=True)
    assert torch.equal(a["sample"], b["sample"])


# ------------------------------------------------------------------ BASELINE configs 3 and 5 at their full sizes
def tp_ddim(model_out, x, tab, t):
    """eta = 0 DDIM update on torch tensors (gaussian_diffusion.py:538-600), fp32 like the reference."""
    x0 = model_out.clamp(-1, 1)
    f = lambda r: torch.tensor(float(tab[r][t]), dtype=torch.float32)            # rows as oracle.TABLE_ROWS
    eps = (f(3) * x - x0) / f(4)
    abp = f(2)
    return x0 * torch.sqrt(abp) + torch.sqrt(1 - abp) * eps, x0


def test_config3_batch8_ddim100(oracle):
    """BASELINE configs[2]: 128-ch UNet, (128,128,128), DDIM-100, 8 samples per GPU (src/sample.py:33-38 batches
    bs <= diff_batch_size).  (i) every element of the batch-8 forward is bit-identical to the same sample run alone;
    (ii) three DDIM-100 steps of the whole batch, element 0 checked against the CPU port of the reference on the same
    noise-free chain (eta = 0), elements 0 and 5 against their batch-1 runs bit for bit."""
    sys_path_oracle()
    mc, hwd, B = 128, (128, 128, 128), 8
    H, W, D = hwd
    kw = dict(H=H, W=W, D=D)
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0)
    model = make_model(mc)
    diff = make_diffusion("100")
    assert diff.num_timesteps == 100
    tab, tmap = oracle.schedule_tables(sorted(diff.use_timesteps))
    x = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 31))
    with torch.no_grad():
        xb = x.to(dev())
        x0, x5 = x[0:1].to(dev()), x[5:6].to(dev())
        x_cpu = x[0:1].clone()
        worst = 0.0
        for k in range(3):
            ti = 99 - k
            t8 = torch.full((B,), ti, device=dev(), dtype=torch.int64)
            xb = diff.ddim_sample(model, xb, t8, model_kwargs=kw)["sample"]
            x0 = diff.ddim_sample(model, x0, t8[:1], model_kwargs=kw)["sample"]
            x5 = diff.ddim_sample(model, x5, t8[:1], model_kwargs=kw)["sample"]
            assert torch.equal(xb[0:1], x0) and torch.equal(xb[5:6], x5), f"step {k}: batch elements must not interact"
            import torch_port as tp
            out = tp.unet_forward(sd, x_cpu, torch.tensor([float(tmap[ti])]), H, W, D, mc)
            x_cpu, _ = tp_ddim(out, x_cpu, tab, ti)
            worst = max(worst, relerr(x0.cpu().numpy(), x_cpu.numpy()))
    assert worst < TOL_FWD, worst
    assert torch.isfinite(xb).all()


def sys_path_oracle():
    import os, sys
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)


def test_config5_retargeted_256x256x128(oracle):
    """BASELINE configs[4], diffusion half: --resize 2 2 1 from a 128^3 encoding -> (H,W,D) = (256,256,128), composed
    map [12, 384, 384] (src/sample.py:26-38), 128-ch UNet, DDPM-1000: three ancestral steps with identical noise on the
    HIP path and on the CPU port of the reference; repeatable bits, finite, x0 prediction clamped."""
    sys_path_oracle()
    import torch_port as tp
    mc, hwd = 128, (256, 256, 128)
    H, W, D = hwd
    kw = dict(H=H, W=W, D=D)
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0)
    model = make_model(mc)
    diff = make_diffusion("")
    tab, _ = oracle.schedule_tables(None, 1000)
    g = torch.Generator().manual_seed(5)
    x_cpu = torch.randn((1, 12, H + D, W + D), generator=g)
    x_gpu = x_cpu.to(dev())
    worst = 0.0
    for k in range(3):
        t = 999 - k
        eps = torch.randn(x_cpu.shape, generator=g)
        diff.noise_fn = lambda z, e=eps: e.to(z.device)
        with torch.no_grad():
            tt = torch.tensor([t], device=dev())
            o = diff.p_sample(model, x_gpu, tt, model_kwargs=kw)
            if k == 0:
                o2 = diff.p_sample(model, x_gpu, tt, model_kwargs=kw)
                assert torch.equal(o["sample"], o2["sample"]), "bit-repeatable"
            assert float(o["pred_xstart"].abs().max()) <= 1.0
            x_gpu = o["sample"]
            out = tp.unet_forward(sd, x_cpu, torch.tensor([float(t)]), H, W, D, mc)
            x_cpu, _ = tp.p_sample_update(out, x_cpu, eps, tab, t)
        worst = max(worst, relerr(x_gpu.cpu().numpy(), x_cpu.numpy()))
    assert worst < TOL_FWD, worst
    assert torch.isfinite(x_gpu).all()


def test_config5_decode_grid_512x512x256(oracle):
    """BASELINE configs[4], decode half: decode_grid at --reso 512 of a (256,256,128) triplane whose aabb is the 128^3
    encoding's scaled by (2,2,1) (_resize_aabb, src/encoding/model.py:351-360) -> 512 x 512 x 256 cells
    (decode_grid :335-349, grid points utils3d.py:13-25).  Shape, finiteness, colour range, and 4096 randomly indexed
    cells against the C oracle's decode of the same cell centres."""
    from types import SimpleNamespace
    from sin3dm_amd.encoding.model import ShapeAutoEncoder
    up, hid, (H, W, D), reso = 64, 256, (256, 256, 128), 512
    cfg = SimpleNamespace(enc_net_type="skip", fdim_geo=4, fdim_tex=8, fdim_up=up, hidden_dim=hid, n_hidden_layers=4, data_type="sdftex")
    ae = ShapeAutoEncoder("/nonexistent", cfg, device=dev())
    sdt = T.synthetic_state_dict(T.ae_param_shapes(4, 8, up, hid, 4), 5)
    missing, unexpected = ae.net.load_state_dict(sdt, strict=False)
    assert not unexpected
    ae.net.eval()
    ae.aabb = torch.tensor([-0.5, -0.5, -0.5, 0.5, 0.5, 0.5], device=dev())
    ae.featmap_size = (128, 128, 128)
    aabb = ae._resize_aabb((H, W, D))
    assert np.allclose(aabb.cpu().numpy(), [-1.0, -1.0, -0.5, 1.0, 1.0, 0.5])
    fm = [0.8 * np.tanh(T.synthetic_noise(s, 60 + i)) for i, s in enumerate(((1, 12, H, W), (1, 12, H, D), (1, 12, W, D)))]
    grid = ae.decode_grid([cu(f) for f in fm], reso, aabb=aabb)
    assert tuple(grid.shape) == (512, 512, 256, 4)
    assert torch.isfinite(grid).all()
    assert float(grid[..., 1:].min()) >= 0.0 and float(grid[..., 1:].max()) <= 1.0
    rng = np.random.Generator(np.random.PCG64(77))
    idx = np.stack([rng.integers(0, n, 4096) for n in (512, 512, 256)], axis=1)
    idx[:8] = [[0, 0, 0], [511, 511, 255], [0, 511, 0], [511, 0, 255], [0, 0, 255], [511, 511, 0], [255, 256, 127], [256, 255, 128]]
    a = aabb.cpu().numpy().astype(np.float32)
    size = a[3:] - a[:3]
    res = np.array([512, 512, 256], np.float32)
    pts = ((idx.astype(np.float32) + np.float32(0.5)) / res * size + a[:3]).astype(np.float32)
    sd = oracle.Params(T.synthetic_state_dict(T.ae_param_shapes(4, 8, up, hid, 4), 5, as_torch=False))
    want = oracle.ae_decode(sd, pts, *fm, a, 4, 8, up, hid, 4)
    want[:, 1:] = np.clip(want[:, 1:], 0, 1)
    got = grid[idx[:, 0], idx[:, 1], idx[:, 2]].cpu().numpy()
    assert relerr(got, want) < TOL_FWD


def test_host_known_timesteps_change_nothing():
    """The sampling loops hand the model HostTimesteps (values known on the host): the timestep_map gather and the
    timestep MLP are then served from per-value caches.  Bit-identical to plain tensors, per sample and per batch with
    mixed values, across respacing; and the cache follows the weights."""
    from sin3dm_amd.diffusion.gaussian_diffusion import HostTimesteps
    H, W, D = 10, 14, 6
    kw = dict(H=H, W=W, D=D)
    model = make_model(32)
    x = cu(T.synthetic_noise((2, 12, H + D, W + D), 71))
    eps = cu(T.synthetic_noise((2, 12, H + D, W + D), 72))
    for resp in ("", "10"):
        diff = make_diffusion(resp)
        diff.noise_fn = lambda z: eps
        for vals in ((3, 3), (diff.num_timesteps - 1, 1)):
            t = torch.tensor(vals, device=dev(), dtype=torch.int64)
            with torch.no_grad():
                a = diff.p_sample(model, x, t, model_kwargs=kw)
                b = diff.p_sample(model, x, HostTimesteps(t, vals), model_kwargs=kw)
                c = diff.p_sample(model, x, HostTimesteps(t, vals), model_kwargs=kw)       # served from the caches
            assert torch.equal(a["sample"], b["sample"]) and torch.equal(a["sample"], c["sample"]), (resp, vals)
    # the loop API (which uses them) against manual stepping with plain tensors
    diff = make_diffusion("5")
    noise = iter([cu(T.synthetic_noise((2, 12, H + D, W + D), 80 + k)) for k in range(10)])
    diff.noise_fn = lambda z: next(noise)
    xT = cu(T.synthetic_noise((2, 12, H + D, W + D), 79))
    with torch.no_grad():
        got = diff.p_sample_loop(model, tuple(xT.shape), noise=xT.clone(), model_kwargs=kw)
        noise = iter([cu(T.synthetic_noise((2, 12, H + D, W + D), 80 + k)) for k in range(10)])
        y = xT.clone()
        for i in range(4, -1, -1):
            y = diff.p_sample(model, y, torch.full((2,), i, device=dev(), dtype=torch.int64), model_kwargs=kw)["sample"]
    assert torch.equal(got, y)
    # weights change -> the cached FiLM tables must not survive
    t = torch.tensor((3, 3), device=dev(), dtype=torch.int64)
    diff = make_diffusion("")
    diff.noise_fn = lambda z: eps
    with torch.no_grad():
        before = diff.p_sample(model, x, HostTimesteps(t, (3, 3)), model_kwargs=kw)["sample"]
        dict(model.named_parameters())["time_embed.2.bias"].add_(0.25)
        after_h = diff.p_sample(model, x, HostTimesteps(t, (3, 3)), model_kwargs=kw)["sample"]
        after_p = diff.p_sample(model, x, t, model_kwargs=kw)["sample"]
    assert torch.equal(after_h, after_p) and not torch.equal(before, after_h)


def test_config1_towerruins64_ddim10_whole_run(oracle):
    """BASELINE configs[0] end to end on the diffusion side: the default 64-ch UNet on the towerruins triplane at
    --fm_reso 64, (H,W,D) = (46,64,46), a whole DDIM-10 run (respacing "10": timesteps 0, 111, ..., 999) through
    ddim_sample_loop against the CPU port of the reference stepping the same chain; eta = 0, so the only random input
    is x_T.  The corner must end at exactly 0."""
    sys_path_oracle()
    import torch_port as tp
    mc, (H, W, D) = 64, (46, 64, 46)
    kw = dict(H=H, W=W, D=D)
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc), 0)
    model = make_model(mc)
    diff = make_diffusion("10")
    tab, tmap = oracle.schedule_tables(sorted(diff.use_timesteps))
    assert list(tmap) == [0, 111, 222, 333, 444, 555, 666, 777, 888, 999]
    xT = torch.from_numpy(T.synthetic_noise((1, 12, H + D, W + D), 91))
    with torch.no_grad():
        got = diff.ddim_sample_loop(model, tuple(xT.shape), noise=xT.to(dev()), model_kwargs=kw)
        x = xT.clone()
        for i in range(9, -1, -1):
            out = tp.unet_forward(sd, x, torch.tensor([float(tmap[i])]), H, W, D, mc)
            x, _ = tp_ddim(out, x, tab, i)
    assert relerr(got.cpu().numpy(), x.numpy()) < TOL_FWD
    assert float(got[..., H:, W:].abs().max()) == 0.0


def test_rank1_table_slices_agree_with_the_unsliced_tables(tmp_path):
    """From 256 own channels on, the rank-1 rollout tables (unet_triplane.py:37-58) are built as two K slices by twice as many
    k_rank1 blocks with half the stage chain each, and k_conv_wino24s adds the slices in its epilogue (s3d_rank1.h).  Only the
    order of one fp32 addition changes: the 128-channel cases (256- and 384-channel rollout inputs) against S3D_RANK1_SLICES=0 in
    a separate process, to 2e-6 of the output's scale — and the sliced default keeps its golden / oracle gates elsewhere."""
    import os, subprocess, sys
    cases = [c for c in R1_CASES if c[0] == 128]
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, 'tests')\n"
        "import test_hip_parity as tp\n"
        "for i, (mc, B, hwd) in enumerate([c for c in tp.R1_CASES if c[0] == 128]):\n"
        "    y, name = tp._r1_forward(mc, B, hwd, 90 + i)\n"
        f"    np.save(r'{tmp_path}/sl_' + str(i) + '.npy', y)\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, S3D_RANK1_SLICES="0"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    for i, (mc, B, hwd) in enumerate(cases):
        y, _ = _r1_forward(mc, B, hwd, 90 + i)
        want = np.load(f"{tmp_path}/sl_{i}.npy")
        assert relerr(y, want) < 2e-6, (mc, B, hwd, relerr(y, want))
