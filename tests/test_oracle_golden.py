"""Pins the C oracle (oracle/sin3dm_oracle.c) against golden vectors captured from the reference.

CPU-only.  Tolerance: 2e-5 relative (max|a-b|/max|b|) per op — fp32 with a different summation
order than oneDNN; the north_star gate for the HIP path is 1e-3.
"""
import numpy as np
import pytest

from conftest import golden, relerr, digest_errors, zero_grad_params
from sin3dm_amd import testing as T

TOL = 2e-5


def test_schedule_tables(oracle):
    g = golden("schedules")
    from sin3dm_amd.diffusion.respace import space_timesteps
    for tag, resp in (("full", None), ("r100", "100"), ("r10", "10"), ("ddim50", "ddim50"), ("r20", "20")):
        keep = None if resp is None else space_timesteps(1000, resp)
        tab, tmap = oracle.schedule_tables(keep)
        assert np.array_equal(tmap, g[f"{tag}.timestep_map"])
        for i, row in enumerate(oracle.TABLE_ROWS):
            np.testing.assert_allclose(tab[i], g[f"{tag}.{row}"], rtol=1e-12, atol=1e-15, err_msg=f"{tag}.{row}")


def test_timestep_embedding(oracle):
    g = golden("temb")
    for mc in (32, 64):
        e = oracle.timestep_embedding(g["t"], mc)
        # cos/sin of arguments up to 999 rad: fp32 argument rounding dominates -> absolute tolerance
        np.testing.assert_allclose(e, g[f"emb{mc}"], atol=2e-4, rtol=0)


@pytest.mark.parametrize("tag,C", [("a", 32), ("b", 64)])
def test_leaves(oracle, tag, C):
    g = golden("leaves")
    fm = [g[f"{tag}.in_{p}"] for p in T.PLANES]
    shapes = {f"0.norm_{p}.{l}": (C,) for p in T.PLANES for l in ("weight", "bias")}
    out = oracle.triplane_norm_silu(T.synthetic_state_dict(shapes, 1, as_torch=False), "0", fm)
    for p, y in zip(T.PLANES, out):
        assert relerr(y, g[f"{tag}.normsilu_{p}"]) < TOL
    for name, k, roll, cout in (("conv3r", 3, True, 48), ("conv3", 3, False, 48), ("conv1", 1, False, 40)):
        shapes = {}
        for p in T.PLANES:
            shapes[f"conv_{p}.weight"] = (cout, C * 3 if roll else C, k, k)
            shapes[f"conv_{p}.bias"] = (cout,)
        sd = {"c." + k_: T.synthetic_tensor(k_, v, 2) for k_, v in shapes.items()}
        out = oracle.triplane_conv(sd, "c", fm, cout, k, roll)
        for p, y in zip(T.PLANES, out):
            assert relerr(y, g[f"{tag}.{name}_{p}"]) < TOL, (name, p)
    for p, x in zip(T.PLANES, fm):
        assert relerr(oracle.avgpool2(x), g[f"{tag}.down_{p}"]) < 1e-6
        h, w = x.shape[-2:]
        assert relerr(oracle.bilinear(x, 2 * h, 2 * w), g[f"{tag}.up_{p}"]) < 1e-6
        assert relerr(oracle.bilinear(x, 2 * h + 1, 2 * w + 1), g[f"{tag}.resize_{p}"]) < 2e-6


@pytest.mark.parametrize("tag,C,Cout,ssn", [("same", 32, 32, True), ("skip", 32, 64, True), ("add", 32, 32, False)])
def test_resblock(oracle, tag, C, Cout, ssn):
    g = golden("resblock")
    full = T.unet_param_shapes(model_channels=32, use_scale_shift_norm=ssn)
    # the generator named the block's tensors by its own state_dict keys
    pre = "input_blocks.0.0." if C == Cout else "input_blocks.1.1."
    shapes = {k[len(pre):]: v for k, v in full.items() if k.startswith(pre)}
    sd = {"b." + k: T.synthetic_tensor(k, v, 3) for k, v in shapes.items()}
    fm = [g[f"{tag}.in_{p}"] for p in T.PLANES]
    out = oracle.triplane_resblock(sd, "b", fm, g[f"{tag}.emb"], Cout, ssn)
    for p, y in zip(T.PLANES, out):
        assert relerr(y, g[f"{tag}.out_{p}"]) < TOL


UNET_CASES = [("mc32_a", 32, False, True, (1, 2)), ("mc32_odd", 32, False, True, (1, 2)),
              ("mc64_b", 64, False, True, (1, 2)), ("mc32_raw", 32, True, True, (1, 2)),
              ("mc32_add", 32, False, False, (1, 2)), ("mc32_3lev", 32, False, True, (1, 2, 2))]


@pytest.mark.parametrize("tag,mc,raw,ssn,cm", UNET_CASES)
def test_unet_forward(oracle, tag, mc, raw, ssn, cm):
    g = golden("unet_fwd")
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc, rollout=not raw, use_scale_shift_norm=ssn,
                                                    channel_mult=cm), 0, as_torch=False)
    H, W, D = (int(v) for v in g[f"{tag}.hwd"])
    y = oracle.unet_forward(sd, g[f"{tag}.x"], g[f"{tag}.t"], H, W, D, mc, cm, ssn, not raw)
    assert relerr(y, g[f"{tag}.y"]) < 5e-5
    assert np.all(y[..., H:, W:] == 0)


def test_manifest_matches_reference():
    g = golden("unet_manifest")
    for mc in (32, 64, 128):
        mine = T.unet_param_shapes(model_channels=mc)
        ref = {k.split("/", 1)[1]: tuple(int(i) for i in g[k]) for k in g.files if k.startswith(f"mc{mc}/")}
        assert ref == dict(mine)


def test_sampler_updates(oracle):
    """p_sample / ddim_sample element-wise updates, given the oracle's own UNet output."""
    g = golden("sampler_steps")
    from sin3dm_amd.diffusion.respace import space_timesteps
    H, W, D = (int(v) for v in g["hwd"])
    sd = oracle.Params(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 0, as_torch=False))
    for tag, resp in (("full", None), ("r20", "20")):
        tab, tmap = oracle.schedule_tables(None if resp is None else space_timesteps(1000, resp))
        Tn = tab.shape[1]
        for ti in (Tn - 1, 1, 0):
            pre = f"{tag}.t{ti}"
            x, eps = g[pre + ".x"], g[pre + ".eps"]
            mo = oracle.unet_forward(sd, x, [tmap[ti]] * x.shape[0], H, W, D, 32)
            s, p = oracle.p_sample_update(mo, x, eps, tab, ti)
            assert relerr(s, g[pre + ".p_sample"]) < 5e-5 and relerr(p, g[pre + ".p_xstart"]) < 5e-5
            s, p = oracle.ddim_update(mo, x, eps, tab, ti)
            assert relerr(s, g[pre + ".ddim_sample"]) < 5e-5 and relerr(p, g[pre + ".ddim_xstart"]) < 5e-5
            s, _ = oracle.ddim_update(mo, x, eps, tab, ti, eta=0.7)
            assert relerr(s, g[pre + ".ddim_eta_sample"]) < 5e-5


BRANCH_TAGS = ["noclip", "eps", "eps_noclip", "eps_r20", "small", "small_r20_noclip", "eps_small"]


@pytest.mark.parametrize("tag", BRANCH_TAGS)
def test_sampler_update_branches(oracle, tag):
    """clip_denoised=False, ModelMeanType.EPSILON and ModelVarType.FIXED_SMALL (gaussian_diffusion.py:286-289, 294-315) at
    t in {T-1, 1, 0}: p_sample, ddim_sample (eta 0 and 0.7) and the model mean / (log-)variance of p_mean_variance."""
    g = golden("sampler_branches")
    from sin3dm_amd.diffusion.respace import space_timesteps
    H, W, D = (int(v) for v in g["hwd"])
    px, small, resp, clip = (int(v) for v in g[f"{tag}.cfg"])
    sd = oracle.Params(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 0, as_torch=False))
    tab, tmap = oracle.schedule_tables(space_timesteps(1000, str(resp)) if resp else None)
    Tn = tab.shape[1]
    kw = dict(clip=bool(clip), mean_eps=not px)
    for ti in (Tn - 1, 1, 0):
        pre = f"{tag}.t{ti}"
        x, eps = g[pre + ".x"], g[pre + ".eps"]
        # (1) the update arithmetic alone, on the reference's own model output: fp32 round-off only
        mo = g[pre + ".model_out"]
        s, p, m = oracle.p_sample_update(mo, x, eps, tab, ti, var_small=bool(small), want_mean=True, **kw)
        assert relerr(s, g[pre + ".p_sample"]) < 2e-6 and relerr(p, g[pre + ".p_xstart"]) < 2e-6 and relerr(m, g[pre + ".mean"]) < 2e-6
        s, p = oracle.ddim_update(mo, x, eps, tab, ti, **kw)
        assert relerr(s, g[pre + ".ddim_sample"]) < 2e-6 and relerr(p, g[pre + ".p_xstart"]) < 2e-6
        s, _ = oracle.ddim_update(mo, x, eps, tab, ti, eta=0.7, **kw)
        assert relerr(s, g[pre + ".ddim_eta_sample"]) < 2e-6
        # (2) end to end through the oracle's UNet.  An eps-derived x0 multiplies the UNet's round-off by
        # sqrt(1 / alphas_cumprod - 1) (157 at t = 999): the bound is the forward gate times that factor
        mo2 = oracle.unet_forward(sd, x, [tmap[ti]] * x.shape[0], H, W, D, 32)
        assert relerr(mo2, mo) < 5e-5
        amp = 1.0 if px else max(1.0, float(tab[4][ti]) * float(np.abs(mo).max()) / float(np.abs(g[pre + ".p_xstart"]).max()))
        s, p = oracle.p_sample_update(mo2, x, eps, tab, ti, var_small=bool(small), **kw)
        assert relerr(s, g[pre + ".p_sample"]) < 5e-5 * amp and relerr(p, g[pre + ".p_xstart"]) < 5e-5 * amp
        var = tab[5][max(ti, 1)] if small else (tab[5][1] if ti == 0 else tab[0][ti])
        assert abs(float(g[pre + ".log_variance"][0]) - np.log(var)) < 1e-5 * abs(np.log(var))
        assert abs(float(g[pre + ".variance"][0]) - (tab[5][ti] if small else var)) <= 1e-6 * var


@pytest.mark.parametrize("tag,resp,ddim", [("ddim10", "10", True), ("ddpm20", "20", False)])
def test_trajectories(oracle, tag, resp, ddim):
    g = golden("trajectories")
    from sin3dm_amd.diffusion.respace import space_timesteps
    H, W, D = (int(v) for v in g["hwd"])
    sd = oracle.Params(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 0, as_torch=False))
    tab, tmap = oracle.schedule_tables(space_timesteps(1000, resp))
    Tn = tab.shape[1]
    x = g[f"{tag}.xT"].copy()
    inter = []
    for k, ti in enumerate(range(Tn - 1, -1, -1)):
        mo = oracle.unet_forward(sd, x, [tmap[ti]] * x.shape[0], H, W, D, 32)
        upd = oracle.ddim_update if ddim else oracle.p_sample_update
        x, _ = upd(mo, x, g[f"{tag}.eps"][k], tab, ti)
        if k % 5 == 4 or k == Tn - 1:
            inter.append(x.copy())
    assert relerr(x, g[f"{tag}.final"]) < 2e-4
    assert relerr(np.stack(inter), g[f"{tag}.inter"]) < 2e-4
    assert np.all(x[..., H:, W:] == 0)


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_decoder(oracle, tag):
    g = golden("decoder")
    up, hid, H, W, D = (int(v) for v in g[f"{tag}.cfg"])
    sd = oracle.Params(T.synthetic_state_dict(T.ae_param_shapes(4, 8, up, hid, 4), 5, as_torch=False))
    fm = [g[f"{tag}.{p}"] for p in T.PLANES]
    geo = oracle.ae_plane_block(sd, "geo_convs", [f[:, :4] for f in fm], up)
    tex = oracle.ae_plane_block(sd, "tex_convs", [f[:, 4:] for f in fm], up)
    for p, a, b in zip(T.PLANES, geo, tex):
        assert relerr(a, g[f"{tag}.geo_{p}"]) < TOL and relerr(b, g[f"{tag}.tex_{p}"]) < TOL
    out = oracle.ae_decode(sd, g[f"{tag}.pts"], *fm, g[f"{tag}.aabb"], 4, 8, up, hid, 4)
    assert relerr(out, g[f"{tag}.out"]) < 5e-5
    out = oracle.ae_decode(sd, g[f"{tag}.pts"][:33], *fm, np.array([-1, -1, -1, 1, 1, 1], np.float32), 4, 8, up, hid, 4)
    assert relerr(out, g[f"{tag}.out_default_aabb"]) < 5e-5


def test_compose(oracle):
    g = golden("compose")
    H, W, D = (int(v) for v in g["hwd"])
    comp = oracle.compose(g["xy"], g["xz"], g["yz"])
    assert np.array_equal(comp, g["composed"])
    for a, b in zip(oracle.decompose(comp, H, W, D), (g["xy"], g["xz"], g["yz"])):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tag,mc,raw,ssn,cm", UNET_CASES)
def test_torch_port_unet_forward(oracle, tag, mc, raw, ssn, cm):
    """oracle/torch_port.py (bench.py's cpu_baseline engine) against the same golden vectors."""
    import torch
    import torch_port as tp
    g = golden("unet_fwd")
    sd = T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc, rollout=not raw, use_scale_shift_norm=ssn,
                                                    channel_mult=cm), 0)
    H, W, D = (int(v) for v in g[f"{tag}.hwd"])
    with torch.no_grad():
        y = tp.unet_forward(sd, torch.from_numpy(g[f"{tag}.x"]), torch.from_numpy(g[f"{tag}.t"]), H, W, D, mc, cm, ssn,
                            not raw).numpy()
    assert relerr(y, g[f"{tag}.y"]) < 5e-6


TRAIN_CASES = [("mc32_a", 32, False, True, (1, 2), 2), ("mc32_odd", 32, False, True, (1, 2), 2),
               ("mc32_raw", 32, True, True, (1, 2), 2), ("mc32_add", 32, False, False, (1, 2), 1),
               ("mc32_3lev", 32, False, True, (1, 2, 2), 1)]


def _train_inputs(g, tag, B):
    import torch
    H, W, D = (int(v) for v in g[f"{tag}.hwd"])
    x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1)
    noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 401))
    return H, W, D, x0, noise, torch.from_numpy(g[f"{tag}.t"])


@pytest.mark.parametrize("tag,mc,raw,ssn,cm,B", TRAIN_CASES)
def test_torch_port_training_losses_and_grads(oracle, tag, mc, raw, ssn, cm, B):
    """training tier oracle: loss terms and every parameter gradient against the reference's autograd."""
    import torch
    import torch_port as tp
    g = golden("train_grads")
    shapes = T.unet_param_shapes(model_channels=mc, rollout=not raw, use_scale_shift_norm=ssn, channel_mult=cm)
    sd = {k: v.requires_grad_(True) for k, v in T.synthetic_state_dict(shapes, 0).items()}
    H, W, D, x0, noise, t = _train_inputs(g, tag, B)
    tabs = oracle.schedule_tables_named(1000)
    terms, x_t = tp.training_losses(sd, x0, t, noise, tabs, H, W, D, model_channels=mc, channel_mult=cm,
                                    use_scale_shift_norm=ssn, rollout=not raw)
    assert relerr(x_t.detach().numpy(), g[f"{tag}.x_t"]) < 1e-6
    for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
        assert relerr(terms[k].detach().numpy(), g[f"{tag}.{k}"]) < 1e-5
    terms["loss"].mean().backward()
    w = digest_errors({k: v.grad.numpy() for k, v in sd.items()}, g, f"{tag}.grad")
    assert w["norm"] < 1e-4 and w["proj"] < 1e-4 and w["head"] < 1e-3 and w["full"] < 1e-4, w


@pytest.mark.parametrize("tag,ssn,B", [("mc32_a", True, 2), ("mc32_add", False, 1)])
def test_torch_port_training_epsilon_target(oracle, tag, ssn, B):
    """predict_xstart=False: the MSE target is the noise (gaussian_diffusion.py:829-835); loss terms and every parameter
    gradient against the reference's autograd."""
    import torch
    import torch_port as tp
    g = golden("train_eps")
    shapes = T.unet_param_shapes(model_channels=32, use_scale_shift_norm=ssn)
    sd = {k: v.requires_grad_(True) for k, v in T.synthetic_state_dict(shapes, 0).items()}
    H, W, D, x0, noise, t = _train_inputs(g, tag, B)
    tabs = oracle.schedule_tables_named(1000)
    terms, _ = tp.training_losses(sd, x0, t, noise, tabs, H, W, D, predict_xstart=False, model_channels=32,
                                  use_scale_shift_norm=ssn)
    for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
        assert relerr(terms[k].detach().numpy(), g[f"{tag}.{k}"]) < 1e-5
    terms["loss"].mean().backward()
    w = digest_errors({k: v.grad.numpy() for k, v in sd.items()}, g, f"{tag}.grad")
    assert w["norm"] < 1e-4 and w["proj"] < 1e-4 and w["head"] < 1e-3 and w["full"] < 1e-4, w


@pytest.mark.parametrize("wd_tag", ["wd0", "wd01"])
def test_torch_port_optimizer_steps(oracle, wd_tag):
    """three AdamW + EMA + lr-anneal steps as TrainLoop.run_step orders them."""
    import torch
    import torch_port as tp
    g = golden("train_steps")
    lr0, ema_rate, wd, anneal = (float(v) for v in g[f"{wd_tag}.hyper"])
    mc, B, (H, W, D) = 32, 2, (10, 14, 6)
    shapes = T.unet_param_shapes(model_channels=mc)
    sd = {k: v.requires_grad_(True) for k, v in T.synthetic_state_dict(shapes, 0).items()}
    init = {k: v.detach().clone() for k, v in sd.items()}
    names = list(shapes)
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    ema = [sd[k].detach().clone() for k in names]
    tabs = oracle.schedule_tables_named(1000)
    x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1)
    lr = lr0
    for step in range(3):
        noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 500 + step))
        t = torch.tensor([[700, 3], [12, 999], [450, 451]][step])
        for p in sd.values():
            p.grad = None
        terms, _ = tp.training_losses(sd, x0, t, noise, tabs, H, W, D, model_channels=mc)
        terms["loss"].mean().backward()
        assert relerr(terms["loss"].detach().numpy(), g[f"{wd_tag}.losses"][step]) < 1e-4
        with torch.no_grad():
            tp.adamw_ema_step([sd[k] for k in names], [sd[k].grad for k in names], m, v, ema, step + 1, lr, wd, ema_rate)
        lr = lr0 * (1 - step / anneal)
    wp = digest_errors({k: (sd[k].detach() - init[k]).numpy() for k in names}, g, f"{wd_tag}.dparam", zero_grad_params())
    we = digest_errors({k: (e - init[k]).numpy() for k, e in zip(names, ema)}, g, f"{wd_tag}.dema", zero_grad_params())
    # Adam turns a near-zero gradient element into a +-lr step: single elements may flip, the L2 error stays small
    assert wp["norm"] < 1e-3 and wp["proj"] < 2e-3 and wp["full_l2"] < 5e-3, wp
    assert we["norm"] < 1e-3 and we["proj"] < 2e-3, we


def test_torch_port_ae_training(oracle):
    """auto-encoder tier oracle: encode, forward, losses, every gradient and three AdamW/ExponentialLR steps against
    the reference's AutoEncoderGroupSkip under autograd."""
    import torch
    import torch_port as tp
    g = golden("ae_train")
    H, W, D, N = (int(v) for v in g["hwdn"])
    shapes = T.ae_param_shapes(with_encoder=True)
    sd = {k: v.requires_grad_(True) for k, v in T.synthetic_state_dict(shapes, 5).items()}
    vol = torch.tanh(torch.from_numpy(T.synthetic_noise((1, 4, 2 * H, 2 * W, 2 * D), 1200)))
    vol[:, 1:] = 0.5 * vol[:, 1:] + 0.5
    aabb, thr = torch.from_numpy(g["aabb"]), float(g["thr"])
    fm = tp.ae_encode(sd, vol)
    for f, k in zip(fm, ("xy", "xz", "yz")):
        assert relerr(f.detach().numpy(), g[k]) < 1e-5
    pts, sdf, tex = (torch.from_numpy(g[k]) for k in ("pts", "sdf", "tex"))
    pred = tp.ae_decode(sd, pts, fm, aabb)
    assert relerr(pred.detach().numpy(), g["pred"]) < 1e-5
    losses = tp.ae_losses(pred, sdf, tex, thr)
    assert abs(float(losses["sdf_loss"]) - float(g["sdf_loss"])) < 1e-6 and abs(float(losses["tex_loss"]) - float(g["tex_loss"])) < 1e-6
    sum(losses.values()).backward()
    w = digest_errors({k: v.grad.numpy() for k, v in sd.items()}, g, "grad")
    assert w["norm"] < 1e-4 and w["proj"] < 2e-4 and w["full"] < 2e-4, w
    # three steps: AdamW (weight_decay 0.01 = torch default), lr groups geo = lr*split / tex = lr, ExponentialLR
    lr, split, decay = (float(v) for v in g["steps.hyper"])
    sd = {k: v.requires_grad_(True) for k, v in T.synthetic_state_dict(shapes, 5).items()}
    init = {k: v.detach().clone() for k, v in sd.items()}
    geo, tex_names = tp.ae_param_groups(list(sd))
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items()}
    for step in range(3):
        rng = np.random.Generator(np.random.PCG64(int(g["steps.seeds"][step])))
        ext = np.asarray([0.7, 1.0, 0.45], np.float32)
        p = torch.from_numpy(rng.uniform(-1.1, 1.1, size=(N, 3)).astype(np.float32) * ext)
        s = torch.from_numpy(np.clip(rng.normal(0, 0.04, size=(N, 1)), -thr, thr).astype(np.float32))
        c = torch.from_numpy(rng.uniform(0, 1, size=(N, 3)).astype(np.float32))
        for v in sd.values():
            v.grad = None
        ls = tp.ae_losses(tp.ae_decode(sd, p, tp.ae_encode(sd, vol), aabb), s, c, thr)
        assert abs(float(ls["sdf_loss"]) - g["steps.losses"][step][0]) < 2e-5
        sum(ls.values()).backward()
        with torch.no_grad():
            for names, glr in ((geo, lr * split), (tex_names, lr)):
                tp.adamw_ema_step([sd[k] for k in names], [sd[k].grad for k in names], [state[k][0] for k in names],
                                  [state[k][1] for k in names], [], step + 1, glr * decay ** step, 0.01, 0.0)
    skip = zero_grad_params(None, "ae_train")
    assert skip == {"geo_convs.in_layers.0.bias", "tex_convs.in_layers.0.bias", "geo_encoder.bias", "tex_encoder.bias"}  # feed an InstanceNorm
    wp = digest_errors({k: (sd[k].detach() - init[k]).numpy() for k in sd}, g, "steps.dparam", skip)
    assert wp["norm"] < 2e-3 and wp["proj"] < 1e-2, wp
