#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (PyTorch-CPU).

Runs only in the build container, where /root/reference exists; the reference never travels to
the GPU box.  What is committed is data only: inputs + the reference's outputs.
Weights are NOT stored: they are regenerated bit-identically from sin3dm_amd.testing
(numpy PCG64 keyed by parameter name), and this script asserts that the name->shape manifest it
uses equals the reference modules' own state_dict.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_SRC = os.environ.get("SIN3DM_REFERENCE", "/root/reference") + "/src"
sys.path.insert(0, REPO)
sys.path.insert(0, REF_SRC)

from sin3dm_amd import testing as T  # noqa: E402

torch.set_num_threads(1)           # 1-thread oneDNN: the most reproducible reference arithmetic
torch.set_grad_enabled(False)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def rnd(shape, seed):
    return torch.from_numpy(T.synthetic_noise(shape, seed))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


def planes(B, C, H, W, D, seed):
    return (rnd((B, C, H, W), seed), rnd((B, C, H, D), seed + 1), rnd((B, C, W, D), seed + 2))


def load_synth(module, shapes, seed, prefix=""):
    sd = T.synthetic_state_dict(shapes, seed)
    ref_sd = module.state_dict()
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    assert set(sub) == set(ref_sd), (sorted(set(sub) ^ set(ref_sd))[:8])
    for k in sub:
        assert tuple(sub[k].shape) == tuple(ref_sd[k].shape), k
    module.load_state_dict(sub)
    module.eval()
    return sd


# ---------------------------------------------------------------------------------------------
def gen_schedules():
    from diffusion.script_util import create_gaussian_diffusion
    out = {}
    for tag, resp in (("full", ""), ("r100", "100"), ("r10", "10"), ("ddim50", "ddim50"), ("r20", "20")):
        d = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=True,
                                      timestep_respacing=resp)
        for f in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod"):
            out[f"{tag}.{f}"] = getattr(d, f)
        out[f"{tag}.timestep_map"] = np.asarray(d.timestep_map, dtype=np.int64)
    save("schedules", **out)


def gen_temb():
    from diffusion.nn import timestep_embedding
    out = {}
    t = torch.tensor([0, 1, 500, 999], dtype=torch.int64)
    out["t"] = t
    for mc in (32, 64):
        out[f"emb{mc}"] = timestep_embedding(t, mc)
    save("temb", **out)


def gen_leaves():
    import diffusion.unet_triplane as U
    out = {}
    for tag, (B, C, H, W, D) in (("a", (2, 32, 10, 14, 6)), ("b", (1, 64, 9, 13, 7))):
        fm = planes(B, C, H, W, D, 100)
        out[f"{tag}.in_xy"], out[f"{tag}.in_xz"], out[f"{tag}.in_yz"] = fm
        # TriplaneNorm + SiLU  (unet_triplane.py:63-95)
        m = torch.nn.Sequential(U.TriplaneNorm(C), U.TriplaneSiLU())
        shapes = {f"0.norm_{p}.{l}": (C,) for p in T.PLANES for l in ("weight", "bias")}
        load_synth(m, shapes, 1)
        for p, y in zip(T.PLANES, m(fm)):
            out[f"{tag}.normsilu_{p}"] = y
        # TriplaneConv 3x3 rollout / 3x3 plain / 1x1 (unet_triplane.py:21-60)
        for name, k, roll, cout in (("conv3r", 3, True, 48), ("conv3", 3, False, 48), ("conv1", 1, False, 40)):
            m = U.TriplaneConv(C, cout, k, padding=k // 2, is_rollout=roll)
            shapes = {}
            for p in T.PLANES:
                shapes[f"conv_{p}.weight"] = (cout, C * 3 if roll else C, k, k)
                shapes[f"conv_{p}.bias"] = (cout,)
            load_synth(m, shapes, 2)
            for p, y in zip(T.PLANES, m(fm)):
                out[f"{tag}.{name}_{p}"] = y
        for p, y in zip(T.PLANES, U.TriplaneDownsample2x()(fm)):
            out[f"{tag}.down_{p}"] = y
        for p, y in zip(T.PLANES, U.TriplaneUpsample2x()(fm)):
            out[f"{tag}.up_{p}"] = y
        # size-targeted bilinear resize (unet_triplane.py:494-499): to (2H+1, 2W+1) etc.
        tgt = {"xy": (2 * H + 1, 2 * W + 1), "xz": (2 * H + 1, 2 * D + 1), "yz": (2 * W + 1, 2 * D + 1)}
        for p, x in zip(T.PLANES, fm):
            out[f"{tag}.resize_{p}"] = torch.nn.functional.interpolate(x, size=tgt[p], mode="bilinear",
                                                                     align_corners=False)
    save("leaves", **out)


def gen_resblock():
    import diffusion.unet_triplane as U
    out = {}
    B, H, W, D = 2, 10, 14, 6
    for tag, (C, Cout, ssn) in (("same", (32, 32, True)), ("skip", (32, 64, True)), ("add", (32, 32, False))):
        m = U.TriplaneResBlock(C, 128, 0, out_channels=Cout, use_scale_shift_norm=ssn)
        full = T.unet_param_shapes(model_channels=32, use_scale_shift_norm=ssn)
        # borrow the names of an equivalent block
        shapes = {}
        for k, v in m.state_dict().items():
            shapes[k] = tuple(v.shape)
        load_synth(m, shapes, 3)
        fm = planes(B, C, H, W, D, 200)
        emb = rnd((B, 128), 210)
        out[f"{tag}.in_xy"], out[f"{tag}.in_xz"], out[f"{tag}.in_yz"] = fm
        out[f"{tag}.emb"] = emb
        for p, y in zip(T.PLANES, m(fm, emb)):
            out[f"{tag}.out_{p}"] = y
    save("resblock", **out)


def make_unet(mc, raw=False, ssn=True, channel_mult="1,2"):
    from diffusion.script_util import create_model_and_diffusion_from_args
    args = SimpleNamespace(learn_sigma=False, steps=1000, noise_schedule="linear", timestep_respacing="",
                           use_kl=False, predict_xstart=True, rescale_timesteps=False,
                           rescale_learned_sigmas=False, in_channels=12, model_channels=mc, out_channels=12,
                           num_res_blocks=1, dropout=0, channel_mult=channel_mult, use_checkpoint=False,
                           use_fp16=False, use_scale_shift_norm=ssn,
                           diff_net_type="unet_raw" if raw else "unet_small")
    model, _ = quiet(create_model_and_diffusion_from_args, args)
    shapes = T.unet_param_shapes(model_channels=mc, rollout=not raw, use_scale_shift_norm=ssn,
                                 channel_mult=channel_mult)
    load_synth(model, shapes, 0)
    return model


def make_diffusion(resp):
    from diffusion.script_util import create_gaussian_diffusion
    return create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=True,
                                     timestep_respacing=resp)


def gen_unet():
    out = {}
    cases = (("mc32_a", 32, False, (2, 10, 14, 6), True, "1,2"),
             ("mc32_odd", 32, False, (2, 9, 13, 7), True, "1,2"),
             ("mc64_b", 64, False, (1, 12, 8, 10), True, "1,2"),
             ("mc32_raw", 32, True, (2, 10, 14, 6), True, "1,2"),
             ("mc32_add", 32, False, (1, 10, 14, 6), False, "1,2"),
             ("mc32_3lev", 32, False, (1, 12, 16, 8), True, "1,2,2"))
    for tag, mc, raw, (B, H, W, D), ssn, cm in cases:
        model = make_unet(mc, raw, ssn, cm)
        x = rnd((B, 12, H + D, W + D), 300)
        t = torch.tensor([999, 17][:B], dtype=torch.int64)
        y = model(x, t, H=H, W=W, D=D)
        out[f"{tag}.x"], out[f"{tag}.t"], out[f"{tag}.y"] = x, t, y
        out[f"{tag}.hwd"] = np.asarray([H, W, D])
    save("unet_fwd", **out)

    # name -> shape manifest for mc in {32, 64, 128} straight from the reference modules
    man = {}
    for mc in (32, 64, 128):
        from diffusion.unet_triplane import TriplaneUNetModelSmall
        m = quiet(TriplaneUNetModelSmall, 12, mc, 12, use_scale_shift_norm=True)
        for k, v in m.state_dict().items():
            man[f"mc{mc}/{k}"] = np.asarray(v.shape, dtype=np.int64)
        mine = T.unet_param_shapes(model_channels=mc)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(mine)
    save("unet_manifest", **man)


def gen_sampler():
    out = {}
    model = make_unet(32)
    H, W, D = 10, 14, 6
    kw = dict(H=H, W=W, D=D)
    B = 2
    shape = (B, 12, H + D, W + D)
    # single steps with stored noise: patch randn_like so the reference consumes OUR eps
    import diffusion.gaussian_diffusion as gd
    for resp_tag, resp in (("full", ""), ("r20", "20")):
        diff = make_diffusion(resp)
        Tn = diff.num_timesteps
        for ti in (Tn - 1, 1, 0):
            x = rnd(shape, 400 + ti)
            eps = rnd(shape, 500 + ti)
            t = torch.tensor([ti] * B)
            orig = gd.th.randn_like
            gd.th.randn_like = lambda z: eps.clone()
            try:
                o1 = diff.p_sample(model, x, t, model_kwargs=kw)
                o2 = diff.ddim_sample(model, x, t, model_kwargs=kw)
                o3 = diff.ddim_sample(model, x, t, model_kwargs=kw, eta=0.7)
            finally:
                gd.th.randn_like = orig
            pre = f"{resp_tag}.t{ti}"
            out[pre + ".x"], out[pre + ".eps"] = x, eps
            out[pre + ".p_sample"], out[pre + ".p_xstart"] = o1["sample"], o1["pred_xstart"]
            out[pre + ".ddim_sample"], out[pre + ".ddim_xstart"] = o2["sample"], o2["pred_xstart"]
            out[pre + ".ddim_eta_sample"] = o3["sample"]
    out["hwd"] = np.asarray([H, W, D])
    save("sampler_steps", **out)

    # trajectories: store x_T and every eps explicitly (RNG streams are not portable across torch versions)
    tr = {}
    for tag, resp, ddim in (("ddim10", "10", True), ("ddpm20", "20", False)):
        diff = make_diffusion(resp)
        Tn = diff.num_timesteps
        xT = rnd(shape, 600)
        epss = [rnd(shape, 700 + i) for i in range(Tn)]
        it = iter(epss)
        orig = gd.th.randn_like
        gd.th.randn_like = lambda z: next(it).clone()
        try:
            fn = diff.ddim_sample_loop_progressive if ddim else diff.p_sample_loop_progressive
            inter = []
            for k, o in enumerate(fn(model, shape, noise=xT.clone(), model_kwargs=kw)):
                if k % 5 == 4 or k == Tn - 1:
                    inter.append(o["sample"].clone())
                final = o["sample"]
        finally:
            gd.th.randn_like = orig
        tr[f"{tag}.xT"] = xT
        tr[f"{tag}.eps"] = torch.stack(epss)
        tr[f"{tag}.inter"] = torch.stack(inter)
        tr[f"{tag}.final"] = final
        corner = final[..., H:, W:]
        assert float(corner.abs().max()) == 0.0, "DxD corner must end at exactly 0"
    tr["hwd"] = np.asarray([H, W, D])
    save("trajectories", **tr)


def gen_sampler_branches():
    """Reachable sampler / training branches the default fixtures do not touch:
    clip_denoised=False (the reference's own trainer samples that way, train_util.py:181), predict_xstart=False
    (ModelMeanType.EPSILON: p_mean_variance :306-315 through _predict_xstart_from_eps, and the training target
    :829-835 with gradients), sigma_small=True (ModelVarType.FIXED_SMALL :286-289); each at t in {T-1, 1, 0}."""
    import diffusion.gaussian_diffusion as gd
    from diffusion.script_util import create_gaussian_diffusion
    out = {}
    model = make_unet(32)
    H, W, D = 10, 14, 6
    kw = dict(H=H, W=W, D=D)
    B = 1
    shape = (B, 12, H + D, W + D)
    cfgs = (("noclip", dict(predict_xstart=True), "", False),
            ("eps", dict(predict_xstart=False), "", True),
            ("eps_noclip", dict(predict_xstart=False), "", False),
            ("eps_r20", dict(predict_xstart=False), "20", True),
            ("small", dict(predict_xstart=True, sigma_small=True), "", True),
            ("small_r20_noclip", dict(predict_xstart=True, sigma_small=True), "20", False),
            ("eps_small", dict(predict_xstart=False, sigma_small=True), "", True))
    for tag, dkw, resp, clip in cfgs:
        diff = create_gaussian_diffusion(steps=1000, noise_schedule="linear", timestep_respacing=resp, **dkw)
        Tn = diff.num_timesteps
        for ti in (Tn - 1, 1, 0):
            x = rnd(shape, 1500 + ti)
            if "eps" in tag:
                x = x * 0.5           # keeps the eps-derived x0 = (x - sqrt(1-ab) eps) / sqrt(ab) partly inside [-1, 1] at small t
            eps = rnd(shape, 1600 + ti)
            t = torch.tensor([ti] * B)
            orig = gd.th.randn_like
            gd.th.randn_like = lambda z: eps.clone()
            try:
                o1 = diff.p_sample(model, x, t, clip_denoised=clip, model_kwargs=kw)
                o2 = diff.ddim_sample(model, x, t, clip_denoised=clip, model_kwargs=kw)
                o3 = diff.ddim_sample(model, x, t, clip_denoised=clip, model_kwargs=kw, eta=0.7)
                pm = diff.p_mean_variance(model, x, t, clip_denoised=clip, model_kwargs=kw)
                mo = model(x, diff._scale_timesteps(torch.tensor([diff.timestep_map[ti]] * B)), **kw)
            finally:
                gd.th.randn_like = orig
            pre = f"{tag}.t{ti}"
            out[pre + ".x"], out[pre + ".eps"] = x, eps
            out[pre + ".model_out"] = mo                     # the update arithmetic can be pinned without the UNet's round-off (EPSILON amplifies it by sqrt(1/ac - 1), 157 at t = 999)
            assert torch.equal(o1["pred_xstart"], o2["pred_xstart"])
            out[pre + ".p_sample"], out[pre + ".p_xstart"] = o1["sample"], o1["pred_xstart"]
            out[pre + ".ddim_sample"] = o2["sample"]
            out[pre + ".ddim_eta_sample"] = o3["sample"]
            out[pre + ".mean"] = pm["mean"]
            out[pre + ".variance"], out[pre + ".log_variance"] = pm["variance"][:, 0, 0, 0], pm["log_variance"][:, 0, 0, 0]
        out[f"{tag}.cfg"] = np.asarray([int(dkw.get("predict_xstart", False)), int(dkw.get("sigma_small", False)),
                                        int(resp or 0), int(clip)])
    out["hwd"] = np.asarray([H, W, D])
    save("sampler_branches", **out)

    # a short unclipped ancestral trajectory, as TrainLoop._sample_and_visualize runs it (train_util.py:181)
    tr = {}
    for tag, dkw, resp in (("noclip_ddpm20", dict(predict_xstart=True), "20"), ("eps_ddim10", dict(predict_xstart=False), "10")):
        diff = create_gaussian_diffusion(steps=1000, noise_schedule="linear", timestep_respacing=resp, **dkw)
        Tn = diff.num_timesteps
        xT = rnd(shape, 1700)
        epss = [rnd(shape, 1710 + i) for i in range(Tn)]
        it = iter(epss)
        orig = gd.th.randn_like
        gd.th.randn_like = lambda z: next(it).clone()
        try:
            if "ddim" in tag:
                final = diff.ddim_sample_loop(model, shape, noise=xT.clone(), clip_denoised=True, model_kwargs=kw)
            else:
                final = diff.p_sample_loop(model, shape, noise=xT.clone(), clip_denoised=False, model_kwargs=kw)
        finally:
            gd.th.randn_like = orig
        tr[f"{tag}.xT"], tr[f"{tag}.eps"], tr[f"{tag}.final"] = xT, torch.stack(epss), final
    tr["hwd"] = np.asarray([H, W, D])
    save("trajectories_branches", **tr)

    # training_losses with the EPSILON target (:829-835): loss terms and every parameter gradient
    torch.set_grad_enabled(True)
    out = {}
    diffusion = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=False)
    for tag, mc, raw, (B, H, W, D), ssn, cm in (("mc32_a", 32, False, (2, 10, 14, 6), True, "1,2"),
                                                ("mc32_add", 32, False, (1, 10, 14, 6), False, "1,2")):
        model = make_unet(mc, raw, ssn, cm)
        model.train()
        x0 = rnd((B, 12, H + D, W + D), 400).clamp(-1, 1)
        noise = rnd((B, 12, H + D, W + D), 401)
        t = torch.tensor([700, 3][:B], dtype=torch.int64)
        terms = diffusion.training_losses(model, x0, t, model_kwargs=dict(H=H, W=W, D=D), noise=noise)
        model.zero_grad()
        (terms["loss"] * torch.ones(B)).mean().backward()
        out[f"{tag}.t"] = t
        out[f"{tag}.hwd"] = np.asarray([H, W, D])
        for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
            out[f"{tag}.{k}"] = terms[k].detach()
        grad_digest({k: p.grad for k, p in model.named_parameters()}, f"{tag}.grad", out,
                    full_names=TRAIN_FULL if tag == "mc32_a" else ())
    save("train_eps", **out)
    torch.set_grad_enabled(False)


def gen_decoder():
    from encoding.networks import AutoEncoderGroupSkip
    from encoding.blocks import TriplaneGroupResnetBlock
    out = {}
    for tag, (up, hid, H, W, D) in (("small", (16, 32, 10, 14, 6)), ("wide", (64, 256, 9, 8, 11))):
        net = AutoEncoderGroupSkip(4, 8, up, hid, 4)
        shapes = T.ae_param_shapes(4, 8, up, hid, 4)
        sd = T.synthetic_state_dict(shapes, 5)
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith(("geo_encoder", "tex_encoder", "aabb")) for k in missing), missing
        net.eval()
        fm = [0.8 * torch.tanh(x) for x in planes(1, 12, H, W, D, 800)]
        aabb = torch.tensor([-0.7, -1.0, -0.45, 0.7, 1.0, 0.45])
        g = np.random.Generator(np.random.PCG64(900))
        pts = torch.from_numpy(g.uniform(-1.15, 1.15, size=(257, 3)).astype(np.float32)) * aabb[3:]
        y = net.decode(pts, fm, aabb=aabb)
        y_default = net.decode(pts[:33], fm)            # module buffer aabb = [-1,-1,-1,1,1,1]
        out[f"{tag}.xy"], out[f"{tag}.xz"], out[f"{tag}.yz"] = fm
        out[f"{tag}.aabb"], out[f"{tag}.pts"], out[f"{tag}.out"] = aabb, pts, y
        out[f"{tag}.out_default_aabb"] = y_default
        geo = net.geo_convs([f[:, :4] for f in fm])
        tex = net.tex_convs([f[:, 4:] for f in fm])
        for p, a, b in zip(T.PLANES, geo, tex):
            out[f"{tag}.geo_{p}"], out[f"{tag}.tex_{p}"] = a, b
        out[f"{tag}.cfg"] = np.asarray([up, hid, H, W, D])
    save("decoder", **out)


def gen_compose():
    from utils.triplane_util import compose_featmaps, decompose_featmaps
    fm = planes(2, 3, 5, 7, 4, 950)
    comp, (H, W, D) = compose_featmaps(*fm)
    back = decompose_featmaps(comp, (H, W, D))
    for a, b in zip(fm, back):
        assert torch.equal(a, b)
    save("compose", xy=fm[0], xz=fm[1], yz=fm[2], composed=comp, hwd=np.asarray([H, W, D]))


def grad_digest(named, prefix, out, full_max=512, full_names=()):
    """Digest of an ordered {name: tensor} map (order = the module's named_parameters()): per tensor the L2 norm, the
    projection on a fixed pseudo-random direction (keyed by the name, sin3dm_amd.testing.synthetic_tensor) and the
    first 8 elements, stacked into three arrays; the tensor itself when small or explicitly listed."""
    norms, projs, heads = [], [], []
    for k, g in named.items():
        g = g.detach().double().reshape(-1)
        r = torch.from_numpy(T.synthetic_tensor("digest/" + k, (g.numel(),), 7)).double()
        norms.append(float(g.norm()))
        projs.append(float((g * r).sum()))
        heads.append(torch.nn.functional.pad(g[:8], (0, max(0, 8 - g.numel()))).float().numpy())
        if g.numel() <= full_max or k in full_names:
            out[f"{prefix}/full/{k}"] = g.float()
    out[f"{prefix}/names"] = np.asarray(list(named))
    out[f"{prefix}/norm"] = np.asarray(norms)
    out[f"{prefix}/proj"] = np.asarray(projs)
    out[f"{prefix}/head"] = np.stack(heads)


TRAIN_FULL = ("input_blocks.0.0.in_layers.2.conv_xy.weight", "output_blocks.1.0.skip_connection.conv_yz.weight",
              "input_blocks.1.1.emb_layers.1.weight", "in_conv.0.conv_xz.weight", "out.2.conv_yz.weight")


def gen_train():
    """training_losses + backward (gaussian_diffusion.py:771-856, train_util.py:205-236) and three AdamW/EMA/anneal
    steps (train_util.py:163-172, 238-247; nn.py:55-65) on the tiny configs."""
    torch.set_grad_enabled(True)
    out = {}
    diffusion = make_diffusion("")
    cases = (("mc32_a", 32, False, (2, 10, 14, 6), True, "1,2"),
             ("mc32_odd", 32, False, (2, 9, 13, 7), True, "1,2"),
             ("mc32_raw", 32, True, (2, 10, 14, 6), True, "1,2"),
             ("mc32_add", 32, False, (1, 10, 14, 6), False, "1,2"),
             ("mc32_3lev", 32, False, (1, 12, 16, 8), True, "1,2,2"))
    for tag, mc, raw, (B, H, W, D), ssn, cm in cases:
        model = make_unet(mc, raw, ssn, cm)
        model.train()
        x0 = rnd((B, 12, H + D, W + D), 400).clamp(-1, 1)
        noise = rnd((B, 12, H + D, W + D), 401)
        t = torch.tensor([700, 3][:B], dtype=torch.int64)
        w = torch.ones(B)
        terms = diffusion.training_losses(model, x0, t, model_kwargs=dict(H=H, W=W, D=D), noise=noise)
        loss = (terms["loss"] * w).mean()
        model.zero_grad()
        loss.backward()
        out[f"{tag}.t"] = t                                # x0 / noise: synthetic_noise seeds 400 / 401
        out[f"{tag}.hwd"] = np.asarray([H, W, D])
        for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
            out[f"{tag}.{k}"] = terms[k].detach()
        out[f"{tag}.x_t"] = diffusion.q_sample(x0, t, noise=noise)
        grad_digest({k: p.grad for k, p in model.named_parameters()}, f"{tag}.grad", out,
                    full_names=TRAIN_FULL if tag == "mc32_a" else ())
    save("train_grads", **out)

    # three optimizer steps exactly as TrainLoop.run_step does them (AdamW -> EMA -> linear anneal), fixed t / noise
    out = {}
    tag, mc, (B, H, W, D) = "mc32_a", 32, (2, 10, 14, 6)
    lr0, ema_rate, wd, anneal = 5e-4, 0.99, 0.0, 10        # ema 0.99 (default 0.9999) so that three steps move the EMA by more than fp32 round-off
    from diffusion.nn import update_ema
    import copy
    for wd_tag, wd in (("wd0", 0.0), ("wd01", 0.01)):
        model = make_unet(mc)
        model.train()
        params = list(model.parameters())
        names = [k for k, _ in model.named_parameters()]
        init = [p.detach().clone() for p in params]
        opt = torch.optim.AdamW(params, lr=lr0, weight_decay=wd)
        ema = copy.deepcopy(params)
        x0 = rnd((B, 12, H + D, W + D), 400).clamp(-1, 1)
        losses = []
        for step in range(3):
            noise = rnd((B, 12, H + D, W + D), 500 + step)
            t = torch.tensor([[700, 3], [12, 999], [450, 451]][step], dtype=torch.int64)
            for p in params:
                if p.grad is not None:
                    p.grad.detach_(); p.grad.zero_()
            terms = diffusion.training_losses(model, x0, t, model_kwargs=dict(H=H, W=W, D=D), noise=noise)
            (terms["loss"] * torch.ones(B)).mean().backward()
            opt.step()
            update_ema(ema, params, rate=ema_rate)
            lr = lr0 * (1 - step / anneal)                     # _anneal_lr runs after the step, with self.step = step
            for gr in opt.param_groups:
                gr["lr"] = lr
            losses.append(terms["loss"].detach().clone())
        out[f"{wd_tag}.losses"] = torch.stack(losses)
        out[f"{wd_tag}.hyper"] = np.asarray([lr0, ema_rate, wd, anneal])
        grad_digest({k: p.detach() - i for k, p, i in zip(names, params, init)}, f"{wd_tag}.dparam", out, full_names=TRAIN_FULL[:1])
        grad_digest({k: e.detach() - i for k, e, i in zip(names, ema, init)}, f"{wd_tag}.dema", out, full_max=64)
    save("train_steps", **out)
    torch.set_grad_enabled(False)


def ae_losses(pred, sdf, tex, sdf_threshold, tex_threshold_ratio=0.999, tex_weight=1.0):
    """ShapeAutoEncoder._forward_batch with the default sdf_loss="weightedl1", tex_loss="l1", data_type "sdftex",
    sdf_renorm=0 (src/encoding/model.py:186-237) — restated here because encoding.model does not import without
    tensorboardX/open3d; the network forward/backward below is the reference's own."""
    pred_sdf = pred[..., :1]
    weight = 1 + 0.5 * torch.sign(sdf) * torch.sign(sdf - pred_sdf)
    sdf_loss = ((pred_sdf - sdf).abs() * weight).mean()
    mask = sdf.squeeze(1).abs() < sdf_threshold * tex_threshold_ratio
    tex_loss = torch.nn.functional.l1_loss(pred[..., 1:][mask], tex[mask]) * tex_weight
    return {"sdf_loss": sdf_loss, "tex_loss": tex_loss}


def gen_ae_train():
    """Auto-encoder stage: encode (networks.py:164-180), forward + losses + backward, and three optimizer steps as
    ShapeAutoEncoder.train runs them (model.py:129-139, 178-184, 239-258: AdamW with two lr groups + ExponentialLR)."""
    from encoding.networks import AutoEncoderGroupSkip
    torch.set_grad_enabled(True)
    out = {}
    H, W, D, N = 12, 16, 10, 192
    shapes = T.ae_param_shapes(with_encoder=True)
    thr = 0.05

    def make():
        net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
        sd = T.synthetic_state_dict(shapes, 5)
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected and missing == ["aabb"], (missing, unexpected)
        aabb = torch.tensor([-0.7, -1.0, -0.45, 0.7, 1.0, 0.45])
        net.reset_aabb(aabb)
        net.train()
        return net, aabb

    def relu_margin(net, vol, pts):
        """smallest |pre-activation| over the hidden ReLUs of both MLPs and their gradient-carrying inputs: a batch whose
        margin is at round-off level has an ill-defined gradient (relu'(0+-eps)) and cannot pin an implementation"""
        margins = []
        hooks = [m.register_forward_hook(lambda mod, i, o: margins.append(float(i[0].abs().min())))
                 for m in net.modules() if isinstance(m, torch.nn.ReLU)]
        with torch.no_grad():
            net(vol, pts)
        for h in hooks:
            h.remove()
        return min(margins)

    def pick_batch(net, vol, seed0):
        for seed in range(seed0, seed0 + 50):
            b = batch(seed)
            if relu_margin(net, vol, b[0]) > 2e-6:          # ~10x the fp32 differences between implementations
                return seed, b
        raise RuntimeError("no well-conditioned batch found")

    def batch(seed):
        g = np.random.Generator(np.random.PCG64(seed))
        aabb = np.asarray([0.7, 1.0, 0.45], np.float32)
        pts = torch.from_numpy(g.uniform(-1.1, 1.1, size=(N, 3)).astype(np.float32) * aabb)
        sdf = torch.from_numpy(np.clip(g.normal(0, 0.04, size=(N, 1)), -thr, thr).astype(np.float32))
        tex = torch.from_numpy(g.uniform(0, 1, size=(N, 3)).astype(np.float32))
        return pts, sdf, tex

    vol = torch.tanh(rnd((1, 4, 2 * H, 2 * W, 2 * D), 1200))
    vol[:, 1:] = 0.5 * vol[:, 1:] + 0.5
    net, aabb = make()
    with contextlib.redirect_stdout(io.StringIO()):
        net, aabb = make()
    fm = net.encode(vol)
    seed, (pts, sdf, tex) = pick_batch(net, vol, 1300)
    out["seed"] = np.asarray(seed)
    pred = net(vol, pts)
    losses = ae_losses(pred, sdf, tex, thr)
    net.zero_grad()
    sum(losses.values()).backward()
    out["hwdn"] = np.asarray([H, W, D, N])
    out["aabb"], out["thr"] = aabb, np.asarray(thr)
    out["pts"], out["sdf"], out["tex"] = pts, sdf, tex
    out["xy"], out["xz"], out["yz"] = [f.detach() for f in fm]
    out["pred"] = pred.detach()
    out["sdf_loss"], out["tex_loss"] = losses["sdf_loss"].detach(), losses["tex_loss"].detach()
    grad_digest({k: p.grad for k, p in net.named_parameters()}, "grad", out, full_max=512,
                full_names=("geo_encoder.weight", "tex_encoder.weight", "geo_convs.shortcut.weight", "tex_convs.in_layers.0.weight"))

    # three training iterations: AdamW(two groups: geo lr*split, tex lr; default weight_decay 0.01) + ExponentialLR
    with contextlib.redirect_stdout(io.StringIO()):
        net, aabb = make()
    lr, split, decay = 5e-3, 0.2, 0.1 ** (1 / 10)
    opt = torch.optim.AdamW([{"params": net.geo_parameters(), "lr": lr * split}, {"params": net.tex_parameters(), "lr": lr}], lr)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, decay)
    init = {k: p.detach().clone() for k, p in net.named_parameters()}
    ls, seeds = [], []
    for step in range(3):
        seed, (pts, sdf, tex) = pick_batch(net, vol, 1400 + 50 * step)
        seeds.append(seed)
        pred = net(vol, pts)
        losses = ae_losses(pred, sdf, tex, thr)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        sched.step()
        ls.append([float(losses["sdf_loss"]), float(losses["tex_loss"])])
    out["steps.losses"] = np.asarray(ls)
    out["steps.seeds"] = np.asarray(seeds)
    out["steps.hyper"] = np.asarray([lr, split, decay])
    grad_digest({k: p.detach() - init[k] for k, p in net.named_parameters()}, "steps.dparam", out, full_max=64)
    save("ae_train", **out)
    torch.set_grad_enabled(False)


def gen_respaced_train():
    """SpacedDiffusion.training_losses (respace.py:93-96) with timestep_respacing="50" and rescale_timesteps=True: the
    denoiser must be conditioned on timestep_map[t] * 1000 / original_num_steps during TRAINING too."""
    from diffusion.script_util import create_gaussian_diffusion
    torch.set_grad_enabled(True)
    out = {}
    tag, mc, (B, H, W, D) = "mc32_a", 32, (2, 10, 14, 6)
    for dtag, resp, rescale, steps in (("r50_rescale", "50", True, 1000), ("s200_rescale", "", True, 200), ("r50", "50", False, 1000)):
        diffusion = create_gaussian_diffusion(steps=steps, noise_schedule="linear", predict_xstart=True,
                                              timestep_respacing=resp, rescale_timesteps=rescale)
        model = make_unet(mc)
        model.train()
        x0 = rnd((B, 12, H + D, W + D), 400).clamp(-1, 1)
        noise = rnd((B, 12, H + D, W + D), 401)
        t = torch.tensor([diffusion.num_timesteps - 3, 1], dtype=torch.int64)
        terms = diffusion.training_losses(model, x0, t, model_kwargs=dict(H=H, W=W, D=D), noise=noise)
        model.zero_grad()
        (terms["loss"] * torch.ones(B)).mean().backward()
        out[f"{dtag}.t"] = t
        out[f"{dtag}.cfg"] = np.asarray([steps, int(rescale), diffusion.num_timesteps])
        for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
            out[f"{dtag}.{k}"] = terms[k].detach()
        grad_digest({k: p.grad for k, p in model.named_parameters()}, f"{dtag}.grad", out, full_max=0)
    out["hwd"] = np.asarray([H, W, D])
    save("train_respaced", **out)
    torch.set_grad_enabled(False)


def gen_formats():
    """On-disk formats WRITTEN BY THE REFERENCE'S OWN CODE, kept as files under tests/golden/formats/:
      feat.npz                 utils.triplane_util.save_triplane_data             (triplane_util.py:38-41)
      encoding_args.json,      utils.parser_util.train_args() on the reference's parser: get_args_by_group dumps
      diffusion_args.json      (parser_util.py:102-145; argv below)
      ema_0.9999_000003.pt     a TriplaneUNetModelSmall.state_dict() saved as TrainLoop.save does (train_util.py:258-270:
                               `th.save(state_dict, path)` of name -> tensor)
      loaded.npz               what the reference's own readers return for those files (load_triplane_data composed map;
                               sample_args() namespace after load_and_overwrite_args)."""
    import json
    import shutil
    import tempfile
    from utils import parser_util as rpu, triplane_util as rtu
    dst = os.path.join(HERE, "formats")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    tmp = tempfile.mkdtemp()
    tag = os.path.join(tmp, "exp")
    argv = ["train.py", "--tag", tag, "--data_path", "data/towerruins.npz", "--fm_reso", "64", "--model_channels", "32", "--channel_mult", "1",
            "--enc_net_type", "skip", "--diff_n_iters", "3", "--ema_rate", "0.9999", "--use_scale_shift_norm", "True",
            "--enc_lr_split", "0.2", "--timestep_respacing", "100"]
    old = sys.argv
    sys.argv = argv
    try:
        args = quiet(rpu.train_args)
    finally:
        sys.argv = old
    H, W, D = 6, 8, 5
    fm = [np.tanh(T.synthetic_noise(s, 970 + i)) for i, s in enumerate(((12, H, W), (12, H, D), (12, W, D)))]
    rtu.save_triplane_data(rpu.encoding_feat_path(tag), *fm)
    model = make_unet(32, channel_mult="1")                 # one level: keeps the committed checkpoint at ~1.4 MB
    torch.save(model.state_dict(), rpu.diffusion_model_path(tag, args.ema_rate, 3))
    shutil.copy(rpu.encoding_feat_path(tag), os.path.join(dst, "feat.npz"))
    shutil.copy(os.path.join(rpu.encoding_log_dir(tag), "args.json"), os.path.join(dst, "encoding_args.json"))
    shutil.copy(os.path.join(rpu.diffusion_log_dir(tag), "args.json"), os.path.join(dst, "diffusion_args.json"))
    shutil.copy(rpu.diffusion_model_path(tag, args.ema_rate, 3), os.path.join(dst, "ema_0.9999_000003.pt"))
    # the reference's readers on its own files
    comp, sizes = rtu.load_triplane_data(rpu.encoding_feat_path(tag), device="cpu")
    sys.argv = ["sample.py", "--tag", tag, "--n_samples", "2", "--timestep_respacing", "10", "--resize", "1", "1.5", "1"]
    try:
        sargs = quiet(rpu.sample_args)
    finally:
        sys.argv = old
    ns = {k: v for k, v in vars(sargs).items() if k != "tag"}
    with open(os.path.join(dst, "sample_args_namespace.json"), "w") as f:
        json.dump(ns, f, indent=1, sort_keys=True)
    tns = {k: v for k, v in vars(args).items() if k != "tag"}
    with open(os.path.join(dst, "train_args_namespace.json"), "w") as f:
        json.dump(tns, f, indent=1, sort_keys=True)
    x = rnd((1, 12, H + D, W + D), 971)
    y = model(x, torch.tensor([321]), H=H, W=W, D=D)
    save("formats/loaded", composed=comp, sizes=np.asarray(sizes), x=x, y=y,
         model_path=np.asarray(os.path.relpath(rpu.diffusion_model_path(tag, args.ema_rate, 3), tag)),
         feat_path=np.asarray(os.path.relpath(rpu.encoding_feat_path(tag), tag)))
    shutil.rmtree(tmp)
    for f in sorted(os.listdir(dst)):
        print(f"formats/{f}: {os.path.getsize(os.path.join(dst, f)) / 1024:.1f} KiB")


def gen_ae_ckpt():
    """formats/ckpt_final.pth: the dict ShapeAutoEncoder.save_ckpt writes (src/encoding/model.py:141-156 — {net, optimizer,
    scheduler, Ka, Kd, Ks, Ns, aabb, featmap_size}), built from the reference's OWN objects: AutoEncoderGroupSkip.state_dict(),
    the two-group AdamW of _set_optimizer (:129-139) after one step, its ExponentialLR.  ShapeAutoEncoder itself imports
    tensorboard / mcubes at module level and cannot be imported here, hence the dict is assembled with its exact keys.  A
    small network (up 16, hidden 32) keeps the file small.  formats/ckpt_decode.npz: what the reference's
    net.decode returns for that checkpoint (the load test's expectation)."""
    from encoding.networks import AutoEncoderGroupSkip
    from torch import optim
    torch.set_grad_enabled(True)
    cfg = dict(geo_feat_channels=4, tex_feat_channels=8, feat_channel_up=16, mlp_hidden_channels=32, mlp_hidden_layers=4)
    with contextlib.redirect_stdout(io.StringIO()):
        net = AutoEncoderGroupSkip(*cfg.values())
    sd = T.synthetic_state_dict(T.ae_param_shapes(**cfg, with_encoder=True), 6)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and missing == ["aabb"], (missing, unexpected)
    aabb = torch.tensor([-0.6, -0.9, -0.5, 0.6, 0.9, 0.5])
    net.reset_aabb(aabb)
    lr, split, decay = 5e-3, 0.2, 0.999
    opt = optim.AdamW([{"params": net.geo_parameters(), "lr": lr * split}, {"params": net.tex_parameters(), "lr": lr}], lr)
    sched = optim.lr_scheduler.ExponentialLR(opt, decay)
    H, W, D, N = 8, 12, 6, 64
    vol = torch.tanh(rnd((1, 4, 2 * H, 2 * W, 2 * D), 1400))
    pts = rnd((N, 3), 1401) * aabb[3:]
    net.train()
    net(vol, pts).square().mean().backward()
    opt.step(); sched.step()
    net.eval()
    torch.set_grad_enabled(False)
    dst = os.path.join(HERE, "formats")
    os.makedirs(dst, exist_ok=True)
    save_dict = {"net": net.cpu().state_dict(), "optimizer": opt.state_dict(), "scheduler": sched.state_dict(),
                 "Ka": [1.0, 1.0, 1.0], "Kd": [0.8, 0.8, 0.8], "Ks": [0.5, 0.5, 0.5], "Ns": 250.0,
                 "aabb": aabb.tolist(), "featmap_size": (H, W, D)}
    torch.save(save_dict, os.path.join(dst, "ckpt_final.pth"))
    fm = [torch.tanh(rnd(s, 1410 + i)) for i, s in enumerate(((1, 12, H, W), (1, 12, H, D), (1, 12, W, D)))]
    q = rnd((96, 3), 1420) * aabb[3:] * 1.1                      # some points outside the box (border clamp)
    pred = net.decode(q, fm)
    save("formats/ckpt_decode", cfg=np.asarray(list(cfg.values())), xy=fm[0], xz=fm[1], yz=fm[2], pts=q, pred=pred,
         first_param=net.state_dict()["geo_convs.in_layers.0.weight"].flatten()[:16])
    print(f"formats/ckpt_final.pth: {os.path.getsize(os.path.join(dst, 'ckpt_final.pth')) / 1024:.1f} KiB")


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for name, fn in (("schedules", gen_schedules), ("temb", gen_temb), ("leaves", gen_leaves),
                     ("resblock", gen_resblock), ("unet", gen_unet), ("sampler", gen_sampler), ("sampler_branches", gen_sampler_branches),
                     ("decoder", gen_decoder), ("compose", gen_compose), ("train", gen_train), ("ae_train", gen_ae_train), ("respaced", gen_respaced_train),
                     ("formats", gen_formats), ("ae_ckpt", gen_ae_ckpt)):
        if not only or name in only:
            fn()
