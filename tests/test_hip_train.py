"""Training tier on the MI355X: loss terms, every parameter gradient and three optimizer steps
against digests captured from the reference's autograd / torch.optim.AdamW (tests/golden/train_*.npz)."""
import numpy as np
import pytest

from conftest import golden, relerr, digest_errors, zero_grad_params
from sin3dm_amd import testing as T

pytestmark = pytest.mark.gpu

TRAIN_CASES = [("mc32_a", 32, False, True, (1, 2), 2), ("mc32_odd", 32, False, True, (1, 2), 2),
               ("mc32_raw", 32, True, True, (1, 2), 2), ("mc32_add", 32, False, False, (1, 2), 1),
               ("mc32_3lev", 32, False, True, (1, 2, 2), 1)]


def _model(mc, raw=False, ssn=True, cm=(1, 2), seed=0):
    import torch
    from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall, TriplaneUNetModelSmallRaw
    cls = TriplaneUNetModelSmallRaw if raw else TriplaneUNetModelSmall
    m = cls(12, mc, 12, channel_mult=cm, use_scale_shift_norm=ssn)
    m.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=mc, rollout=not raw,
                                                                 use_scale_shift_norm=ssn, channel_mult=cm), seed))
    return m.to(torch.device("cuda:0"))


def _diffusion():
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    return create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=True)


def _inputs(g, tag, B, noise_seed=401):
    import torch
    H, W, D = (int(v) for v in g[f"{tag}.hwd"])
    dev = torch.device("cuda:0")
    x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1).to(dev)
    noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), noise_seed)).to(dev)
    return H, W, D, x0, noise


def _offenders(named, g, prefix, k=6):
    """The tensors with the largest norm / projection error, for the assertion message."""
    names = [str(n) for n in g[f"{prefix}/names"]]
    gn, gp = g[f"{prefix}/norm"], g[f"{prefix}/proj"]
    rows = []
    for i, n in enumerate(names):
        a = np.asarray(named[n], dtype=np.float64).reshape(-1)
        r = T.synthetic_tensor("digest/" + n, (a.size,), 7).astype(np.float64)
        ref = max(float(gn[i]), 1e-2 * float(gn.max()))
        rows.append((max(abs(np.linalg.norm(a) - gn[i]), abs(a @ r - gp[i])) / ref, n, float(np.linalg.norm(a)), float(gn[i])))
    return sorted(rows, reverse=True)[:k]


def test_device_repack_reproduces_host_packing():
    """Attaching the flat parameter vector rebuilds the whole kernel-layout image on the device; the inference
    forward must not change by a single bit."""
    import torch
    g = golden("unet_fwd")
    for tag, mc, cm in (("mc32_a", 32, (1, 2)), ("mc64_b", 64, (1, 2)), ("mc32_3lev", 32, (1, 2, 2))):
        m = _model(mc, cm=cm)
        H, W, D = (int(v) for v in g[f"{tag}.hwd"])
        x = torch.from_numpy(g[f"{tag}.x"]).cuda()
        t = torch.from_numpy(g[f"{tag}.t"]).cuda()
        with torch.no_grad():
            y0 = m(x, t, H=H, W=W, D=D).clone()
            _ = m.flat_parameters                 # attach + device repack
            y1 = m(x, t, H=H, W=W, D=D)
        assert torch.equal(y0, y1), tag
        assert relerr(y1.cpu().numpy(), g[f"{tag}.y"]) < 5e-6


@pytest.mark.parametrize("tag,mc,raw,ssn,cm,B", TRAIN_CASES)
def test_training_losses_and_grads(tag, mc, raw, ssn, cm, B):
    import torch
    g = golden("train_grads")
    m = _model(mc, raw, ssn, cm)
    diffusion = _diffusion()
    H, W, D, x0, noise = _inputs(g, tag, B)
    t = torch.from_numpy(g[f"{tag}.t"]).cuda()
    assert relerr(diffusion.q_sample_hip(x0, t, noise).cpu().numpy(), g[f"{tag}.x_t"]) < 1e-6
    terms = diffusion.training_losses(m, x0, t, model_kwargs=dict(H=H, W=W, D=D), noise=noise)
    for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
        assert terms[k].shape == (B,)
        assert relerr(terms[k].detach().cpu().numpy(), g[f"{tag}.{k}"]) < 2e-5, k
    (terms["loss"] * torch.ones(B, device="cuda")).mean().backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    assert all(np.isfinite(v).all() for v in grads.values())
    w = digest_errors(grads, g, f"{tag}.grad")
    assert w["norm"] < 2e-4 and w["proj"] < 2e-4 and w["head"] < 2e-3 and w["full"] < 2e-4, (w, _offenders(grads, g, f"{tag}.grad"))


@pytest.mark.parametrize("tag,ssn,B", [("mc32_a", True, 2), ("mc32_add", False, 1)])
def test_training_losses_and_grads_epsilon_target(tag, ssn, B):
    """predict_xstart=False (ModelMeanType.EPSILON, script_util.py:29,47): the MSE target is the noise
    (gaussian_diffusion.py:829-835); loss terms and every parameter gradient against the reference's autograd."""
    import torch
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    g = golden("train_eps")
    m = _model(32, ssn=ssn)
    diffusion = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_xstart=False)
    H, W, D, x0, noise = _inputs(g, tag, B)
    t = torch.from_numpy(g[f"{tag}.t"]).cuda()
    terms = diffusion.training_losses(m, x0, t, model_kwargs=dict(H=H, W=W, D=D), noise=noise)
    for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
        assert relerr(terms[k].detach().cpu().numpy(), g[f"{tag}.{k}"]) < 2e-5, k
    (terms["loss"] * torch.ones(B, device="cuda")).mean().backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    w = digest_errors(grads, g, f"{tag}.grad")
    assert w["norm"] < 2e-4 and w["proj"] < 2e-4 and w["head"] < 2e-3 and w["full"] < 2e-4, (w, _offenders(grads, g, f"{tag}.grad"))
    # the graph-free path (TrainLoop) takes the same target
    m2 = _model(32, ssn=ssn)
    terms2, flat = diffusion.training_losses_and_grads(m2, x0, t, torch.ones(B, device="cuda"), dict(H=H, W=W, D=D), noise=noise)
    assert relerr(terms2["loss"].cpu().numpy(), g[f"{tag}.loss"]) < 2e-5
    g2 = {k: v.cpu().numpy() for k, v in m2.split_flat(flat).items()}
    w2 = digest_errors(g2, g, f"{tag}.grad")
    assert w2["norm"] < 2e-4 and w2["proj"] < 2e-4, w2


def test_grads_with_the_direct_weight_gradient_kernel():
    """S3D_WGRAD_WINO=0 (the direct 3x3 weight gradient, default for 1x1 / 5x5) against the same golden gradients; the
    switch is read once per process, hence the subprocess."""
    import os, subprocess, sys
    code = ("import sys; sys.path.insert(0, 'tests')\n"
            "import test_hip_train as tt\n"
            "tt.test_training_losses_and_grads(*tt.TRAIN_CASES[0]); tt.test_training_losses_and_grads(*tt.TRAIN_CASES[4])\n"
            "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, S3D_WGRAD_WINO="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_training_on_the_f2x2_winograd_kernels():
    """S3D_WINO=4: forward and dgrad on k_conv_wino4 and a repack plan that keeps the F(2x2) images (and their transposes)
    current instead of the mixed kernel's — the branch the default process never takes.  Gradients against the golden
    digests, then optimizer steps (each one repacks) against the golden parameter norms."""
    import os, subprocess, sys
    code = ("import sys; sys.path.insert(0, 'tests')\n"
            "import test_hip_train as tt\n"
            "tt.test_training_losses_and_grads(*tt.TRAIN_CASES[0]); tt.test_training_losses_and_grads(*tt.TRAIN_CASES[4])\n"
            "tt.test_device_repack_reproduces_host_packing(); tt.test_optimizer_steps('wd01')\n"
            "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, S3D_WINO="4"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_fast_path_equals_autograd_and_is_repeatable():
    """training_losses_and_grads (no autograd graph) gives the same flat gradient as loss.backward(), bit for bit,
    and twice the same bits (all reductions have a fixed order)."""
    import torch
    g = golden("train_grads")
    tag, mc, B = "mc32_a", 32, 2
    m = _model(mc)
    diffusion = _diffusion()
    H, W, D, x0, noise = _inputs(g, tag, B)
    t = torch.from_numpy(g[f"{tag}.t"]).cuda()
    w = torch.tensor([1.0, 0.5], device="cuda")
    kw = dict(H=H, W=W, D=D)
    terms, g1 = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
    g1 = g1.clone()
    _, g2 = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
    assert torch.equal(g1, g2)
    t2 = diffusion.training_losses(m, x0, t, model_kwargs=kw, noise=noise)
    (t2["loss"] * w).mean().backward()
    assert torch.equal(terms["loss"], t2["loss"].detach())
    for name, view in m.split_flat(g1).items():
        assert torch.equal(view, dict(m.named_parameters())[name].grad), name
    # gradient accumulation keeps its meaning: a second backward adds
    t3 = diffusion.training_losses(m, x0, t, model_kwargs=kw, noise=noise)
    (t3["loss"] * w).mean().backward()
    p = dict(m.named_parameters())["input_blocks.0.0.in_layers.2.conv_xy.weight"]
    assert torch.allclose(p.grad, 2 * m.split_flat(g1)["input_blocks.0.0.in_layers.2.conv_xy.weight"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("wd_tag", ["wd0", "wd01"])
def test_optimizer_steps(wd_tag):
    """Three TrainLoop.run_step iterations (AdamW -> EMA -> linear anneal) with the fused flat optimizer."""
    import torch
    from sin3dm_amd.diffusion.train_util import FlatAdamW
    g = golden("train_steps")
    lr0, ema_rate, wd, anneal = (float(v) for v in g[f"{wd_tag}.hyper"])
    mc, B, (H, W, D) = 32, 2, (10, 14, 6)
    m = _model(mc)
    diffusion = _diffusion()
    init = {k: v.detach().clone() for k, v in m.named_parameters()}
    opt = FlatAdamW(m, lr=lr0, weight_decay=wd, ema_rates=[ema_rate])
    x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1).cuda()
    for step in range(3):
        noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 500 + step)).cuda()
        t = torch.tensor([[700, 3], [12, 999], [450, 451]][step], device="cuda")
        terms, grads = diffusion.training_losses_and_grads(m, x0, t, torch.ones(B, device="cuda"), dict(H=H, W=W, D=D), noise=noise)
        assert relerr(terms["loss"].cpu().numpy(), g[f"{wd_tag}.losses"][step]) < 2e-4, step
        opt.step(grads)
        opt.lr = lr0 * (1 - step / anneal)
    skip = zero_grad_params()
    dparam = {k: (p.detach() - init[k]).cpu().numpy() for k, p in m.named_parameters()}
    dema = {k: (v - init[k]).cpu().numpy() for k, v in m.split_flat(opt.ema[0]).items()}
    wp = digest_errors(dparam, g, f"{wd_tag}.dparam", skip)
    we = digest_errors(dema, g, f"{wd_tag}.dema", skip)
    # Adam's update is ~ lr * g/|g| per element: where a gradient element is within round-off of zero, 1e-4-level
    # gradient differences (fp32, different summation order) flip a full +-lr step.  Those few elements bound the L2
    # agreement of the parameter deltas at the percent level; the norms agree to 1e-3.
    assert wp["norm"] < 5e-3 and wp["proj"] < 3e-2 and wp["full_l2"] < 3e-2, wp
    assert we["norm"] < 5e-3 and we["proj"] < 3e-2, we


def test_torch_optimizer_also_works():
    """Drop-in use: a stock torch.optim.AdamW on model.parameters() (views of the flat vector) trains the HIP model."""
    import torch
    g = golden("train_grads")
    tag, B = "mc32_a", 2
    m = _model(32)
    diffusion = _diffusion()
    H, W, D, x0, noise = _inputs(g, tag, B)
    t = torch.from_numpy(g[f"{tag}.t"]).cuda()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.0)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        terms = diffusion.training_losses(m, x0, t, model_kwargs=dict(H=H, W=W, D=D), noise=noise)
        loss = terms["loss"].mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.8 * losses[0], losses          # same batch every step: the loss must fall
    with torch.no_grad():                                 # and sampling sees the updated weights
        y = m(x0, t, H=H, W=W, D=D)
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("mc,hwd,B", [(64, (12, 8, 10), 2), (64, (46, 64, 46), 1), (128, (20, 24, 12), 1),
                                      (64, (92, 128, 92), 1)])         # the last one is BASELINE config 4's towerruins size
def test_grads_vs_oracle_wider(oracle, mc, hwd, B):
    """Wider models (2 and 4 channels per GroupNorm group, 64-wide MFMA tiles with several K slices) against the
    CPU oracle's autograd (oracle/torch_port.py, itself pinned to the reference's gradients).  The towerruins-size case runs at
    batch 1 — BASELINE config 4's per-GPU batch is 4 — because the CPU oracle's autograd at that size costs ~20 s per sample;
    every kernel of the step is batch-parallel over independent samples (GroupNorm statistics and rollout means are per sample),
    tools/bench_train.py times the real batch of 4, test_full_size_directional_derivative runs batch 2 and
    test_full_size_batch_of_four_is_the_mean_of_its_samples ties the batch-4 step to its four batch-1 steps."""
    import torch
    import torch_port as tp
    H, W, D = hwd
    shapes = T.unet_param_shapes(model_channels=mc)
    m = _model(mc)
    diffusion = _diffusion()
    x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1)
    noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 401))
    t = torch.tensor([700, 3][:B])
    sd = {k: v.requires_grad_(True) for k, v in T.synthetic_state_dict(shapes, 0).items()}
    terms_ref, _ = tp.training_losses(sd, x0, t, noise, oracle.schedule_tables_named(1000), H, W, D, model_channels=mc)
    terms_ref["loss"].mean().backward()
    terms, g = diffusion.training_losses_and_grads(m, x0.cuda(), t.cuda(), torch.ones(B, device="cuda"), dict(H=H, W=W, D=D),
                                                   noise=noise.cuda())
    assert relerr(terms["loss"].cpu().numpy(), terms_ref["loss"].detach().numpy()) < 2e-5
    gmax = max(float(v.grad.norm()) for v in sd.values())
    worst = []
    for name, view in m.split_flat(g).items():
        ref = sd[name].grad
        err = float((view.cpu() - ref).norm()) / max(float(ref.norm()), 1e-2 * gmax)
        worst.append((err, name))
    worst.sort(reverse=True)
    assert worst[0][0] < 5e-4, worst[:5]


def test_full_size_directional_derivative():
    """BASELINE config 4 per GPU (64-ch UNet, towerruins (92,128,92); batch 2 here where the config's per-GPU batch is 4 — three
    loss evaluations + one backward at this size per test; the step is batch-parallel): the gradient of the whole step agrees
    with a central finite difference of the loss along a random parameter direction — a size-independent check that
    needs no reference output."""
    import torch
    mc, (H, W, D), B = 64, (92, 128, 92), 2
    m = _model(mc)
    diffusion = _diffusion()
    g = torch.Generator(device="cuda").manual_seed(11)
    x0 = torch.rand((B, 12, H + D, W + D), device="cuda", generator=g) * 2 - 1
    noise = torch.randn((B, 12, H + D, W + D), device="cuda", generator=g)
    t = torch.tensor([650, 40], device="cuda")
    w = torch.ones(B, device="cuda")
    kw = dict(H=H, W=W, D=D)
    flat = m.flat_parameters

    def loss_and_grad():
        terms, grad = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
        return float(terms["loss"].double().mean()), grad

    L0, grad = loss_and_grad()
    grad = grad.clone()
    r = torch.randn(flat.shape, device="cuda", generator=g)
    v = grad / grad.norm() + r / r.norm()                      # half along the gradient, half random
    gv = float((grad.double() * v.double()).sum())
    eps = 2e-3 * L0 / abs(gv)                                  # moves the loss by ~0.2 %
    base = flat.clone()
    vals = []
    for s in (+1.0, -1.0):
        flat.copy_(base + s * eps * v)
        m.mark_parameters_changed()
        vals.append(loss_and_grad()[0])
    flat.copy_(base); m.mark_parameters_changed()
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - gv) < 2e-2 * abs(gv), (fd, gv, L0, eps)


def test_full_size_batch_of_four_is_the_mean_of_its_samples():
    """BASELINE config 4 per GPU at its own batch: 64-ch UNet, towerruins (92,128,92), batch 4.  The step's loss terms are per
    sample and its gradient is the mean over the batch, so the batch-4 gradient must equal the mean of the four batch-1 gradients
    (same x0, noise, t per sample) up to the order of fp32 sums — a size-independent check of the batched launches (batch offsets
    of every kernel, per-sample GroupNorm statistics and rollout means, split-K weight-gradient sums over B x tiles)."""
    import torch
    mc, (H, W, D), B = 64, (92, 128, 92), 4
    m = _model(mc)
    diffusion = _diffusion()
    g = torch.Generator(device="cuda").manual_seed(23)
    x0 = torch.rand((B, 12, H + D, W + D), device="cuda", generator=g) * 2 - 1
    noise = torch.randn((B, 12, H + D, W + D), device="cuda", generator=g)
    t = torch.tensor([650, 40, 999, 0], device="cuda")
    kw = dict(H=H, W=W, D=D)
    terms4, g4 = diffusion.training_losses_and_grads(m, x0, t, torch.ones(B, device="cuda"), kw, noise=noise)
    g4, loss4 = g4.clone(), terms4["loss"].clone()
    acc = torch.zeros_like(g4, dtype=torch.float64)
    for b in range(B):
        terms1, g1 = diffusion.training_losses_and_grads(m, x0[b:b + 1], t[b:b + 1], torch.ones(1, device="cuda"), kw, noise=noise[b:b + 1])
        assert abs(float(terms1["loss"][0]) - float(loss4[b])) <= 2e-6 * abs(float(loss4[b])), b
        acc += g1.double()
    mean = (acc / B).float()
    named4, named1 = m.split_flat(g4), m.split_flat(mean)
    gmax = max(float(v.norm()) for v in named1.values())
    worst = sorted(((float((named4[k] - named1[k]).norm()) / max(float(named1[k].norm()), 1e-2 * gmax), k) for k in named1), reverse=True)
    assert worst[0][0] < 2e-5, worst[:5]


@pytest.mark.parametrize("dtag", ["r50_rescale", "s200_rescale", "r50"])
def test_respaced_training_conditions_on_original_timesteps(dtag):
    """SpacedDiffusion.training_losses (reference respace.py:93-96): with --timestep_respacing and/or --rescale_timesteps
    the denoiser is conditioned on timestep_map[t] (* 1000 / original steps) during TRAINING as well as sampling — loss
    terms and every gradient against the reference's autograd, on both the autograd and the graph-free path."""
    import torch
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    g = golden("train_respaced")
    steps, rescale, n_t = (int(v) for v in g[f"{dtag}.cfg"])
    resp = "50" if dtag.startswith("r50") else ""
    diffusion = create_gaussian_diffusion(steps=steps, noise_schedule="linear", predict_xstart=True, timestep_respacing=resp,
                                          rescale_timesteps=bool(rescale))
    assert diffusion.num_timesteps == n_t
    m = _model(32)
    H, W, D = (int(v) for v in g["hwd"])
    B = 2
    dev = torch.device("cuda:0")
    x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1).to(dev)
    noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 401)).to(dev)
    t = torch.from_numpy(g[f"{dtag}.t"]).cuda()
    kw = dict(H=H, W=W, D=D)
    terms = diffusion.training_losses(m, x0, t, model_kwargs=kw, noise=noise)
    for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
        assert relerr(terms[k].detach().cpu().numpy(), g[f"{dtag}.{k}"]) < 2e-5, k
    (terms["loss"] * torch.ones(B, device="cuda")).mean().backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    w = digest_errors(grads, g, f"{dtag}.grad")
    assert w["norm"] < 2e-4 and w["proj"] < 2e-4 and w["head"] < 2e-3, w
    terms2, flat = diffusion.training_losses_and_grads(m, x0, t, torch.ones(B, device="cuda"), kw, noise=noise)
    assert torch.equal(terms2["loss"], terms["loss"].detach())
    for name, view in m.split_flat(flat).items():
        assert torch.equal(view, dict(m.named_parameters())[name].grad), name


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adamw_ema_kernel_elementwise(wd):
    """The fused AdamW + EMA kernel element by element against torch.optim.AdamW (what the reference constructs,
    train_util.py:84) + update_ema (nn.py:55-65) in float64, on gradients bounded away from zero (|g| in [0.1, 2]) so that
    Adam's g / (sqrt(v) + eps) is well conditioned: pins the bias-correction exponents, the decoupled weight decay, the
    order lr / step and the EMA lerp, none of which the norm-level check of test_optimizer_steps can see."""
    import ctypes as C
    import torch
    from sin3dm_amd import _lib
    n, steps, lr, b1, b2, eps, rate = 4099, 4, 3e-3, 0.9, 0.999, 1e-8, 0.97
    gen = np.random.Generator(np.random.PCG64(5))
    p0 = gen.normal(0, 1, n).astype(np.float32)
    grads = [(gen.uniform(0.1, 2.0, n) * gen.choice([-1.0, 1.0], n)).astype(np.float32) for _ in range(steps)]
    ref = torch.nn.Parameter(torch.from_numpy(p0).double())
    opt = torch.optim.AdamW([ref], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    ema_ref = torch.from_numpy(p0).double()
    p = torch.from_numpy(p0).cuda()
    m, v, ema = torch.zeros_like(p), torch.zeros_like(p), p.clone()
    lib = _lib.load()
    for s in range(steps):
        lr_s = lr * (1 - s / 10)                                      # the linear anneal changes lr between steps
        for gr in opt.param_groups:
            gr["lr"] = lr_s
        ref.grad = torch.from_numpy(grads[s]).double()
        opt.step()
        ema_ref.mul_(rate).add_(ref.detach(), alpha=1 - rate)
        g = torch.from_numpy(grads[s]).cuda()
        _lib.check(lib.s3d_train_adamw_ema(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), (C.c_void_p * 1)(ema.data_ptr()),
                                           (C.c_float * 1)(rate), 1, n, lr_s, b1, b2, eps, wd, s + 1, _lib.stream_ptr()))
        dp = (p.double().cpu() - torch.from_numpy(p0).double()).numpy()
        dref = (ref.detach() - torch.from_numpy(p0).double()).numpy()
        # p ~ N(0,1) is stored in fp32 (half an ulp at 4.0 = 2.4e-7 per step); a wrong bias-correction exponent or decay order
        # moves the update by >= 1e-4
        assert np.max(np.abs(dp - dref)) < 1.5e-6, (s, np.max(np.abs(dp - dref)))
        assert np.max(np.abs(ema.double().cpu().numpy() - ema_ref.numpy())) < 1.5e-6, s


def test_backward_progress_marks_and_overlapped_exchange_groups():
    """s3d_unet_backward_marked: the same gradient bits as the unmarked call; its two events fire in order before the pass ends,
    and at each of them the ranges grad_ready_groups() assigns to it already hold their FINAL values — nothing writes them
    afterwards.  A second stream waits for each mark, copies the group's ranges out and then POISONS them in place (what the
    data-parallel trainer's communication stream does at that moment is an in-place all-reduce): the copies must equal the plain
    gradients, the poison must still be there at the end (a kernel that touched a range after its mark would have overwritten
    it), and the poisoning of the first group must have finished before the backward pass did (so the check really ran inside the pass)."""
    import torch
    tag, mc, B = "big", 64, 4
    m = _model(mc)
    diffusion = _diffusion()
    # a GPU-bound pass: issuing a step's ~190 launches costs the host ~2 ms whatever the size (tools/debug_streams_after.py), so the
    # planes must be large enough for the GPU to need several times that — only then has the host enqueued the whole pass, and
    # the side stream's copy-and-poison behind it, long before the first mark fires
    H, W, D = 128, 128, 128
    dev = torch.device("cuda:0")
    x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1).to(dev)
    noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 401)).to(dev)
    t = torch.tensor([700, 3, 250, 10], device=dev)
    w = torch.tensor([1.0, 0.5, 2.0, 1.5], device="cuda")
    kw = dict(H=H, W=W, D=D)
    _, g_plain = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
    g_plain = g_plain.clone()
    groups = m.grad_ready_groups()
    marks = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    end = torch.cuda.Event(enable_timing=True)
    poisoned = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    out = torch.full_like(g_plain, float("nan"))
    side = _concurrent_stream(dev)
    POISON = 12345.678
    idx = [torch.cat([torch.arange(b, e, device=dev) for b, e in groups[k]]) for k in range(2)]   # one gather / one fill per group
    snaps = [torch.empty(len(idx[k]), device=dev) for k in range(2)]
    with torch.cuda.stream(side):                        # warm-up: the gather / fill kernels are loaded and their buffers exist before the
        scratch = torch.zeros_like(out)                  # measured pass (a first-use code-object load or allocation would stall the side stream
        for k in range(2):                               # for longer than the whole backward pass)
            torch.index_select(scratch, 0, idx[k], out=snaps[k])
            scratch.index_fill_(0, idx[k], POISON)
    torch.cuda.synchronize()
    # Twice.  In line (BWD_SIDE = 0) the marks sit at fixed places of ONE stream's program order, and the copy-and-poison of group 0
    # must land INSIDE the pass (a race between two index kernels and the second half of the pass: a few attempts, one must land) —
    # that is what makes "the poison is still there at the end" a statement about the kernels after the mark.  By default the weight
    # gradients run on a low-priority side stream and a mark is recorded THERE, behind the caller's progress: the same kernels write
    # the same ranges in the same per-stream order, so the in-line result carries over; when the mark fires is up to the side
    # stream's backlog (often close to the end of the pass), so only the data checks are asserted in that mode.
    from sin3dm_amd import _lib
    try:
        for mode in (0, None):
            _lib.set_option("BWD_SIDE", mode)
            inside = []
            for attempt in range(4):
                out.fill_(float("nan"))
                torch.cuda.synchronize()
                _, g_marked = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise, grad_out=out, grad_marks=marks)
                end.record()
                for k in range(2):                       # (the host is far ahead of the GPU: these run when the mark fires)
                    side.wait_event(marks[k])
                    with torch.cuda.stream(side):
                        torch.index_select(out, 0, idx[k], out=snaps[k])
                        out.index_fill_(0, idx[k], POISON)
                        poisoned[k].record(side)
                torch.cuda.synchronize()
                assert g_marked.data_ptr() == out.data_ptr()
                assert marks[0].elapsed_time(marks[1]) > 0 and marks[1].elapsed_time(end) > 0
                for k in range(2):
                    assert torch.equal(snaps[k], g_plain.index_select(0, idx[k])), f"group {k} was not final at its mark (BWD_SIDE={mode})"
                    assert bool((out.index_select(0, idx[k]) == POISON).all()), f"a kernel wrote a gradient range of group {k} after its mark (BWD_SIDE={mode})"
                inside.append(poisoned[0].elapsed_time(end) > 0)   # (after mark 1 only the ~10 small timestep-linear launches remain: no such bound there)
                if inside[-1] or mode is None:
                    break
            assert mode is None or any(inside), "in line: group 0 was only copied and poisoned after the backward pass had ended, in every attempt"
    finally:
        _lib.set_option("BWD_SIDE", None)
    for b, e in groups[2]:                               # the late group (timestep linears) is final at the end of the pass
        assert torch.equal(out[b:e], g_plain[b:e])


def _concurrent_stream(dev, tries=12):
    """A torch stream whose work really runs BESIDE the current stream's.  HIP maps streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default); a process that has created many streams — the chain tests before this one do — gets new
    ones that share a queue with an old one, possibly with the current stream's: work on such a stream starts when the current
    stream's queued work has drained, however independent it is.  Probe: a spin on the current stream, a tiny kernel on the
    candidate; keep the first candidate whose kernel finishes while the spin is still running."""
    import torch
    main = torch.cuda.current_stream(dev)
    keep = []
    for _ in range(tries):
        cand = torch.cuda.Stream(device=dev)
        keep.append(cand)                                # (held, so that the next candidate is another pool stream)
        e0, e_main, e_c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        x = torch.zeros(1024, device=dev)
        torch.cuda.synchronize()
        e0.record(main)
        torch.cuda._sleep(int(4e6))                      # ~2 ms of spinning on the current stream
        e_main.record(main)
        with torch.cuda.stream(cand):
            x.add_(1)
            e_c.record(cand)
        torch.cuda.synchronize()
        if e0.elapsed_time(e_c) < 0.5 * e0.elapsed_time(e_main):
            return cand
    pytest.skip("no torch stream of this process runs beside the current stream (hardware queues shared)")


def _flat_grad_digest(expand):
    """sha256 of the flat gradient + loss terms of one graph-free training step at a size whose backward pass is GPU-bound."""
    import hashlib
    import torch
    m = _model(64)
    diffusion = _diffusion()
    H, W, D, B = 48, 64, 40, 3
    dev = torch.device("cuda:0")
    x1 = torch.from_numpy(T.synthetic_noise((12, H + D, W + D), 410)).clamp(-1, 1).to(dev)
    x0 = x1.unsqueeze(0).expand(B, -1, -1, -1) if expand else x1.unsqueeze(0).repeat(B, 1, 1, 1)
    noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 411)).to(dev)
    t = torch.tensor([700, 3, 250], device=dev)
    w = torch.tensor([1.0, 0.5, 2.0], device=dev)
    h = hashlib.sha256()
    for _ in range(3):                                   # (the side stream's hand-offs must hold when passes follow each other)
        terms, g = diffusion.training_losses_and_grads(m, x0, t, w, dict(H=H, W=W, D=D), noise=noise)
        torch.cuda.synchronize()
        h.update(g.cpu().numpy().tobytes())
        for k in ("mse_xy", "mse_xz", "mse_yz", "loss"):
            h.update(terms[k].cpu().numpy().tobytes())
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    assert torch.equal(terms["loss"], (terms["mse_xy"] + terms["mse_xz"]) + terms["mse_yz"])
    return h.hexdigest()


def test_weight_gradients_on_the_side_stream_change_no_bit():
    """The backward pass enqueues every weight-gradient launch (k_wgrad_wino / k_wgrad_mfma / k_slot_wgrad and their split-K
    reductions, the bias sums, in_conv's outer products) on a handle-owned side stream; S3D_BWD_SIDE=0 keeps them in line.  Same
    kernels on the same operands: identical gradients, pass after pass.  Also: the single training triplane expanded to a batch
    (batch stride 0, utils/triplane_util.py:64-69) is read in place and gives the bits of the materialised batch."""
    import os, subprocess, sys
    here = _flat_grad_digest(expand=True)
    assert here == _flat_grad_digest(expand=False)
    code = ("import sys; sys.path.insert(0, 'tests')\n"
            "import test_hip_train as tt\n"
            "print('DIGEST', tt._flat_grad_digest(expand=True))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("0",):                                  # everything in line
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, S3D_BWD_SIDE=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        other = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]
        assert other == here, mode


def test_a_form_change_after_attach_fails_instead_of_using_stale_weights():
    """The per-step device repack of a training handle rewrites only the weight images of the kernel forms selected when the plan
    was built (with the mixed Winograd kernel: not the dense / F(2x2) images of the 3x3 layers).  Selecting another form
    afterwards would read load-time weights; the library refuses the launch — and works again once the option is
    cleared."""
    import torch
    from sin3dm_amd import _lib
    m = _model(32)
    diffusion = _diffusion()
    H, W, D, B = 10, 14, 6, 2
    x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 400)).clamp(-1, 1).cuda()
    noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 401)).cuda()
    t = torch.tensor([700, 3], device="cuda")
    w = torch.ones(B, device="cuda")
    kw = dict(H=H, W=W, D=D)
    _, g0 = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
    g0 = g0.clone()
    try:
        _lib.set_option("WINO", 4)
        with pytest.raises(AssertionError, match="repack plan"):
            diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
        with torch.no_grad(), pytest.raises(AssertionError, match="repack plan"):
            m(x0, t.float(), **kw)
    finally:
        _lib.set_option("WINO", None)
    _, g1 = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
    assert torch.equal(g0, g1)


def test_groupnorm_backward_sums_from_the_dgrad_epilogue_agree_with_the_read_pass():
    """GroupNorm(+FiLM)+SiLU backward (forward: src/diffusion/unet_triplane.py:63-95, 269-311) needs sum(dz) and sum(dz * xh) per
    channel before it can form dx.  By default the input-gradient convolution in front of it (k_conv_wino24s_gnb) leaves them as
    per-tile records from its epilogue, where dy already sits in registers; GNB_FUSED = 0 takes them from the read pass over (x, dy)
    (k_gn_bwd_partials).  Same mathematics, another summation order: every parameter gradient agrees to round-off, both forms are
    repeatable, the fused form is what the default step runs (the library names the kernel), and the losses are the same bits
    (the forward is untouched)."""
    import torch
    from sin3dm_amd import _lib
    diffusion = _diffusion()
    dev = torch.device("cuda:0")
    for mc, (H, W, D), B, ssn in ((64, (48, 64, 40), 3, True), (32, (9, 13, 7), 2, True), (32, (12, 16, 10), 2, False)):
        x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 420)).clamp(-1, 1).to(dev)
        noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 421)).to(dev)
        t = torch.tensor([700, 3, 250][:B], device=dev)
        w = torch.tensor([1.0, 0.5, 2.0][:B], device=dev)
        kw = dict(H=H, W=W, D=D)
        out = {}
        try:
            for mode in (None, 0):
                _lib.set_option("GNB_FUSED", mode)
                m = _model(mc, ssn=ssn)
                m.profile(1, classes=1)
                terms, g = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
                g = g.clone()
                _, g2 = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
                assert torch.equal(g, g2), mode
                m.profile_read()
                out[mode] = (terms["loss"].clone(), g, m.profile_kernel(0), m.split_flat(g))
        finally:
            _lib.set_option("GNB_FUSED", None)
        assert "k_conv_wino24s_gnb" in out[None][2] and "gnb" not in out[0][2], (out[None][2], out[0][2])
        assert torch.equal(out[None][0], out[0][0])
        for name, a in out[None][3].items():
            b = out[0][3][name]
            scale = max(float(b.norm()), 1e-3 * float(out[0][1].norm()))      # (floor: a conv bias in front of a GroupNorm has a zero gradient, round-off noise in both forms)
            # (the gate of the golden-gradient tests; the FiLM / affine gradients ARE these sums, each a cancelling sum of thousands of terms)
            assert float((a - b).norm()) <= 2e-4 * scale, (mc, name, float((a - b).norm()), scale)
        assert not torch.equal(out[None][1], out[0][1]) or mc == 0          # (another summation order: not the same bits)


def test_lds_dma_conv_form_gives_the_same_gradient_bits():
    """S3D_WINO24G = 1: every mixed-Winograd launch of the training step — forward, input gradient, and the input gradient with the
    GroupNorm-backward epilogue (k_conv_wino24g_gnb) — takes the LDS-DMA / persistent form (TriplaneConv forward and transpose,
    src/diffusion/unet_triplane.py:27-58).  Same products and sums in the same order: losses and the whole flat gradient are the
    same bits as with the default kernels, and the library names the kernel that ran."""
    import torch
    from sin3dm_amd import _lib
    diffusion = _diffusion()
    dev = torch.device("cuda:0")
    for mc, (H, W, D), B, ssn in ((64, (48, 64, 40), 3, True), (32, (9, 13, 7), 2, False)):
        x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 430)).clamp(-1, 1).to(dev)
        noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 431)).to(dev)
        t = torch.tensor([700, 3, 250][:B], device=dev)
        w = torch.tensor([1.0, 0.5, 2.0][:B], device=dev)
        kw = dict(H=H, W=W, D=D)
        out = {}
        try:
            for mode in (None, 1):
                _lib.set_option("WINO24G", mode)
                m = _model(mc, ssn=ssn)
                m.profile(1, classes=1)
                terms, g = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
                g = g.clone()
                _, g2 = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
                assert torch.equal(g, g2), mode
                m.profile_read()
                out[mode] = (terms["loss"].clone(), g, m.profile_kernel(0))
        finally:
            _lib.set_option("WINO24G", None)
        assert "k_conv_wino24g" in out[1][2] and "k_conv_wino24g" not in out[None][2], (out[None][2], out[1][2])
        assert torch.equal(out[None][0], out[1][0]) and torch.equal(out[None][1], out[1][1]), mc


def test_step_inputs_drawn_ahead_give_the_same_training_run():
    """TrainLoop draws step k + 1's timesteps, importance weights and noise while step k's backward pass runs (one pinned staging
    row + one asynchronous copy, the timestep_map lookup done on the host; src/diffusion/train_util.py:198-232,
    resample.py:33-48, respace.py:93-96).  Each generator is still consumed once per step in step order: four run_steps end with
    the same parameter and EMA bits as with every input drawn at the start of its own step (S3D_PREFETCH_INPUTS=0), for the plain
    and for a respaced + rescaled diffusion (whose model timesteps are map[t] * 1000 / T)."""
    import os
    import torch
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    from sin3dm_amd.diffusion.train_util import TrainLoop
    dev = torch.device("cuda:0")
    H, W, D = 20, 28, 12
    x0 = torch.from_numpy(T.synthetic_noise((12, H + D, W + D), 400)).clamp(-1, 1).to(dev)
    batch, cond = x0.unsqueeze(0).expand(3, -1, -1, -1), dict(H=H, W=W, D=D)

    def data():
        while True:
            yield batch, cond

    for dkw in (dict(), dict(timestep_respacing="250", rescale_timesteps=True)):
        ends = {}
        for mode in ("0", "1"):
            os.environ["S3D_PREFETCH_INPUTS"] = mode
            try:
                torch.manual_seed(77); np.random.seed(77)
                m = _model(32)
                loop = TrainLoop(model=m, diffusion=create_gaussian_diffusion(steps=1000, predict_xstart=True, **dkw), data=data(),
                                 batch_size=3, microbatch=-1, lr=1e-3, ema_rate="0.99", log_interval=10 ** 9, save_interval=10 ** 9,
                                 resume_checkpoint=False, lr_anneal_steps=10, log_dir=None)
                assert loop.prefetch_inputs == (mode == "1")
                for _ in range(4):
                    loop.run_step(batch, cond)
                    loop.step += 1
                torch.cuda.synchronize()
                assert (loop._next_inputs is not None) == (mode == "1")
                ends[mode] = (m.flat_parameters.clone(), loop.opt.ema[0].clone())
            finally:
                os.environ.pop("S3D_PREFETCH_INPUTS", None)
        assert torch.isfinite(ends["0"][0]).all()
        assert torch.equal(ends["0"][0], ends["1"][0]) and torch.equal(ends["0"][1], ends["1"][1]), dkw


def test_weight_gradient_operands_by_lds_dma_give_the_same_gradient_bits():
    """The 3x3 weight gradient (the transpose of TriplaneConv's dense part in the own channels, src/diffusion/unet_triplane.py:27-58) on
    k_wgrad_wino_dma (WGRAD_WINO = 2: half regions double-buffered in LDS, filled by buffer_load ... lds while the previous half is
    multiplied; round 6, measured slower and not the default) against k_wgrad_wino (load -> registers -> ds_write, one region at a time): the same
    products added in the same order, so the whole flat gradient is the same bits; ragged planes and a 96-channel layer (three
    32-channel tiles) included."""
    import torch
    from sin3dm_amd import _lib
    diffusion = _diffusion()
    dev = torch.device("cuda:0")
    for mc, (H, W, D), B in ((64, (48, 64, 40), 3), (32, (9, 13, 7), 2), (32, (20, 28, 12), 2)):
        x0 = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 440)).clamp(-1, 1).to(dev)
        noise = torch.from_numpy(T.synthetic_noise((B, 12, H + D, W + D), 441)).to(dev)
        t = torch.tensor([700, 3, 250][:B], device=dev)
        w = torch.tensor([1.0, 0.5, 2.0][:B], device=dev)
        kw = dict(H=H, W=W, D=D)
        out = {}
        try:
            for mode in (None, 2):
                _lib.set_option("WGRAD_WINO", mode)
                m = _model(mc)
                _, g = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
                g = g.clone()
                _, g2 = diffusion.training_losses_and_grads(m, x0, t, w, kw, noise=noise)
                assert torch.equal(g, g2), mode
                out[mode] = g
        finally:
            _lib.set_option("WGRAD_WINO", None)
        assert torch.isfinite(out[None]).all() and torch.equal(out[None], out[2]), (mc, float((out[None] - out[2]).abs().max()))
