"""Auto-encoder training tier on the MI355X against vectors captured from the reference's
AutoEncoderGroupSkip under autograd (tests/golden/ae_train.npz)."""
import numpy as np
import pytest

from conftest import golden, relerr, digest_errors, zero_grad_params
from sin3dm_amd import testing as T

pytestmark = pytest.mark.gpu


def _net():
    import torch
    from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip
    net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
    sd = T.synthetic_state_dict(T.ae_param_shapes(with_encoder=True), 5)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and missing == ["aabb"]
    return net.to(torch.device("cuda:0"))


def _volume(H, W, D):
    import torch
    vol = torch.tanh(torch.from_numpy(T.synthetic_noise((1, 4, 2 * H, 2 * W, 2 * D), 1200)))
    vol[:, 1:] = 0.5 * vol[:, 1:] + 0.5
    return vol.cuda()


def _loss_cfg(thr):
    from sin3dm_amd import _lib
    return _lib.AeLossCfg(1, 0, thr, 0.999, 1.0)


def _batch(seed, N, thr):
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    ext = np.asarray([0.7, 1.0, 0.45], np.float32)
    p = torch.from_numpy(rng.uniform(-1.1, 1.1, size=(N, 3)).astype(np.float32) * ext)
    s = torch.from_numpy(np.clip(rng.normal(0, 0.04, size=(N, 1)), -thr, thr).astype(np.float32))
    c = torch.from_numpy(rng.uniform(0, 1, size=(N, 3)).astype(np.float32))
    return p.cuda(), s.cuda(), c.cuda()


def test_encode_forward_losses_and_grads():
    import torch
    g = golden("ae_train")
    H, W, D, N = (int(v) for v in g["hwdn"])
    net = _net()
    vol = _volume(H, W, D)
    aabb, thr = torch.from_numpy(g["aabb"]).cuda(), float(g["thr"])
    net.reset_aabb(aabb)
    fm = net.encode(vol)
    for f, k in zip(fm, ("xy", "xz", "yz")):
        assert relerr(f.cpu().numpy(), g[k]) < 2e-5, k
    pts, sdf, tex = (torch.from_numpy(g[k]).cuda() for k in ("pts", "sdf", "tex"))
    pred = net(vol, pts)
    assert relerr(pred.cpu().numpy(), g["pred"]) < 2e-5
    # the training-tier forward and the inference decoder (fused gather+MLP kernel) agree
    assert relerr(net.decode(pts, fm).cpu().numpy(), g["pred"]) < 2e-5
    losses, pred2, grads = net.loss_and_grads(vol, pts, sdf, tex, _loss_cfg(thr), want_pred=True)
    assert torch.equal(pred2, pred)
    assert abs(float(losses[0]) - float(g["sdf_loss"])) < 1e-5 and abs(float(losses[1]) - float(g["tex_loss"])) < 1e-5
    grads = grads.clone()
    _, _, again = net.loss_and_grads(vol, pts, sdf, tex, _loss_cfg(thr))
    assert torch.equal(grads, again)                               # every reduction has a fixed order (no float atomics)
    named = {k: v.cpu().numpy() for k, v in net.split_flat(grads).items()}
    assert all(np.isfinite(v).all() for v in named.values())
    w = digest_errors(named, g, "grad")
    assert w["norm"] < 2e-4 and w["proj"] < 5e-4 and w["full"] < 5e-4, w


def test_optimizer_steps():
    """three ShapeAutoEncoder iterations: two-group AdamW (weight_decay 0.01) + ExponentialLR on the flat vector.
    The golden batches were picked (make_golden.py:pick_batch) so that no hidden ReLU sits within 2e-6 of zero: a
    pre-activation at round-off distance from zero makes relu' — and with it the gradient — implementation-defined."""
    import torch
    from sin3dm_amd.encoding.model import FlatGroupAdamW
    g = golden("ae_train")
    H, W, D, N = (int(v) for v in g["hwdn"])
    lr, split, decay = (float(v) for v in g["steps.hyper"])
    net = _net()
    vol = _volume(H, W, D)
    thr = float(g["thr"])
    net.reset_aabb(torch.from_numpy(g["aabb"]).cuda())
    init = {k: v.detach().clone() for k, v in net.named_parameters()}
    opt = FlatGroupAdamW(net, lr, split, lr_decay=decay)
    for step in range(3):
        p, s, c = _batch(int(g["steps.seeds"][step]), N, thr)
        losses, _, grads = net.loss_and_grads(vol, p, s, c, _loss_cfg(thr))
        assert abs(float(losses[0]) - g["steps.losses"][step][0]) < 5e-4 * max(1.0, g["steps.losses"][step][0]), step
        opt.step(grads)
    skip = zero_grad_params(None, "ae_train")
    d = {k: (p.detach() - init[k]).cpu().numpy() for k, p in net.named_parameters()}
    w = digest_errors(d, g, "steps.dparam", skip)
    assert w["norm"] < 5e-3 and w["proj"] < 3e-2, w
    # the decode path sees the trained weights
    fm = net.encode(vol)
    p, _, _ = _batch(7, 64, thr)
    assert relerr(net.decode(p, fm).cpu().numpy(), net(vol, p).cpu().numpy()) < 2e-5


def test_train_loop_reduces_the_loss(tmp_path):
    """ShapeAutoEncoder.train on a synthetic .npz in the reference's data format; checkpoint loads back for decoding."""
    import torch
    from types import SimpleNamespace
    from sin3dm_amd.encoding.model import ShapeAutoEncoder
    rng = np.random.Generator(np.random.PCG64(3))
    R = (24, 32, 20)
    ax = [np.linspace(-1, 1, r) * s for r, s in zip(R, (0.7, 1.0, 0.45))]
    pts_grid = np.stack(np.meshgrid(*ax, indexing="ij"), -1).astype(np.float32)
    sdf = (np.linalg.norm(pts_grid / np.asarray([0.7, 1.0, 0.45]), axis=-1) - 0.6).astype(np.float32) * 0.3
    tex = (0.5 + 0.5 * np.sin(3 * pts_grid)).astype(np.float32)
    near = (rng.uniform(-1, 1, size=(4000, 3)) * np.asarray([0.7, 1.0, 0.45])).astype(np.float32)
    sdf_near = ((np.linalg.norm(near / np.asarray([0.7, 1.0, 0.45]), axis=-1) - 0.6) * 0.3).astype(np.float32)
    path = str(tmp_path / "shape.npz")
    np.savez(path, aabb=np.asarray([-0.7, -1.0, -0.45, 0.7, 1.0, 0.45], np.float32), threshold=0.05, pts_grid=pts_grid, sdf_grid=sdf,
             tex_grid=tex, pts_near_surf=near, sdf_near_surf=sdf_near, tex_near_surf=(0.5 + 0.5 * np.sin(3 * near)).astype(np.float32),
             pts_on_surf=near[:500], tex_on_surf=(0.5 + 0.5 * np.sin(3 * near[:500])).astype(np.float32))
    cfg = SimpleNamespace(enc_net_type="skip", fdim_geo=4, fdim_tex=8, fdim_up=64, hidden_dim=256, n_hidden_layers=4, data_type="sdftex",
                          enc_batch_size=2048, enc_n_iters=60, vol_ratio=0.1, fm_reso=32, sdf_loss="weightedl1", tex_loss="l1",
                          tex_weight=1.0, tex_threshold_ratio=0.999, sdf_renorm=0, enc_lr=5e-3, enc_lr_split=0.2, enc_lr_decay=0.1, gpu_id=0)
    ae = ShapeAutoEncoder(str(tmp_path / "encoding"), cfg)
    ae.train(path, log_every=20)
    assert ae.featmap_size == [24, 32, 20] or ae.featmap_size == [24, 32, 20]
    import json
    logs = [json.loads(l) for l in open(tmp_path / "encoding" / "progress.jsonl")]
    assert logs[-1]["sdf_loss"] < 0.6 * logs[0]["sdf_loss"], logs
    stat = json.load(open(tmp_path / "encoding" / "eval_stat.json"))
    assert np.isfinite(stat["mean_tsdf_l1_error"])
    ae2 = ShapeAutoEncoder(str(tmp_path / "encoding"), cfg)
    ae2.load_ckpt("final")
    fm = ae.encode()
    a = ae.decode_batch(fm, ae.pts_grid[:777]); b = ae2.decode_batch(fm, ae.pts_grid[:777])
    assert torch.equal(a, b)


def test_full_size_directional_derivative():
    """128^3 feature maps, 65 536 points (the training configuration): gradient vs a central finite difference of
    sdf_loss + tex_loss along a random direction of the smooth (non-bias-before-norm) parameters."""
    import torch
    H = W = D = 128
    N = 65536
    net = _net()
    g = torch.Generator(device="cuda").manual_seed(5)
    vol = torch.rand((1, 4, 2 * H, 2 * W, 2 * D), device="cuda", generator=g)
    vol[:, :1] = vol[:, :1] * 0.2 - 0.1
    pts = torch.rand((N, 3), device="cuda", generator=g) * 2 - 1
    sdf = torch.rand((N, 1), device="cuda", generator=g) * 0.1 - 0.05
    tex = torch.rand((N, 3), device="cuda", generator=g)
    cfg = _loss_cfg(0.05)
    flat = net.flat_parameters

    def loss_and_grad():
        losses, _, grad = net.loss_and_grads(vol, pts, sdf, tex, cfg)
        return float(losses.double().sum()), grad

    L0, grad = loss_and_grad()
    grad = grad.clone()
    assert torch.isfinite(grad).all()
    r = torch.randn(flat.shape, device="cuda", generator=g)
    v = grad / grad.norm() + r / r.norm()                      # half along the gradient, half random
    gv = float((grad.double() * v.double()).sum())
    eps = 1e-3 * L0 / abs(gv)                                  # small (0.1 % of the loss): the L1 losses and ReLUs have kinks
    base = flat.clone()
    vals = []
    for s in (+1.0, -1.0):
        flat.copy_(base + s * eps * v)
        net.mark_parameters_changed()
        vals.append(loss_and_grad()[0])
    flat.copy_(base); net.mark_parameters_changed()
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - gv) < 5e-2 * abs(gv), (fd, gv, L0, eps)


def test_weight_gradients_on_the_side_stream_change_no_bit():
    """The auto-encoder's backward pass enqueues its weight-gradient launches (the MLPs' k_wgrad_mfma<1>, the plane blocks'
    k_wgrad_mfma<25> / <1>) on a handle-owned side stream beside the chain of input gradients (one pre-activation-gradient buffer
    per layer, so the side stream reads what nobody overwrites); BWD_SIDE = 0 keeps them in line.  Selected through
    s3d_set_option in ONE process: identical gradients and losses, iteration after iteration."""
    import torch
    from sin3dm_amd import _lib
    H, W, D, N = 24, 32, 20, 8192
    thr = 0.05
    vol = _volume(H, W, D)

    def run():
        net = _net()
        net.reset_aabb(torch.tensor([-0.7, -1.0, -0.45, 0.7, 1.0, 0.45]).cuda())
        out = []
        for it in range(3):
            p, s, c = _batch(50 + it, N, thr)
            losses, _, grads = net.loss_and_grads(vol, p, s, c, _loss_cfg(thr))
            torch.cuda.synchronize()
            out.append((losses.clone(), grads.clone()))
        return out

    try:
        _lib.set_option("BWD_SIDE", 0)
        inline = run()
        _lib.set_option("BWD_SIDE", None)
        side = run()
    finally:
        _lib.set_option("BWD_SIDE", None)
    for (la, ga), (lb, gb) in zip(inline, side):
        assert torch.equal(la, lb) and torch.equal(ga, gb)
        assert torch.isfinite(ga).all() and float(ga.abs().max()) > 0
