"""Edge cases on the MI355X: empty / single inputs, no-surface grids, minimum plane sizes, error reporting."""
import numpy as np
import pytest

from sin3dm_amd import testing as T

pytestmark = pytest.mark.gpu


def _decoder():
    import torch
    from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip
    net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
    net.load_state_dict(T.synthetic_state_dict(T.ae_param_shapes(), 5), strict=False)
    return net.to(torch.device("cuda:0")).eval()


def test_decode_zero_and_one_point():
    import torch
    net = _decoder()
    fm = [torch.rand(1, 12, a, b, device="cuda") for a, b in ((6, 8), (6, 5), (8, 5))]
    out0 = net.decode(torch.empty((0, 3), device="cuda"), fm)
    assert out0.shape == (0, 4)
    p = torch.tensor([[0.1, -0.2, 0.3]], device="cuda")
    out1 = net.decode(p, fm)
    many = net.decode(p.repeat(130, 1), fm)
    assert out1.shape == (1, 4) and torch.isfinite(out1).all() and torch.equal(many, out1.expand(130, -1))


def test_marching_cubes_without_a_surface():
    import torch
    from sin3dm_amd.encoding.isosurface import marching_cubes
    g = torch.ones((9, 7, 5), device="cuda")
    v, t, _ = marching_cubes(g, 0.0, 1.0)
    assert v.shape == (0, 3) and t.shape == (0, 3)
    v, t, _ = marching_cubes(-g, 0.0, 1.0)                        # all inside: the padded border closes a box
    assert len(t) > 0 and float(v.min()) >= -1.0 and float(v.max()) <= 9.0
    v, t, _ = marching_cubes(torch.full((1, 1, 1), -1.0, device="cuda"), 0.0, 1.0)    # a single vertex: an octahedron
    assert v.shape == (6, 3) and t.shape == (8, 3)


def test_unet_minimum_and_asymmetric_sizes():
    """two levels need every plane dimension >= 2; (2,2,2) is the smallest triplane, (2,33,5) a lopsided one"""
    import torch
    from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
    m = TriplaneUNetModelSmall(12, 32, 12, use_scale_shift_norm=True)
    m.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 0))
    m.cuda().eval()
    for H, W, D in ((2, 2, 2), (2, 33, 5), (3, 3, 3)):
        x = torch.randn(1, 12, H + D, W + D, device="cuda")
        with torch.no_grad():
            y = m(x, torch.tensor([5], device="cuda"), H=H, W=W, D=D)
        assert y.shape == x.shape and torch.isfinite(y).all()
        assert torch.all(y[..., H:, W:] == 0)
    with pytest.raises(AssertionError):                              # a plane that cannot be halved
        with torch.no_grad():
            m(torch.randn(1, 12, 2, 2, device="cuda"), torch.tensor([5], device="cuda"), H=1, W=1, D=1)
    with pytest.raises(AssertionError):                              # composed map does not match (H, W, D)
        with torch.no_grad():
            m(torch.randn(1, 12, 9, 9, device="cuda"), torch.tensor([5], device="cuda"), H=4, W=4, D=4)


def test_training_step_on_tiny_and_odd_planes():
    import torch
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
    m = TriplaneUNetModelSmall(12, 32, 12, use_scale_shift_norm=True)
    m.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 0))
    m.cuda()
    diffusion = create_gaussian_diffusion(steps=1000, predict_xstart=True)
    for H, W, D in ((2, 2, 2), (3, 5, 2), (7, 2, 9)):
        x0 = torch.rand(3, 12, H + D, W + D, device="cuda") * 2 - 1
        terms, g = diffusion.training_losses_and_grads(m, x0, torch.tensor([0, 500, 999], device="cuda"), torch.ones(3, device="cuda"),
                                                       dict(H=H, W=W, D=D))
        assert torch.isfinite(terms["loss"]).all() and torch.isfinite(g).all() and float(g.abs().max()) > 0


def test_ae_texture_loss_with_empty_band_is_nan_like_the_reference():
    """F.l1_loss over an empty selection is nan in the reference (model.py:217); the sdf loss and its gradients stay finite"""
    import torch
    from sin3dm_amd import _lib
    from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip
    net = AutoEncoderGroupSkip(4, 8, 64, 256, 4)
    net.load_state_dict(T.synthetic_state_dict(T.ae_param_shapes(with_encoder=True), 5), strict=False)
    net.cuda()
    vol = torch.rand(1, 4, 8, 12, 8, device="cuda")
    pts = torch.rand(64, 3, device="cuda") * 2 - 1
    sdf = torch.full((64, 1), 0.05, device="cuda")                 # every point on the truncation value: outside the band
    tex = torch.rand(64, 3, device="cuda")
    losses, _, g = net.loss_and_grads(vol, pts, sdf, tex, _lib.AeLossCfg(1, 0, 0.05, 0.999, 1.0))
    assert torch.isfinite(losses[0]) and torch.isnan(losses[1])
    geo = net.split_flat(g)["geo_decoder.second_layers.4.weight"]
    assert torch.isfinite(geo).all() and float(geo.abs().max()) > 0
