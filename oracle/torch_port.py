"""PyTorch-CPU functional port of the Sin3DM denoising step.  TEST / BASELINE INFRASTRUCTURE ONLY.

A second CPU restatement next to oracle/sin3dm_oracle.c: it runs the reference's algorithm through the very
ATen/oneDNN CPU ops the reference itself dispatches to (F.conv2d, F.group_norm, F.avg_pool2d,
F.interpolate), assembled functionally from a plain state_dict.  bench.py times it as the `cpu_baseline`
("port": the reference cannot travel to the GPU box) because it is the fastest honest CPU path we have —
4-5x faster than the plain-C oracle.  Pinned against the golden vectors in tests/test_oracle_golden.py.
Nothing under sin3dm_amd/ imports this file.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

PLANES = ("xy", "xz", "yz")


def timestep_embedding(t, dim):
    """src/diffusion/nn.py:103-121"""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def silu(x):
    return x * torch.sigmoid(x)


def tconv(sd, prefix, fm, pad, rollout):
    """TriplaneConv.forward, src/diffusion/unet_triplane.py:31-60"""
    xy, xz, yz = fm
    if rollout:
        xy_in = torch.cat([xy, yz.mean(-1, keepdim=True).transpose(-1, -2).expand_as(xy),
                           xz.mean(-1, keepdim=True).expand_as(xy)], 1)
        xz_in = torch.cat([xz, xy.mean(-1, keepdim=True).expand_as(xz), yz.mean(-2, keepdim=True).expand_as(xz)], 1)
        yz_in = torch.cat([yz, xy.mean(-2, keepdim=True).transpose(-1, -2).expand_as(yz),
                           xz.mean(-2, keepdim=True).expand_as(yz)], 1)
        fm = (xy_in, xz_in, yz_in)
    return tuple(F.conv2d(x, sd[f"{prefix}.conv_{p}.weight"], sd[f"{prefix}.conv_{p}.bias"], padding=pad)
                 for p, x in zip(PLANES, fm))


def tnorm(sd, prefix, fm):
    """TriplaneNorm = GroupNorm32(32, C) per plane, src/diffusion/unet_triplane.py:63-84"""
    return tuple(F.group_norm(x, 32, sd[f"{prefix}.norm_{p}.weight"], sd[f"{prefix}.norm_{p}.bias"], 1e-5)
                 for p, x in zip(PLANES, fm))


def resblock(sd, prefix, fm, emb, ssn, rollout):
    """TriplaneResBlock._forward, src/diffusion/unet_triplane.py:269-311"""
    h = tconv(sd, prefix + ".in_layers.2", tuple(silu(x) for x in tnorm(sd, prefix + ".in_layers.0", fm)), 1, rollout)
    e = F.linear(silu(emb), sd[prefix + ".emb_layers.1.weight"], sd[prefix + ".emb_layers.1.bias"])[..., None, None]
    if ssn:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = tuple(x * (1 + scale) + shift for x in tnorm(sd, prefix + ".out_layers.0", h))
    else:
        h = tnorm(sd, prefix + ".out_layers.0", tuple(x + e for x in h))
    h = tconv(sd, prefix + ".out_layers.2", tuple(silu(x) for x in h), 1, rollout)
    skip = tconv(sd, prefix + ".skip_connection", fm, 0, False) if prefix + ".skip_connection.conv_xy.weight" in sd else fm
    return tuple(a + b for a, b in zip(h, skip))


def unet_forward(sd, x, t, H, W, D, model_channels, channel_mult=(1, 2), use_scale_shift_norm=True, rollout=True):
    """TriplaneUNetModelSmall.forward, src/diffusion/unet_triplane.py:465-510"""
    emb = timestep_embedding(t, model_channels)
    emb = F.linear(silu(F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])),
                   sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    fm = (x[..., :H, :W], x[..., :H, W:], x[..., H:, :W].transpose(-1, -2))
    h = tconv(sd, "in_conv.0", fm, 0, False)
    hs = []
    for level in range(len(channel_mult)):
        if level:
            h = tuple(F.avg_pool2d(a, 2, 2) for a in h)
        h = resblock(sd, f"input_blocks.{level}.{0 if level == 0 else 1}", h, emb, use_scale_shift_norm, rollout)
        hs.append(h)
    for oi in range(len(channel_mult)):
        level = len(channel_mult) - 1 - oi
        if oi == 0:
            h = hs.pop()
        else:
            sk = hs.pop()
            h = tuple(a if a.shape[2:] == s.shape[2:] else F.interpolate(a, size=s.shape[2:], mode="bilinear", align_corners=False)
                      for a, s in zip(h, sk))
            h = tuple(torch.cat([a, s], 1) for a, s in zip(h, sk))
        h = resblock(sd, f"output_blocks.{oi}.0", h, emb, use_scale_shift_norm, rollout)
        if level > 0:
            h = tuple(F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=False) for a in h)
    xy, xz, yz = tconv(sd, "out.2", tuple(silu(a) for a in tnorm(sd, "out.0", h)), 0, False)
    out = x.new_zeros(x.shape[0], xy.shape[1], H + D, W + D)
    out[..., :H, :W], out[..., :H, W:], out[..., H:, :W] = xy, xz, yz.transpose(-1, -2)
    return out


def p_sample_update(model_out, x, eps, tab, t):
    """p_mean_variance (START_X, FIXED_LARGE, clip) + p_sample, src/diffusion/gaussian_diffusion.py:233-327, 396-440.
    tab: the float64 [8,T] table of oracle.schedule_tables()."""
    x0 = model_out.clamp(-1, 1)
    mean = float(tab[6][t]) * x0 + float(tab[7][t]) * x
    var = tab[5][1] if t == 0 else tab[0][t]
    return mean + (0.0 if t == 0 else 1.0) * math.exp(0.5 * math.log(var)) * eps, x0


# ---------------------------------------------------------------------------------------------- training tier
def q_sample(x0, noise, sqrt_ac, sqrt_1mac, t):
    """src/diffusion/gaussian_diffusion.py:189-207; sqrt_ac / sqrt_1mac: float64 numpy tables, t: int64 [B]."""
    a = torch.from_numpy(sqrt_ac).to(x0.device)[t].float()[:, None, None, None]
    b = torch.from_numpy(sqrt_1mac).to(x0.device)[t].float()[:, None, None, None]
    return a * x0 + b * noise


def training_losses(sd, x0, t, noise, tabs, H, W, D, predict_xstart=True, **unet_kw):
    """training_losses with MSE (gaussian_diffusion.py:771-856): per-plane mean-squared errors summed; the target is x0
    (ModelMeanType.START_X) or, with predict_xstart=False, the noise (EPSILON, :829-835).
    sd tensors may require grad; returns {"mse_xy","mse_xz","mse_yz","loss"} each [B], and x_t."""
    x_t = q_sample(x0, noise, tabs["sqrt_alphas_cumprod"], tabs["sqrt_one_minus_alphas_cumprod"], t)
    out = unet_forward(sd, x_t, t.float(), H, W, D, **unet_kw)
    terms = {}
    for name, tgt, o in zip(PLANES, _decompose(x0 if predict_xstart else noise, H, W, D), _decompose(out, H, W, D)):
        terms["mse_" + name] = ((tgt - o) ** 2).flatten(1).mean(1)
    terms["loss"] = terms["mse_xy"] + terms["mse_xz"] + terms["mse_yz"]
    return terms, x_t


def _decompose(c, H, W, D):
    return c[..., :H, :W], c[..., :H, W:], c[..., H:, :W].transpose(-1, -2)


def adamw_ema_step(params, grads, m, v, ema, step, lr, wd, ema_rate, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.AdamW step (decoupled decay, bias-corrected) followed by update_ema (src/diffusion/nn.py:55-65),
    in place on lists of tensors; `step` counts from 1."""
    c1, c2 = 1 - b1 ** step, 1 - b2 ** step
    for p, g, mi, vi, e in zip(params, grads, m, v, ema if ema else [None] * len(params)):
        p.mul_(1 - lr * wd)
        mi.mul_(b1).add_(g, alpha=1 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (vi.sqrt() / math.sqrt(c2)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / c1)
        if e is not None:
            e.mul_(ema_rate).add_(p, alpha=1 - ema_rate)


# ---------------------------------------------------------------------------------------------- auto-encoder tier
def _inorm(x, w=None, b=None, eps=1e-5):
    return F.instance_norm(x, weight=w, bias=b, eps=eps)


def ae_encode(sd, vol, geo_dim=4):
    """AutoEncoderGroupSkip.encode, src/encoding/networks.py:164-180: Conv3d(k4,s2,p1) -> axis means -> InstanceNorm2d
    -> tanh(0.5 x).  vol: [1, 1+tex_channels, 2H, 2W, 2D]."""
    geo = F.conv3d(vol[:, :1], sd["geo_encoder.weight"], sd["geo_encoder.bias"], stride=2, padding=1)
    tex = F.conv3d(vol, sd["tex_encoder.weight"], sd["tex_encoder.bias"], stride=2, padding=1)
    f = torch.cat([geo, tex], dim=1)
    return [(_inorm(f.mean(dim=d)) * 0.5).tanh() for d in (4, 3, 2)]


def ae_plane_block(sd, prefix, fm):
    """TriplaneGroupResnetBlock(ks=5, input_norm=False, input_act=False), src/encoding/blocks.py:189-256: the grouped
    conv is three independent per-plane convs (rows p*up:(p+1)*up of the weight)."""
    out = []
    up = sd[prefix + ".in_layers.0.weight"].shape[0] // 3
    for p, (name, x) in enumerate(zip(PLANES, fm)):
        rows = slice(p * up, (p + 1) * up)
        h = F.conv2d(x, sd[prefix + ".in_layers.0.weight"][rows], sd[prefix + ".in_layers.0.bias"][rows], padding=2)
        h = _inorm(h, sd[f"{prefix}.norm_{name}.weight"], sd[f"{prefix}.norm_{name}.bias"], eps=1e-6)
        h = F.conv2d(silu(h), sd[prefix + ".out_layers.1.weight"][rows], sd[prefix + ".out_layers.1.bias"][rows], padding=2)
        out.append(h + F.conv2d(x, sd[prefix + ".shortcut.weight"][rows], sd[prefix + ".shortcut.bias"][rows]))
    return out


def _mlp(sd, prefix, x, n_hidden_layers=4):
    """DecoderMLPSkipConcat, src/encoding/blocks.py:65-91."""
    n = n_hidden_layers // 2
    h = x
    for i in range(n + 1):
        h = F.relu(F.linear(h, sd[f"{prefix}.first_layers.{2 * i}.weight"], sd[f"{prefix}.first_layers.{2 * i}.bias"]))
    h = torch.cat([x, h], dim=-1)
    for i in range(n):
        h = F.relu(F.linear(h, sd[f"{prefix}.second_layers.{2 * i}.weight"], sd[f"{prefix}.second_layers.{2 * i}.bias"]))
    return F.linear(h, sd[f"{prefix}.second_layers.{2 * n}.weight"], sd[f"{prefix}.second_layers.{2 * n}.bias"])


def ae_decode(sd, pts, fm, aabb, geo_dim=4):
    """AutoEncoderGroupSkip.decode, src/encoding/networks.py:192-220."""
    x = 2 * (pts - aabb[:3]) / (aabb[3:] - aabb[:3]) - 1

    def sample(plane, xy):
        g = xy.view(1, 1, -1, 2).flip(-1)
        return F.grid_sample(plane, g, align_corners=False, padding_mode="border")[0, :, 0, :].transpose(0, 1)

    coords = ([0, 1], [0, 2], [1, 2])
    geo = ae_plane_block(sd, "geo_convs", [f[:, :geo_dim] for f in fm])
    tex = ae_plane_block(sd, "tex_convs", [f[:, geo_dim:] for f in fm])
    h_geo = sum(sample(geo[i], x[..., coords[i]]) for i in range(3))
    h_tex = sum(sample(tex[i], x[..., coords[i]]) for i in range(3))
    return torch.cat([_mlp(sd, "geo_decoder", h_geo), _mlp(sd, "tex_decoder", h_tex).sigmoid()], dim=1)


def ae_losses(pred, sdf, tex, sdf_threshold, tex_threshold_ratio=0.999, tex_weight=1.0):
    """ShapeAutoEncoder._forward_batch, src/encoding/model.py:186-237 (weightedl1 sdf loss, l1 texture loss on the
    points within the truncation band)."""
    ps = pred[..., :1]
    weight = 1 + 0.5 * torch.sign(sdf) * torch.sign(sdf - ps)
    sdf_loss = ((ps - sdf).abs() * weight).mean()
    mask = sdf.squeeze(1).abs() < sdf_threshold * tex_threshold_ratio
    tex_loss = F.l1_loss(pred[..., 1:][mask], tex[mask]) * tex_weight
    return {"sdf_loss": sdf_loss, "tex_loss": tex_loss}


def ae_param_groups(names):
    """geo_parameters / tex_parameters of AutoEncoderGroupSkip (networks.py:146-150) as name lists."""
    geo = [n for n in names if n.startswith(("geo_encoder", "geo_convs", "geo_decoder"))]
    tex = [n for n in names if n.startswith(("tex_encoder", "tex_convs", "tex_decoder"))]
    return geo, tex
