"""Deterministic synthetic weights and inputs for parity tests, smoke() and bench.py.

There are no pretrained checkpoints offline, so every test, fixture and
benchmark runs on seeded synthetic parameters.  The generator is numpy PCG64 keyed by the
crc32 of the parameter *name*, so a tensor's values depend on nothing but (name, shape, seed):
the fixture generator (tests/golden/make_golden.py, which loads these into the *reference*
modules) and the GPU-box tests regenerate bit-identical weights without shipping them.

Parameter names/shapes restate the reference constructors:
  TriplaneUNetModelSmall.__init__   src/diffusion/unet_triplane.py:346-449
  TriplaneResBlock.__init__         src/diffusion/unet_triplane.py:194-262
  AutoEncoderGroupSkip.__init__     src/encoding/networks.py:124-149
  TriplaneGroupResnetBlock.__init__ src/encoding/blocks.py:190-235
  DecoderMLPSkipConcat.__init__     src/encoding/blocks.py:66-83
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

PLANES = ("xy", "xz", "yz")


def _parse_mult(channel_mult):
    if isinstance(channel_mult, str):
        return tuple(int(c) for c in channel_mult.split(","))
    return tuple(int(c) for c in channel_mult)


def unet_param_shapes(in_channels=12, model_channels=64, out_channels=12, num_res_blocks=1,
                      channel_mult=(1, 2), use_scale_shift_norm=True, rollout=True):
    """name -> shape for TriplaneUNetModelSmall (rollout=True) / ...SmallRaw (rollout=False)."""
    channel_mult = _parse_mult(channel_mult)
    mc = model_channels
    ted = 4 * mc
    sh = OrderedDict()

    def lin(prefix, o, i):
        sh[prefix + ".weight"] = (o, i)
        sh[prefix + ".bias"] = (o,)

    def tconv(prefix, cin, cout, k, is_rollout):
        for p in PLANES:
            sh[f"{prefix}.conv_{p}.weight"] = (cout, cin * 3 if is_rollout else cin, k, k)
            sh[f"{prefix}.conv_{p}.bias"] = (cout,)

    def tnorm(prefix, c):
        for p in PLANES:
            sh[f"{prefix}.norm_{p}.weight"] = (c,)
            sh[f"{prefix}.norm_{p}.bias"] = (c,)

    def resblock(prefix, c, cout):
        tnorm(prefix + ".in_layers.0", c)
        tconv(prefix + ".in_layers.2", c, cout, 3, rollout)
        lin(prefix + ".emb_layers.1", 2 * cout if use_scale_shift_norm else cout, ted)
        tnorm(prefix + ".out_layers.0", cout)
        tconv(prefix + ".out_layers.2", cout, cout, 3, rollout)
        if c != cout:
            tconv(prefix + ".skip_connection", c, cout, 1, False)

    lin("time_embed.0", ted, mc)
    lin("time_embed.2", ted, ted)
    ch = input_ch = int(channel_mult[0] * mc)
    tconv("in_conv.0", in_channels, ch, 1, False)
    # The reference updates `ch` once per level and appends one skip width per level while the
    # decoder pops one per res block (unet_triplane.py:383-419): num_res_blocks != 1 cannot even be
    # constructed there (IndexError on input_block_chans.pop()), so only 1 is a valid topology.
    if num_res_blocks != 1:
        raise NotImplementedError("reference topology is only constructible with num_res_blocks=1")
    chans = [ch]
    for level, mult in enumerate(channel_mult):
        idx = 0 if level == 0 else 1  # a parameter-free Downsample sits at index 0
        resblock(f"input_blocks.{level}.{idx}", ch, int(mult * mc))
        ch = int(mult * mc)
        chans.append(ch)
    nlev = len(channel_mult)
    for oi, (level, mult) in enumerate(list(enumerate(channel_mult))[::-1]):
        for i in range(num_res_blocks):
            ich = chans.pop()
            if level == nlev - 1 and i == 0:
                ich = 0
            resblock(f"output_blocks.{oi}.{i}", ch + ich, int(mc * mult))
        ch = int(mc * mult)
    tnorm("out.0", ch)
    tconv("out.2", input_ch, out_channels, 1, False)
    return sh


def ae_param_shapes(geo_feat_channels=4, tex_feat_channels=8, feat_channel_up=64,
                    mlp_hidden_channels=256, mlp_hidden_layers=4, tex_channels=3, with_encoder=False):
    """name -> shape for AutoEncoderGroupSkip: the decode side, plus the two Conv3d encoders when with_encoder."""
    sh = OrderedDict()
    up, hid = feat_channel_up, mlp_hidden_channels
    if with_encoder:
        sh["geo_encoder.weight"] = (geo_feat_channels, 1, 4, 4, 4)
        sh["geo_encoder.bias"] = (geo_feat_channels,)
        sh["tex_encoder.weight"] = (tex_feat_channels, tex_channels + 1, 4, 4, 4)
        sh["tex_encoder.bias"] = (tex_feat_channels,)

    def block(prefix, cin):
        sh[prefix + ".in_layers.0.weight"] = (3 * up, cin, 5, 5)
        sh[prefix + ".in_layers.0.bias"] = (3 * up,)
        for p in PLANES:
            sh[f"{prefix}.norm_{p}.weight"] = (up,)
            sh[f"{prefix}.norm_{p}.bias"] = (up,)
        sh[prefix + ".out_layers.1.weight"] = (3 * up, up, 5, 5)
        sh[prefix + ".out_layers.1.bias"] = (3 * up,)
        sh[prefix + ".shortcut.weight"] = (3 * up, cin, 1, 1)
        sh[prefix + ".shortcut.bias"] = (3 * up,)

    def mlp(prefix, cin, cout):
        n = mlp_hidden_layers // 2
        sh[f"{prefix}.first_layers.0.weight"] = (hid, cin)
        sh[f"{prefix}.first_layers.0.bias"] = (hid,)
        for i in range(n):
            sh[f"{prefix}.first_layers.{2 * (i + 1)}.weight"] = (hid, hid)
            sh[f"{prefix}.first_layers.{2 * (i + 1)}.bias"] = (hid,)
        sh[f"{prefix}.second_layers.0.weight"] = (hid, cin + hid)
        sh[f"{prefix}.second_layers.0.bias"] = (hid,)
        for i in range(n - 1):
            sh[f"{prefix}.second_layers.{2 * (i + 1)}.weight"] = (hid, hid)
            sh[f"{prefix}.second_layers.{2 * (i + 1)}.bias"] = (hid,)
        sh[f"{prefix}.second_layers.{2 * n}.weight"] = (cout, hid)
        sh[f"{prefix}.second_layers.{2 * n}.bias"] = (cout,)

    block("geo_convs", geo_feat_channels)
    mlp("geo_decoder", up, 1)
    block("tex_convs", tex_feat_channels)
    mlp("tex_decoder", up, tex_channels)
    return sh


def synthetic_tensor(name, shape, seed=0):
    """fp32 array for parameter `name`: fan-in scaled weights, small biases, GN gains near 1."""
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    x = rng.standard_normal(size=shape)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = ".norm_" in name
    if is_norm and leaf == "weight":
        x = 1.0 + 0.2 * x
    elif leaf == "bias":
        x = 0.1 * x
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        x = x * (1.0 / np.sqrt(fan_in))
        if "emb_layers" in name or "time_embed" in name:
            x = x * 1.5
    return np.ascontiguousarray(x.astype(np.float32))


def synthetic_state_dict(shapes, seed=0, as_torch=True):
    out = OrderedDict()
    for name, shape in shapes.items():
        a = synthetic_tensor(name, shape, seed)
        if as_torch:
            import torch
            a = torch.from_numpy(a)
        out[name] = a
    return out


def synthetic_noise(shape, seed):
    """Seeded N(0,1) fp32 array (used where a fixture does not store the noise explicitly)."""
    rng = np.random.Generator(np.random.PCG64([seed, 0xA11CE]))
    return np.ascontiguousarray(rng.standard_normal(size=shape).astype(np.float32))
