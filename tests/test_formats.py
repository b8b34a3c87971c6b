"""CPU-only: the on-disk formats of an experiment directory against files WRITTEN BY THE
REFERENCE'S OWN CODE (tests/golden/formats/, produced by tests/golden/make_golden.py:gen_formats):
feat.npz (src/utils/triplane_util.py:38-61), the grouped args.json pair (src/utils/parser_util.py:102-169), the
ema_<rate>_<step>.pt state_dict (src/diffusion/train_util.py:258-270) and the path helpers (parser_util.py:217-230).
The GPU half (build the model from those args, load the .pt, compare a forward) is tests/test_cli_gpu.py."""
import json
import os
import shutil

import numpy as np
import torch

from conftest import GOLDEN
from sin3dm_amd import testing as T
from sin3dm_amd.utils import parser_util as pu, triplane_util as tu

FMT = os.path.join(GOLDEN, "formats")
TRAIN_ARGV = ["--data_path", "data/towerruins.npz", "--fm_reso", "64", "--model_channels", "32", "--channel_mult", "1",
              "--enc_net_type", "skip", "--diff_n_iters", "3", "--ema_rate", "0.9999", "--use_scale_shift_norm", "True",
              "--enc_lr_split", "0.2", "--timestep_respacing", "100"]


def _json(name):
    with open(os.path.join(FMT, name)) as f:
        return json.load(f)


def reference_experiment(root):
    """An experiment directory assembled only from reference-written files, laid out as the reference lays it out."""
    tag = os.path.join(root, "exp")
    os.makedirs(os.path.join(tag, "encoding"))
    os.makedirs(os.path.join(tag, "diffusion"))
    shutil.copy(os.path.join(FMT, "encoding_args.json"), os.path.join(tag, "encoding", "args.json"))
    shutil.copy(os.path.join(FMT, "diffusion_args.json"), os.path.join(tag, "diffusion", "args.json"))
    shutil.copy(os.path.join(FMT, "feat.npz"), os.path.join(tag, "encoding", "feat.npz"))
    shutil.copy(os.path.join(FMT, "ema_0.9999_000003.pt"), os.path.join(tag, "diffusion", "ema_0.9999_000003.pt"))
    return tag


def test_feat_npz_read_and_written_like_the_reference(tmp_path):
    ref = np.load(os.path.join(FMT, "loaded.npz"))
    comp, sizes = tu.load_triplane_data(os.path.join(FMT, "feat.npz"), device="cpu")
    assert tuple(sizes) == tuple(int(v) for v in ref["sizes"])
    assert np.array_equal(comp.numpy(), ref["composed"])                     # incl. the zero D x D corner
    planes = tu.load_triplane_data(os.path.join(FMT, "feat.npz"), device="cpu", compose=False)
    mine = str(tmp_path / "enc" / "feat.npz")
    tu.save_triplane_data(mine, *[p.numpy() for p in planes])
    a, b = np.load(mine), np.load(os.path.join(FMT, "feat.npz"))
    assert a.files == b.files == ["feat_xy", "feat_xz", "feat_yz"]
    for k in a.files:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k])


def test_train_args_write_the_references_args_json(tmp_path):
    tag = str(tmp_path / "exp")
    args = pu.train_args(["--tag", tag] + TRAIN_ARGV, confirm=lambda _: "y")
    for grp in ("encoding", "diffusion"):
        with open(os.path.join(tag, grp, "args.json")) as f:
            mine = json.load(f)
        ref = _json(f"{grp}_args.json")
        assert list(mine) == list(ref), grp                                  # same keys in the same order
        assert mine == ref, grp
    ns = {k: v for k, v in vars(args).items() if k != "tag"}
    assert ns == _json("train_args_namespace.json")
    # rank > 0 of a multi-process launch parses the same namespace without touching the directory
    other = pu.train_args(["--tag", str(tmp_path / "absent")] + TRAIN_ARGV, write=False)
    assert {k: v for k, v in vars(other).items() if k != "tag"} == ns and not os.path.exists(str(tmp_path / "absent"))


def test_sample_args_read_the_references_directory(tmp_path):
    tag = reference_experiment(str(tmp_path))
    args = pu.sample_args(["--tag", tag, "--n_samples", "2", "--timestep_respacing", "10", "--resize", "1", "1.5", "1"])
    ns = {k: v for k, v in vars(args).items() if k != "tag"}
    ref = _json("sample_args_namespace.json")
    assert {k: (list(v) if isinstance(v, tuple) else v) for k, v in ns.items()} == ref
    assert args.timestep_respacing == "10"                                   # the CLI value survives the saved "100"
    loaded = np.load(os.path.join(FMT, "loaded.npz"))
    assert os.path.relpath(pu.diffusion_model_path(tag, args.ema_rate, args.diff_n_iters), tag) == str(loaded["model_path"])
    assert os.path.relpath(pu.encoding_feat_path(tag), tag) == str(loaded["feat_path"])
    assert os.path.exists(pu.diffusion_model_path(tag, args.ema_rate, args.diff_n_iters))


def test_reference_state_dict_loads_by_name(tmp_path):
    """ema_*.pt as TrainLoop.save writes it: the module built from the saved args takes it with strict key matching and
    writes back a file the reference's load_state_dict would take (same names, shapes, dtypes, order)."""
    from sin3dm_amd.diffusion.script_util import create_model_and_diffusion_from_args
    tag = reference_experiment(str(tmp_path))
    args = pu.sample_args(["--tag", tag])
    model, diffusion = create_model_and_diffusion_from_args(args)
    assert diffusion.num_timesteps == 1000 and len(model.channel_mult) == 1
    sd = torch.load(pu.diffusion_model_path(tag, args.ema_rate, args.diff_n_iters), map_location="cpu")
    res = model.load_state_dict(sd)                                          # strict
    assert not res.missing_keys and not res.unexpected_keys
    mine = model.state_dict()
    assert list(mine) == list(sd)
    for k in sd:
        assert mine[k].dtype == sd[k].dtype and torch.equal(mine[k], sd[k]), k
    want = T.synthetic_state_dict(T.unet_param_shapes(model_channels=32, channel_mult="1"), 0)
    assert all(torch.equal(sd[k], want[k]) for k in want)                    # the generator's weights, bit for bit


def test_reference_shaped_ae_checkpoint_parses():
    """formats/ckpt_final.pth is the dict the reference's ShapeAutoEncoder.save_ckpt writes (src/encoding/model.py:141-156),
    assembled by tests/golden/make_golden.py from the reference's own net / AdamW / ExponentialLR objects: every key the
    reference's load_ckpt reads (:158-176) is there, and the net entry has exactly this build's state_dict names/shapes."""
    import torch
    from sin3dm_amd.encoding.networks import AutoEncoderGroupSkip
    ck = torch.load(os.path.join(GOLDEN, "formats", "ckpt_final.pth"), map_location="cpu", weights_only=False)
    assert list(ck) == ["net", "optimizer", "scheduler", "Ka", "Kd", "Ks", "Ns", "aabb", "featmap_size"]
    g = np.load(os.path.join(GOLDEN, "formats", "ckpt_decode.npz"))
    net = AutoEncoderGroupSkip(*[int(v) for v in g["cfg"]])
    ours = net.state_dict()
    assert set(ours) == set(ck["net"]), set(ours) ^ set(ck["net"])
    for k, v in ck["net"].items():
        assert tuple(ours[k].shape) == tuple(v.shape), k
    assert len(ck["optimizer"]["param_groups"]) == 2 and ck["optimizer"]["param_groups"][0]["lr"] < ck["optimizer"]["param_groups"][1]["lr"]
    assert len(ck["aabb"]) == 6 and tuple(ck["featmap_size"]) == (8, 12, 6)
