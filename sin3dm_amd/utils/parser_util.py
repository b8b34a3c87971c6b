"""Command-line flags, defaults and args.json persistence with the reference's names
(src/utils/parser_util.py) so experiment directories written by either side are interchangeable."""
from __future__ import annotations

import argparse
import json
import os


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")


def diffusion_defaults():
    """Reference :75-85."""
    return dict(learn_sigma=False, steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False,
                predict_xstart=True, rescale_timesteps=False, rescale_learned_sigmas=False)


def diffusion_model_defaults():
    """Reference :88-99."""
    return dict(in_channels=12, model_channels=64, out_channels=12, num_res_blocks=1, dropout=0,
                channel_mult="1,2", use_checkpoint=False, use_fp16=False, use_scale_shift_norm=True)


def add_dict_to_argparser(parser, default_dict):
    for k, v in default_dict.items():
        v_type = str if v is None else (str2bool if isinstance(v, bool) else type(v))
        parser.add_argument(f"--{k}", default=v, type=v_type)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def add_base_options(parser):
    g = parser.add_argument_group("base")
    g.add_argument("--tag", type=str, required=True, help="checkpoint directory")
    g.add_argument("-g", "--gpu_id", default=0, type=int, help="Device id to use.")
    g.add_argument("--only_enc", action="store_true")


def add_encoding_training_options(parser):
    g = parser.add_argument_group("encoding")
    g.add_argument("--data_path", type=str, help="path to source data")
    g.add_argument("--enc_batch_size", type=int, default=65536)
    g.add_argument("--fm_reso", type=int, default=128, help="feature map resolution")
    g.add_argument("--sdf_renorm", type=int, default=0)
    g.add_argument("--data_type", type=str, default="sdftex", choices=["sdf", "sdftex", "sdfpbr"])
    g.add_argument("--enc_net_type", type=str, default="skip")
    g.add_argument("-fdg", "--fdim_geo", type=int, default=4)
    g.add_argument("-fdt", "--fdim_tex", type=int, default=8)
    g.add_argument("-fdup", "--fdim_up", type=int, default=64)
    g.add_argument("-hd", "--hidden_dim", type=int, default=256)
    g.add_argument("-nh", "--n_hidden_layers", type=int, default=4)
    g.add_argument("--enc_n_iters", type=int, default=25000)
    g.add_argument("--enc_lr", type=float, default=5e-3)
    g.add_argument("--enc_lr_decay", type=float, default=0.1)
    g.add_argument("--enc_lr_split", type=float, default=0.2)
    g.add_argument("--vol_ratio", type=float, default=0.1)
    g.add_argument("--tex_threshold_ratio", type=float, default=0.999)
    g.add_argument("--tex_weight", type=float, default=1.0)
    g.add_argument("--sdf_loss", type=str, default="weightedl1", choices=["l1", "weightedl1"])
    g.add_argument("--tex_loss", type=str, default="l1", choices=["l1", "l2", "huber"])


def add_diffusion_training_options(parser):
    g = parser.add_argument_group("diffusion")
    g.add_argument("--enc_log", type=str, default=None)
    g.add_argument("--diff_batch_size", type=int, default=32)
    g.add_argument("--diff_net_type", type=str, default="unet_small")
    g.add_argument("--diff_lr", type=float, default=5e-4)
    g.add_argument("--diff_n_iters", type=int, default=25000)
    g.add_argument("--schedule_sampler", type=str, default="uniform")
    g.add_argument("--ema_rate", type=float, default=0.9999)
    g.add_argument("--weight_decay", type=float, default=0.0)
    g.add_argument("--log_interval", type=int, default=100)
    g.add_argument("--save_interval", type=int, default=25000)
    add_dict_to_argparser(g, diffusion_defaults())
    add_dict_to_argparser(g, diffusion_model_defaults())


def add_sampling_options(parser):
    g = parser.add_argument_group("sampling")
    g.add_argument("--n_samples", type=int, default=1)
    g.add_argument("--input", type=str, default=None)
    g.add_argument("--output", type=str, default="results")
    g.add_argument("--resize", default=(1, 1, 1), type=float, nargs=3)
    g.add_argument("--use_ddim", type=str2bool, default=False)
    g.add_argument("--timestep_respacing", type=str, default="")
    g.add_argument("--app", type=str, default="generate")
    g.add_argument("--reso", type=int, default=256, help="decoding volume resolution")
    g.add_argument("--n_faces", type=int, default=10000)
    g.add_argument("--texreso", type=int, default=2048)
    g.add_argument("--vox", action="store_true")
    g.add_argument("--copy_mtl", type=str2bool, default=True)
    g.add_argument("--file_format", type=str, default="obj", choices=["obj", "glb"])


def get_args_by_group(parser, args, group_name):
    for group in parser._action_groups:
        if group.title == group_name:
            return {a.dest: getattr(args, a.dest, None) for a in group._group_actions}
    raise ValueError("group_name was not found.")


def load_and_overwrite_args(args, path, ignore_keys=()):
    with open(path, "r") as f:
        saved = json.load(f)
    for k, v in saved.items():
        if k not in ignore_keys:
            setattr(args, k, v)
    return args


def encoding_log_dir(exp_tag):
    return os.path.join(exp_tag, "encoding")


def diffusion_log_dir(exp_tag):
    return os.path.join(exp_tag, "diffusion")


def encoding_feat_path(exp_tag):
    return os.path.join(exp_tag, "encoding/feat.npz")


def diffusion_model_path(exp_tag, ema, step):
    return os.path.join(exp_tag, f"diffusion/ema_{ema}_{step:06d}.pt")


def sample_args(argv=None):
    """Sampling CLI: own flags, then the saved encoding/ and diffusion/ args.json (reference :148-169)."""
    parser = argparse.ArgumentParser()
    add_base_options(parser)
    add_sampling_options(parser)
    args = parser.parse_args(argv)
    if not os.path.exists(args.tag):
        raise ValueError(f"Experiment log does not exist: {args.tag}")
    load_and_overwrite_args(args, os.path.join(encoding_log_dir(args.tag), "args.json"))
    load_and_overwrite_args(args, os.path.join(diffusion_log_dir(args.tag), "args.json"),
                            ignore_keys=["timestep_respacing"])
    return args


def train_args(argv=None, confirm=input, write=True):
    """Training CLI; persists the `encoding` and `diffusion` groups as args.json (reference :102-145).
    `write=False` (ranks > 0 of a multi-process launch) only parses: the directory, symlink and args.json side effects
    belong to one process."""
    parser = argparse.ArgumentParser()
    add_base_options(parser)
    add_encoding_training_options(parser)
    add_diffusion_training_options(parser)
    args = parser.parse_args(argv)
    if write:
        if os.path.exists(args.tag) and confirm(f'Folder "{args.tag}" already exists, continue? (y/n) ') != "y":
            raise SystemExit(0)
        os.makedirs(args.tag, exist_ok=True)
    enc_dir, diff_dir = encoding_log_dir(args.tag), diffusion_log_dir(args.tag)
    if args.enc_log is not None:
        load_and_overwrite_args(args, os.path.join(args.enc_log, "args.json"))
        if write and not os.path.lexists(enc_dir):
            os.symlink(os.path.abspath(args.enc_log), enc_dir)
    elif write:
        os.makedirs(enc_dir, exist_ok=True)
        with open(os.path.join(enc_dir, "args.json"), "w") as f:
            json.dump(get_args_by_group(parser, args, "encoding"), f, indent=4)
    n = args.fdim_geo if args.data_type == "sdf" else args.fdim_geo + args.fdim_tex
    args.in_channels = args.out_channels = n
    if write:
        os.makedirs(diff_dir, exist_ok=True)
        with open(os.path.join(diff_dir, "args.json"), "w") as f:
            json.dump(get_args_by_group(parser, args, "diffusion"), f, indent=4)
    return args
