"""Factory functions with the reference's names and argument keys (src/diffusion/script_util.py)."""
from __future__ import annotations

from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps
from .unet_triplane import TriplaneUNetModelSmall, TriplaneUNetModelSmallRaw
from ..utils.parser_util import args_to_dict, diffusion_defaults, diffusion_model_defaults


def create_model_and_diffusion_from_args(args):
    """(model, diffusion) from an argparse namespace / args.json (reference: script_util.py:7-19)."""
    diffusion = create_gaussian_diffusion(**args_to_dict(args, diffusion_defaults().keys()))
    if isinstance(args.channel_mult, str):
        args.channel_mult = tuple(int(c) for c in args.channel_mult.split(","))
    kwargs = args_to_dict(args, diffusion_model_defaults().keys())
    if args.diff_net_type == "unet_small":
        model = TriplaneUNetModelSmall(**kwargs)
    elif args.diff_net_type == "unet_raw":
        model = TriplaneUNetModelSmallRaw(**kwargs)
    else:
        raise ValueError(f"unknown diff_net_type: {args.diff_net_type}")
    return model, diffusion


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear",
                              use_kl=False, predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing=""):
    """Reference: script_util.py:22-64."""
    betas = gd.get_named_beta_schedule(noise_schedule, steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.MSE
    if learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing or [steps]),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON,
        model_var_type=var_type,
        loss_type=loss_type,
        rescale_timesteps=rescale_timesteps,
    )
