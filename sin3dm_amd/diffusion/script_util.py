"""Factory functions with the reference's names and argument keys (src/diffusion/script_util.py:7-64): they turn the
`diffusion` group of args.json into the HIP-backed UNet module and its (respaced) Gaussian diffusion."""
from __future__ import annotations

from ..utils.parser_util import args_to_dict, diffusion_defaults, diffusion_model_defaults
from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps
from .unet_triplane import TriplaneUNetModelSmall, TriplaneUNetModelSmallRaw

_NETWORKS = {"unet_small": TriplaneUNetModelSmall, "unet_raw": TriplaneUNetModelSmallRaw}


def _loss_type(use_kl, rescale_learned_sigmas):
    if use_kl:
        return gd.LossType.RESCALED_KL
    return gd.LossType.RESCALED_MSE if rescale_learned_sigmas else gd.LossType.MSE


def _var_type(learn_sigma, sigma_small):
    if learn_sigma:
        return gd.ModelVarType.LEARNED_RANGE
    return gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear",
                              use_kl=False, predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing=""):
    """A SpacedDiffusion over `steps` base timesteps; `timestep_respacing` "" keeps all of them."""
    kept = space_timesteps(steps, timestep_respacing if timestep_respacing else [steps])
    mean_type = gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON
    return SpacedDiffusion(use_timesteps=kept, betas=gd.get_named_beta_schedule(noise_schedule, steps),
                           model_mean_type=mean_type, model_var_type=_var_type(learn_sigma, sigma_small),
                           loss_type=_loss_type(use_kl, rescale_learned_sigmas), rescale_timesteps=rescale_timesteps)


def create_model_and_diffusion_from_args(args):
    """(model, diffusion) from an argparse namespace or a loaded args.json."""
    try:
        net_cls = _NETWORKS[args.diff_net_type]
    except KeyError:
        raise ValueError(f"unknown diff_net_type: {args.diff_net_type}") from None
    if isinstance(args.channel_mult, str):
        args.channel_mult = tuple(int(c) for c in args.channel_mult.split(","))
    model = net_cls(**args_to_dict(args, diffusion_model_defaults().keys()))
    diffusion = create_gaussian_diffusion(**args_to_dict(args, diffusion_defaults().keys()))
    return model, diffusion
