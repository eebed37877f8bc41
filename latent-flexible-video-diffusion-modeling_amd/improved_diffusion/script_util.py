"""Factories and argparse helpers with the reference's names and defaults (reference script_util.py)."""
import argparse

from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps
from .unet import UNetVideoModel

# image_size -> channel_mult.  16 is a declared extension for the 4x16x16 latent benchmark of
# BASELINE.json (the reference factory rejects it, script_util.py:108-117; same multipliers as 32).
_CHANNEL_MULT = {256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4), 32: (1, 2, 2, 2), 16: (1, 2, 2, 2)}


def model_and_diffusion_defaults():
    """The 22 model/diffusion keys and their defaults (reference script_util.py:9-36)."""
    return dict(
        image_size=64, in_channels=3, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1,
        attention_resolutions="16,8", dropout=0.0, learn_sigma=False, sigma_small=False, class_cond=False,
        diffusion_steps=1000,
        diffusion_space_kwargs=dict(diffusion_space=None, pre_encoded=False, pre_encoded_stats_dict=None),
        noise_schedule="linear", timestep_respacing="", use_kl=False, predict_xstart=False, rescale_timesteps=True,
        rescale_learned_sigmas=True, use_checkpoint=False, use_scale_shift_norm=True, use_rpe_net=True)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, sigma_small, in_channels, num_channels,
                               num_res_blocks, num_heads, num_heads_upsample, attention_resolutions, dropout,
                               diffusion_steps, diffusion_space_kwargs, noise_schedule, timestep_respacing, use_kl,
                               predict_xstart, rescale_timesteps, rescale_learned_sigmas, use_checkpoint,
                               use_scale_shift_norm, use_rpe_net):
    model = create_model(image_size, in_channels, num_channels, num_res_blocks, learn_sigma=learn_sigma,
                         class_cond=class_cond, use_checkpoint=use_checkpoint,
                         attention_resolutions=attention_resolutions, num_heads=num_heads,
                         num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm,
                         dropout=dropout, use_rpe_net=use_rpe_net)
    diffusion = create_gaussian_diffusion(
        steps=diffusion_steps, learn_sigma=learn_sigma, sigma_small=sigma_small, noise_schedule=noise_schedule,
        use_kl=use_kl, predict_xstart=predict_xstart, rescale_timesteps=rescale_timesteps,
        rescale_learned_sigmas=rescale_learned_sigmas, timestep_respacing=timestep_respacing,
        diffusion_space_kwargs=diffusion_space_kwargs)
    return model, diffusion


def create_model(image_size, in_channels, num_channels, num_res_blocks, learn_sigma, class_cond, use_checkpoint,
                 attention_resolutions, num_heads, num_heads_upsample, use_scale_shift_norm, dropout, use_rpe_net):
    if image_size not in _CHANNEL_MULT:
        raise ValueError(f"unsupported image size: {image_size}")
    attention_ds = tuple(image_size // int(res) for res in attention_resolutions.split(","))
    return UNetVideoModel(
        in_channels=in_channels, model_channels=num_channels,
        out_channels=(in_channels if not learn_sigma else in_channels * 2), num_res_blocks=num_res_blocks,
        attention_resolutions=attention_ds, image_size=image_size, dropout=dropout,
        channel_mult=_CHANNEL_MULT[image_size], use_checkpoint=use_checkpoint, num_heads=num_heads,
        num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm, use_rpe_net=use_rpe_net)


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear",
                              use_kl=False, predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing="",
                              diffusion_space_kwargs={"diffusion_space": "pixel", "pre_encoded": False,
                                                      "pre_encoded_stats_dict": None}):
    betas = gd.get_named_beta_schedule(noise_schedule, steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.MSE
    if not timestep_respacing:
        timestep_respacing = [steps]
    if learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas,
        model_mean_type=gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON,
        model_var_type=var_type, loss_type=loss_type, rescale_timesteps=rescale_timesteps,
        diffusion_space_kwargs=diffusion_space_kwargs)


def add_dict_to_argparser(parser, default_dict):
    for k, v in default_dict.items():
        v_type = type(v)
        if v is None:
            v_type = str
        elif isinstance(v, bool):
            v_type = str2bool
        parser.add_argument(f"--{k}", default=v, type=v_type)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")
