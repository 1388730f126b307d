"""Factory / config: host-side mirror of `improved_diffusion/script_util.py` for the video model.

Same defaults dict, same keyword surface and the same derivations (channel_mult by image size,
attention downsample rates, bucket params, EPSILON + FIXED_LARGE defaults) so that
`create_video_model_and_diffusion(**args_to_dict(model_args, defaults.keys()))`
(scripts/video_sample.py:562-564) builds the HIP engine instead of an nn.Module.
"""
import argparse
import random

import numpy as np
import torch

from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps
from .unet import CondMargVideoModel


def video_model_and_diffusion_defaults():
    """script_util.py:15-57 (image defaults overlaid with the video keys)."""
    return dict(
        image_size=-1, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1,
        attention_resolutions="16,8", dropout=0.0, learn_sigma=False, sigma_small=False, class_cond=False,
        diffusion_steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False, predict_xstart=False,
        rescale_timesteps=True, rescale_learned_sigmas=True, use_checkpoint=False, use_scale_shift_norm=True,
        use_spatial_encoding=True, T=-1, use_frame_encoding=False, cross_frame_attention=True, do_cond_marg=True,
        enforce_position_invariance=False, temporal_augment_type="add_manyhead_presoftmax_time", use_rpe_net=True,
        cond_emb_type="channel", rp_alpha=None, rp_beta=None, rp_gamma=None, allow_interactions_between_padding=True,
    )


_CHANNEL_MULT = {256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4), 32: (1, 2, 2, 2)}


def create_video_model(T, image_size, num_channels, num_res_blocks, learn_sigma, class_cond, use_checkpoint,
                       attention_resolutions, num_heads, num_heads_upsample, use_scale_shift_norm, dropout,
                       use_spatial_encoding, use_frame_encoding, cross_frame_attention, do_cond_marg,
                       enforce_position_invariance, temporal_augment_type, use_rpe_net, rp_alpha, rp_beta, rp_gamma,
                       cond_emb_type, allow_interactions_between_padding):
    """script_util.py:229-300."""
    if image_size not in _CHANNEL_MULT:
        raise ValueError(f"unsupported image size: {image_size}")
    if class_cond:
        raise NotImplementedError("class_cond is not supported by the HIP engine")
    if not do_cond_marg:
        # script_util.py:275-300: ModelClass = UNetVideoModel, which is handed cond_emb_type=... and passes it on to
        # UNetModel.__init__, which does not take it -- the reference cannot construct this model (tools/gen_golden_r4.py
        # records the probe); same exception, same message
        raise TypeError("UNetModel.__init__() got an unexpected keyword argument 'cond_emb_type'")
    attention_ds = tuple(image_size // int(res) for res in attention_resolutions.split(","))
    bucket_params = dict(alpha=rp_alpha, beta=rp_beta, gamma=rp_gamma) if any([rp_alpha, rp_beta, rp_gamma]) else None
    return CondMargVideoModel(
        T=T, in_channels=3, model_channels=num_channels, out_channels=(3 if not learn_sigma else 6), num_res_blocks=num_res_blocks,
        attention_resolutions=attention_ds, dropout=dropout, channel_mult=_CHANNEL_MULT[image_size], num_classes=None,
        use_checkpoint=use_checkpoint, num_heads=num_heads, num_heads_upsample=num_heads_upsample,
        use_scale_shift_norm=use_scale_shift_norm, use_spatial_encoding=use_spatial_encoding,
        use_frame_encoding=use_frame_encoding, cross_frame_attention=cross_frame_attention,
        enforce_position_invariance=enforce_position_invariance, image_size=image_size,
        temporal_augment_type=temporal_augment_type, use_rpe_net=use_rpe_net, bucket_params=bucket_params,
        cond_emb_type=cond_emb_type, allow_interactions_between_padding=allow_interactions_between_padding)


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear",
                              use_kl=False, predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing=""):
    """script_util.py:405-436."""
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
        model_var_type=var_type, loss_type=loss_type, rescale_timesteps=rescale_timesteps)


def create_video_model_and_diffusion(T, image_size, class_cond, learn_sigma, sigma_small, num_channels,
                                     num_res_blocks, num_heads, num_heads_upsample, attention_resolutions, dropout,
                                     diffusion_steps, noise_schedule, timestep_respacing, use_kl, predict_xstart,
                                     rescale_timesteps, rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm,
                                     use_spatial_encoding, use_frame_encoding, cross_frame_attention, do_cond_marg,
                                     enforce_position_invariance, temporal_augment_type, use_rpe_net, rp_alpha,
                                     rp_beta, rp_gamma, cond_emb_type, allow_interactions_between_padding):
    """script_util.py:110-181."""
    model = create_video_model(
        T, image_size, num_channels, num_res_blocks, learn_sigma=learn_sigma, class_cond=class_cond,
        use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions, num_heads=num_heads,
        num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm, dropout=dropout,
        use_spatial_encoding=use_spatial_encoding, use_frame_encoding=use_frame_encoding,
        cross_frame_attention=cross_frame_attention, do_cond_marg=do_cond_marg,
        enforce_position_invariance=enforce_position_invariance, temporal_augment_type=temporal_augment_type,
        use_rpe_net=use_rpe_net, rp_alpha=rp_alpha, rp_beta=rp_beta, rp_gamma=rp_gamma, cond_emb_type=cond_emb_type,
        allow_interactions_between_padding=allow_interactions_between_padding)
    diffusion = create_gaussian_diffusion(
        steps=diffusion_steps, learn_sigma=learn_sigma, sigma_small=sigma_small, noise_schedule=noise_schedule,
        use_kl=use_kl, predict_xstart=predict_xstart, rescale_timesteps=rescale_timesteps,
        rescale_learned_sigmas=rescale_learned_sigmas, timestep_respacing=timestep_respacing)
    return model, diffusion


def add_dict_to_argparser(parser, default_dict):
    """script_util.py:439-446."""
    for k, v in default_dict.items():
        v_type = type(v)
        if v is None:
            v_type = str
        elif isinstance(v, bool):
            v_type = str2bool
        parser.add_argument(f"--{k}", default=v, type=v_type)


def args_to_dict(args, keys):
    """script_util.py:449-454."""
    backups = {"allow_interactions_between_padding": True}
    return {k: getattr(args, k) if hasattr(args, k) else backups[k] for k in keys}


def str2bool(v):
    """script_util.py:457-467."""
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")


def set_random_seed(seed, deterministic=False):
    """script_util.py:470-486."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
