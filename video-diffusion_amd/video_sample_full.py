"""The "vertical + horizontal" sampler of the reference's experiments (scripts/video_sample_full.py:50-323;
sample.sh:5-15, test_eval.sh:9-26), driving the same per-step HIP engine as video_sample.py with a different loop
nest:

  * vertical phase   -- for every window of the schedule: the first `vertical_steps` (noisiest) timesteps in a row,
                        starting from the window's current frames, observed frames taken from x_0;
  * horizontal phase -- for every remaining timestep (high to low): the WHOLE schedule again, ONE `p_sample` per
                        window at that timestep, `observed_frames` in {x_0, x_t, x_t_minus_1} choosing what the
                        observed slots show the network (and which timestep they are embedded with, unet.py:991-1013);
                        the window's own frames serve as x, x0 and x_t_minus_1.

    python -m video_diffusion_amd.video_sample_full --inference_mode autoreg --T 32 --obs_length 4 --max_frames 10 \
        --step_size 2 --timestep_respacing ddim50 --vertical_steps 10 --observed_frames x_t_minus_1

Results land where video_eval.py of the reference looks for them (test_util.py naming rules, samples/sample_%04d-%d.npy
uint8); the job around the sampler -- dataset items, sample indices, the skip-before-sampling of finished batches -- is
`video_sample.run`, shared with the windowed CLI.  `adaptive-*` modes (:78,103-113,187,223-234,306 of the reference) pick
their observed frames per batch item from the current samples before every window; `adaptive_distance='l2'` works on the
frames themselves, 'lpips' needs `inference_util.set_lpips_embedder`.
"""
import argparse
import json
import logging
import os

import numpy as np
import torch

from . import inference_util, test_util
from .script_util import str2bool
from .video_sample import add_job_arguments, get_masks, run

logger = logging.getLogger("video_sample_full")


def _window(samples, obs_frame_indices, latent_frame_indices, B, device, adaptive=False):
    """video_sample_full.py:127-153 / :233-260: the window's tensors on the device + (observed, latent) frame counts.
    `samples` stays resident on the device for the whole run: the horizontal loop touches every window at every timestep,
    and a host copy per (timestep, window) pair -- what the reference does -- is a device synchronisation per denoise step.
    Adaptive schedules hand over one index row per batch item; the window is then a per-item gather."""
    if adaptive:
        frame_indices = torch.cat([torch.as_tensor(obs_frame_indices, dtype=torch.int64).reshape(B, -1),
                                   torch.as_tensor(latent_frame_indices, dtype=torch.int64).reshape(B, -1)], dim=1).to(samples.device)
        x0 = samples[torch.arange(B, device=samples.device)[:, None], frame_indices]
        n_obs, n_latent = len(obs_frame_indices[0]), len(latent_frame_indices[0])
    else:
        idx = torch.tensor(list(obs_frame_indices) + list(latent_frame_indices), dtype=torch.int64, device=samples.device)
        x0 = samples.index_select(1, idx)
        frame_indices = idx.view(1, -1).repeat((B, 1))
        n_obs, n_latent = len(obs_frame_indices), len(latent_frame_indices)
    obs_mask, latent_mask, kinda_marg_mask = get_masks(x0, n_obs)
    return [v.to(device) for v in (x0, obs_mask, latent_mask, kinda_marg_mask, frame_indices)], n_obs, n_latent


def _write_back(dst, latent_frame_indices, local, n_obs, n_latent, adaptive):
    """dst[..., latent frames, ...] = the window's latent slots; `dst` / `local` are (B, T, ...) or (B, steps, T, ...)."""
    if adaptive:
        for i, li in enumerate(latent_frame_indices):
            dst[i][..., li, :, :, :] = local[i][..., n_obs:, :, :, :]
    else:
        dst[..., latent_frame_indices, :, :, :] = local[..., -n_latent:, :, :, :]


@torch.no_grad()
def infer_video(mode, model, diffusion, batch, max_frames, obs_length, step_size=1, optimal_schedule_path=None, *,
                use_gradient_method=False, vertical_steps=0, observed_frames="x_0", save_all_timesteps=False,
                adaptive_distance="lpips"):
    """video_sample_full.py:50-323.  `vertical_steps`, `observed_frames` and `save_all_timesteps` are the reference's
    `args.*` globals; `adaptive_distance` its `distance` (the reference passes 'lpips').  Returns (samples (B,T,C,H,W)
    ndarray, all_timestep_samples (B,num_timesteps,T,C,H,W) ndarray or a one-element array)."""
    adaptive = "adaptive" in mode
    B, T, C, H, W = batch.shape
    device = model.device
    batch = batch.to(device=device, dtype=torch.float32)
    samples = torch.zeros_like(batch)
    samples[:, :obs_length] = batch[:, :obs_length]
    if "goal-directed" in mode:
        samples[:, -5] = batch[:, -5]                    # the reference hands over ONE goal frame (index -5) here
    nts = diffusion.num_timesteps
    if save_all_timesteps:
        all_timestep_samples = torch.zeros([B, nts, T, C, H, W], device=device)
        all_timestep_samples[:, :, :obs_length] = samples[:, :obs_length].unsqueeze(1).expand(-1, nts, -1, -1, -1, -1)
    else:
        all_timestep_samples = torch.zeros([1])

    def windows():
        """One pass over the schedule; an adaptive strategy sees the current samples before each of its windows."""
        it = iter(inference_util.inference_strategies[mode](
            video_length=T, num_obs=obs_length, max_frames=max_frames, step_size=step_size,
            optimal_schedule_path=optimal_schedule_path, **(dict(distance=adaptive_distance) if adaptive else {})))
        while True:
            if adaptive:
                it.set_videos(samples.cpu())
            try:
                obs_frame_indices, latent_frame_indices = next(it)
            except StopIteration:
                return
            yield obs_frame_indices, latent_frame_indices

    t_tensors = {}

    def t_of(ts, n):
        if (ts, n) not in t_tensors:
            t_tensors[ts, n] = torch.tensor([ts] * n, device=device)
        return t_tensors[ts, n]

    if vertical_steps > 0:                                             # :88-200
        vertical_diff_timesteps = list(range(nts))[::-1][:vertical_steps]
        for obs_frame_indices, latent_frame_indices in windows():
            logger.info(f"Conditioning on {sorted(obs_frame_indices)} frames, predicting {sorted(latent_frame_indices)}.")
            (x0, obs_mask, latent_mask, kinda_marg_mask, frame_indices), n_obs, n_latent = _window(
                samples, obs_frame_indices, latent_frame_indices, B, device, adaptive)
            model_kwargs = dict(frame_indices=frame_indices, x0=x0, obs_mask=obs_mask, latent_mask=latent_mask,
                                kinda_marg_mask=kinda_marg_mask, x_t_minus_1=x0, observed_frames="x_0")
            local_samples = x0.clone()
            all_local = []
            for timestep in vertical_diff_timesteps:
                local_samples = diffusion.p_sample(model, local_samples, t=t_of(timestep, x0.shape[0]), clip_denoised=True,
                                                   model_kwargs=model_kwargs, return_attn_weights=False,
                                                   use_gradient_method=use_gradient_method)["sample"]
                if save_all_timesteps:
                    all_local.append(local_samples.clone())
            _write_back(samples, latent_frame_indices, local_samples, n_obs, n_latent, adaptive)
            if save_all_timesteps:
                _write_back(all_timestep_samples[:, :len(vertical_diff_timesteps)], latent_frame_indices,
                            torch.stack(all_local, dim=1), n_obs, n_latent, adaptive)

    horizontal = []
    for timestep in list(range(nts))[::-1][vertical_steps:]:           # :202-315
        for obs_frame_indices, latent_frame_indices in windows():
            (x0, obs_mask, latent_mask, kinda_marg_mask, frame_indices), n_obs, n_latent = _window(
                samples, obs_frame_indices, latent_frame_indices, B, device, adaptive)
            local_samples = diffusion.p_sample(
                model, x0, t=t_of(timestep, x0.shape[0]), clip_denoised=True,
                model_kwargs=dict(frame_indices=frame_indices, x0=x0, obs_mask=obs_mask, latent_mask=latent_mask,
                                  kinda_marg_mask=kinda_marg_mask, x_t_minus_1=x0, observed_frames=observed_frames),
                return_attn_weights=False, use_gradient_method=use_gradient_method)["sample"]
            _write_back(samples, latent_frame_indices, local_samples, n_obs, n_latent, adaptive)
        if save_all_timesteps:
            horizontal.append(samples.clone())
    if save_all_timesteps and horizontal:
        all_timestep_samples[:, vertical_steps:] = torch.stack(horizontal, dim=1)
    model.check_device_errors()        # a non-finite network output of any step (vd_device_errors bit 1) raises instead of ending up in the .npy
    return samples.cpu().numpy(), all_timestep_samples.cpu().numpy()


def _infer(args, model, diffusion, batch, optimal_schedule_path):
    return infer_video(args.inference_mode, model, diffusion, batch, args.max_frames, args.obs_length, args.step_size,
                       optimal_schedule_path, use_gradient_method=args.use_gradient_method, vertical_steps=args.vertical_steps,
                       observed_frames=args.observed_frames, save_all_timesteps=args.save_all_timesteps,
                       adaptive_distance=args.adaptive_distance)


def main(argv=None):
    """The reference's option surface (video_sample_full.py:560-690) = the windowed CLI's + --vertical_steps."""
    ap = add_job_arguments(argparse.ArgumentParser())
    ap.add_argument("--vertical_steps", type=int, default=0)
    ap.add_argument("--adaptive_distance", default="l2", choices=["l2", "lpips"])
    args = ap.parse_args(argv)
    return run(args, infer=_infer)


if __name__ == "__main__":
    main()
