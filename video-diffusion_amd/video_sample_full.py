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
uint8, existing files skipped).  No datasets or checkpoints ship with the reference: videos are synthetic unless a
checkpoint path is given.
"""
import argparse
import json
import logging
import os

import numpy as np
import torch

from . import inference_util, test_util
from .script_util import str2bool
from .video_sample import get_masks, load_model, save_samples

logger = logging.getLogger("video_sample_full")


def _window(samples, obs_frame_indices, latent_frame_indices, B, device):
    """video_sample_full.py:127-153 / :233-260 (non-adaptive branch): the window's tensors on the device.  `samples` stays
    resident on the device for the whole run: the horizontal loop touches every window at every timestep, and a host copy
    per (timestep, window) pair -- what the reference does -- is a device synchronisation per denoise step."""
    idx = torch.tensor(list(obs_frame_indices) + list(latent_frame_indices), dtype=torch.int64, device=samples.device)
    x0 = samples.index_select(1, idx)
    frame_indices = idx.view(1, -1).repeat((B, 1))
    obs_mask, latent_mask, kinda_marg_mask = get_masks(x0, len(obs_frame_indices))
    return [v.to(device) for v in (x0, obs_mask, latent_mask, kinda_marg_mask, frame_indices)]


@torch.no_grad()
def infer_video(mode, model, diffusion, batch, max_frames, obs_length, step_size=1, optimal_schedule_path=None, *,
                use_gradient_method=False, vertical_steps=0, observed_frames="x_0", save_all_timesteps=False):
    """video_sample_full.py:50-323 (non-adaptive modes).  `vertical_steps`, `observed_frames` and
    `save_all_timesteps` are the reference's `args.*` globals.  Returns (samples (B,T,C,H,W) ndarray,
    all_timestep_samples (B,num_timesteps,T,C,H,W) ndarray or a one-element array)."""
    if "adaptive" in mode:
        raise NotImplementedError(f"inference mode {mode!r} needs the LPIPS network (out of scope)")
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

    def schedule():
        return iter(inference_util.inference_strategies[mode](
            video_length=T, num_obs=obs_length, max_frames=max_frames, step_size=step_size,
            optimal_schedule_path=optimal_schedule_path))

    t_tensors = {}

    def t_of(ts, n):
        if (ts, n) not in t_tensors:
            t_tensors[ts, n] = torch.tensor([ts] * n, device=device)
        return t_tensors[ts, n]

    if vertical_steps > 0:                                             # :88-200
        vertical_diff_timesteps = list(range(nts))[::-1][:vertical_steps]
        for obs_frame_indices, latent_frame_indices in schedule():
            logger.info(f"Conditioning on {sorted(obs_frame_indices)} frames, predicting {sorted(latent_frame_indices)}.")
            x0, obs_mask, latent_mask, kinda_marg_mask, frame_indices = _window(samples, obs_frame_indices,
                                                                              latent_frame_indices, B, device)
            n_latent = len(latent_frame_indices)
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
            samples[:, latent_frame_indices] = local_samples[:, -n_latent:]
            if save_all_timesteps:
                all_local = torch.stack(all_local, dim=1)
                all_timestep_samples[:, :len(vertical_diff_timesteps), latent_frame_indices] = \
                    all_local[:, :len(vertical_diff_timesteps), -n_latent:]

    horizontal = []
    for timestep in list(range(nts))[::-1][vertical_steps:]:           # :202-315
        for obs_frame_indices, latent_frame_indices in schedule():
            x0, obs_mask, latent_mask, kinda_marg_mask, frame_indices = _window(samples, obs_frame_indices,
                                                                              latent_frame_indices, B, device)
            n_latent = len(latent_frame_indices)
            local_samples = diffusion.p_sample(
                model, x0, t=t_of(timestep, x0.shape[0]), clip_denoised=True,
                model_kwargs=dict(frame_indices=frame_indices, x0=x0, obs_mask=obs_mask, latent_mask=latent_mask,
                                  kinda_marg_mask=kinda_marg_mask, x_t_minus_1=x0, observed_frames=observed_frames),
                return_attn_weights=False, use_gradient_method=use_gradient_method)["sample"]
            samples[:, latent_frame_indices] = local_samples[:, -n_latent:]
        if save_all_timesteps:
            horizontal.append(samples.clone())
    if save_all_timesteps and horizontal:
        all_timestep_samples[:, vertical_steps:] = torch.stack(horizontal, dim=1)
    model.check_device_errors()        # a non-finite network output of any step (vd_device_errors bit 1) raises instead of ending up in the .npy
    return samples.cpu().numpy(), all_timestep_samples.cpu().numpy()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint_path", nargs="?", default="")
    ap.add_argument("--eval_dir", default=None)
    ap.add_argument("--inference_mode", default="autoreg", choices=sorted(inference_util.inference_strategies))
    ap.add_argument("--T", type=int, default=16, help="video length")
    ap.add_argument("--max_frames", type=int, default=10)
    ap.add_argument("--obs_length", type=int, default=4)
    ap.add_argument("--step_size", type=int, default=1)
    ap.add_argument("--vertical_steps", type=int, default=0)
    ap.add_argument("--observed_frames", default="x_0", choices=["x_0", "x_t", "x_t_minus_1"])
    ap.add_argument("--save_all_timesteps", type=str2bool, nargs="?", const=True, default=False)
    ap.add_argument("--use_ddim", type=str2bool, nargs="?", const=True, default=False)
    ap.add_argument("--timestep_respacing", default="ddim50")
    ap.add_argument("--batch_size", type=int, default=2)
    ap.add_argument("--num_videos", type=int, default=2)
    ap.add_argument("--sample_idx", type=int, default=0)
    ap.add_argument("--image_size", type=int, default=64)
    ap.add_argument("--num_channels", type=int, default=128)
    ap.add_argument("--num_res_blocks", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    from . import dist as vdist
    rank, local_rank, world = vdist.init()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed + rank)
    model, diffusion = load_model(args, device, rank, world)      # rank 0 reads the checkpoint; one broadcast of the packed image
    # results/<...>/<run identifier>/ (test_util.py:65-132, video_sample_full.py:712-724); synthetic runs have no
    # checkpoint.  Derived on rank 0 only (a '*latest' checkpoint is opened there for its step) and sent to the others.
    out_dir = None
    if rank == 0:
        if args.eval_dir is None and not args.checkpoint_path:
            args.eval_dir = "results/synthetic"
        out_dir = test_util.get_model_results_path(args) / test_util.get_eval_run_identifier(args)
        os.makedirs(out_dir / "samples", exist_ok=True)
        json_path = out_dir / "model_config.json"
        if not json_path.exists():
            with test_util.Protect(json_path):
                with open(json_path, "w") as f:
                    json.dump(model.config, f, indent=4)
    out_dir = vdist.broadcast_object(out_dir, src=0)
    n_tasks = (args.num_videos + args.batch_size - 1) // args.batch_size
    for task in vdist.task_ids(n_tasks, rank, world):
        idx = vdist.indices_for_task(task, args.batch_size, args.num_videos)
        g = torch.Generator().manual_seed(1234 + task)
        batch = torch.rand(len(idx), args.T, 3, args.image_size, args.image_size, generator=g) * 2 - 1
        recon, _ = infer_video(args.inference_mode, model, diffusion, batch, args.max_frames, args.obs_length, args.step_size,
                               vertical_steps=args.vertical_steps, observed_frames=args.observed_frames)
        for p in save_samples(recon, str(out_dir), first_index=idx[0], sample_idx=args.sample_idx):
            logger.info(f"*** Saved {p} ***")
    vdist.barrier()


if __name__ == "__main__":
    main()
