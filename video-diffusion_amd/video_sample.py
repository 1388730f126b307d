"""Windowed conditional sampling: the caller of the hot path (scripts/video_sample.py:31-47, 50-190,
266-271 of the reference), driving the HIP engine.

    python -m video_diffusion_amd.video_sample --synthetic --inference_mode autoreg --T 32 \
        --obs_length 4 --max_frames 10 --step_size 2 --timestep_respacing ddim50 --out_dir /tmp/samples

`infer_video` keeps the reference's contract: `batch` (B, T, C, H, W) in [-1, 1]; the first
`obs_length` frames are observed; each window starts from `x0.clone()` (observed frames + whatever
the latent slots currently hold, zeros at first -- SURVEY F5), walks every respaced timestep through
`diffusion.p_sample`, and writes the last `n_latent` frames back.  One H2D transfer of the window's
inputs and one D2H of its result per window; nothing crosses the host inside the step loop.
No datasets or checkpoints ship with the reference: the CLI samples from synthetic
videos and either a checkpoint given on the command line or the closed-form weights.
"""
import argparse
import logging
import os

import numpy as np
import torch

from . import inference_util
from .script_util import (args_to_dict, create_video_model_and_diffusion, str2bool,
                          video_model_and_diffusion_defaults)
from .weights_init import synth_param

logger = logging.getLogger("video_sample")
drange = [-1, 1]


def get_masks(x0, num_obs):
    """video_sample.py:31-47: first `num_obs` frames observed, the rest latent, none kinda-marginal."""
    obs_mask = torch.zeros_like(x0[:, :, :1, :1, :1])
    obs_mask[:, :num_obs] = 1
    latent_mask = 1 - obs_mask
    kinda_marg_mask = torch.zeros_like(x0[:, :, :1, :1, :1])
    return obs_mask, latent_mask, kinda_marg_mask


@torch.no_grad()
def infer_video(mode, model, diffusion, batch, max_frames, obs_length, step_size=1, optimal_schedule_path=None, *,
                use_gradient_method=False, observed_frames="x_0", sampler="p_sample", eta=0.0, executor="eager",
                adaptive_distance="lpips", prefix_cache=False, suffix_skip=False):
    """video_sample.py:50-190.  Returns (samples ndarray (B,T,C,H,W), None).

    'adaptive-*' modes (:74,94-95,104-118,176-183): the strategy sees the current samples before every window and hands
    back one index list per batch item.  `adaptive_distance` is the reference's `distance` ('lpips' is what its script
    passes and needs `inference_util.set_lpips_embedder`; 'l2' works on the frames themselves).

    executor='eager' (default): one `diffusion.p_sample` call per step from the host with `th.randn_like` noise, the
    reference's own loop.  executor='graph': each window's step loop runs on the window executor -- one captured hipGraph
    per window shape, step index and noise counter on the device (executor.py).  The device is the bottleneck either way
    (the eager host loop runs ahead of it; measured 7.28 ms eager vs 7.49 ms graph per step at B=1 x T=16 and 28.9 vs 29.1
    at the headline window, profiles/r03a_*), so the graph is an option -- a host too slow to stay ahead, one launch per
    step to trace -- not the default.  `prefix_cache` (graph executor, 'x_0' mode): the observed frames' activations before
    the first attention layer once per window instead of once per step (executor.py; 2.6 % at 4 observed frames of 16, 12 %
    at 10 of 20, profiles/r03x_*).  `suffix_skip` (graph executor, 'x_0' / 'x_t_minus_1'): everything behind the last attention
    layer runs without the purely observed frames, whose step output this function never reads (write_back keeps the latent
    frames only); the latent frames are those of the full step (executor.py, include/vd_amd.h: vd_set_window_suffix_skip)."""
    adaptive = "adaptive" in mode
    B, T, C, H, W = batch.shape
    device = model.device
    samples = torch.zeros_like(batch).cpu()
    samples[:, :obs_length] = batch[:, :obs_length].cpu()
    if "goal-directed" in mode:
        samples[:, -5] = batch[:, -5].cpu()              # the reference hands over ONE goal frame (index -5) here
    schedule = iter(inference_util.inference_strategies[mode](
        video_length=T, num_obs=obs_length, max_frames=max_frames, step_size=step_size,
        optimal_schedule_path=optimal_schedule_path, **(dict(distance=adaptive_distance) if adaptive else {})))
    timesteps = list(range(diffusion.num_timesteps))[::-1]
    t_tensors = None
    use_graph = executor == "graph" and observed_frames in ("x_0", "x_t", "x_t_minus_1") and not use_gradient_method
    if use_graph:
        from .executor import WindowExecutor
        wex = getattr(model, "_window_executor", None)
        if wex is None or wex.diffusion is not diffusion or wex.prefix_cache != bool(prefix_cache) or wex.suffix_skip != bool(suffix_skip):
            wex = model._window_executor = WindowExecutor(model, diffusion, prefix_cache=prefix_cache, suffix_skip=suffix_skip)
    while True:
        if adaptive:
            schedule.set_videos(samples)
        try:
            obs_frame_indices, latent_frame_indices = next(schedule)
        except StopIteration:
            break
        logger.info(f"Conditioning on {sorted(obs_frame_indices)} frames, predicting {sorted(latent_frame_indices)}.")
        if adaptive:                   # one (obs, latent) index row per batch item
            frame_indices = torch.cat([torch.tensor(obs_frame_indices, dtype=torch.int64).reshape(B, -1),
                                       torch.tensor(latent_frame_indices, dtype=torch.int64).reshape(B, -1)], dim=1)
            x0 = torch.stack([samples[i, fi] for i, fi in enumerate(frame_indices)], dim=0).clone()
            n_obs_w, n_latent = len(obs_frame_indices[0]), len(latent_frame_indices[0])
        else:
            x0 = torch.cat([samples[:, obs_frame_indices], samples[:, latent_frame_indices]], dim=1).clone()
            frame_indices = torch.cat([torch.tensor(obs_frame_indices, dtype=torch.int64),
                                       torch.tensor(latent_frame_indices, dtype=torch.int64)], dim=0).repeat((B, 1))
            n_obs_w, n_latent = len(obs_frame_indices), len(latent_frame_indices)
        obs_mask, latent_mask, kinda_marg_mask = get_masks(x0, n_obs_w)

        def write_back(local):
            if adaptive:
                for i, li in enumerate(latent_frame_indices):
                    samples[i, li] = local[i, n_obs_w:].cpu()
            else:
                samples[:, latent_frame_indices] = local[:, -n_latent:].cpu()
        x0, obs_mask, latent_mask, kinda_marg_mask, frame_indices = [
            v.to(device) for v in (x0, obs_mask, latent_mask, kinda_marg_mask, frame_indices)]
        model_kwargs = dict(frame_indices=frame_indices, x0=x0, obs_mask=obs_mask, latent_mask=latent_mask,
                            kinda_marg_mask=kinda_marg_mask, x_t_minus_1=x0, observed_frames=observed_frames)
        if t_tensors is None:          # the reference re-creates this tensor every step (video_sample.py:154-155)
            t_tensors = [torch.tensor([ts] * B, device=device) for ts in range(diffusion.num_timesteps)]
        if use_graph:
            # renoise=False: like the eager loop below (and scripts/video_sample.py:149-166), every step reads x_t_minus_1 = x0 as it is
            write_back(wex.sample_window(x0, model_kwargs, sampler=sampler, eta=eta, renoise=False))
            model.check_device_errors()
            continue
        local_samples = x0.clone()
        for timestep in timesteps:
            if sampler == "p_sample":
                local_samples = diffusion.p_sample(model, local_samples, t=t_tensors[timestep], clip_denoised=True,
                                                   model_kwargs=model_kwargs, return_attn_weights=False,
                                                   use_gradient_method=use_gradient_method)["sample"]
            else:
                local_samples = diffusion.ddim_sample(model, local_samples, t=t_tensors[timestep], clip_denoised=True,
                                                      model_kwargs=model_kwargs, eta=eta)["sample"]
        write_back(local_samples)
        model.check_device_errors()    # once per window (write_back has synchronised): a non-finite network output of any of its steps raises here
    return samples.numpy(), None


def to_uint8(recon):
    """video_sample.py:266-268: ((x+1)/2*255) truncated to uint8."""
    return ((recon - drange[0]) / (drange[1] - drange[0]) * 255).astype(np.uint8)


def save_samples(recon, out_dir, first_index=0, sample_idx=0):
    """samples/sample_%04d-%d.npy, uint8 (T,3,H,W); existing files are skipped (video_sample.py:231-236,269-271)."""
    os.makedirs(os.path.join(out_dir, "samples"), exist_ok=True)
    written = []
    u8 = to_uint8(recon)
    for i in range(len(u8)):
        path = os.path.join(out_dir, "samples", f"sample_{first_index + i:04d}-{sample_idx}.npy")
        if not os.path.exists(path):
            np.save(path, u8[i])
            written.append(path)
    return written


def load_model(args, device, rank=0, world=1, create=None):
    """video_sample.py:547-567: checkpoint dict {'state_dict','config','step'} -> (model, diffusion).

    One process per GPU: ONLY rank 0 opens the checkpoint.  Its `config` (a small dict) goes to the other ranks with
    `broadcast_object_list`, every rank builds the same engine from it, rank 0 packs the state_dict into the engine's
    kernel-ready image and that ONE buffer travels by a single broadcast (RCCL over xGMI; `dist.share_weights`) -- in
    place of the reference's N checkpoint reads or its per-tensor `sync_params` (dist_util.py:139-143)."""
    from . import dist as vdist
    create = create or create_video_model_and_diffusion
    defaults = video_model_and_diffusion_defaults()
    holder = {}
    if rank == 0:
        if args.checkpoint_path:
            data = torch.load(args.checkpoint_path, map_location="cpu")
            cfg = dict(data["config"])
            cfg.setdefault("enforce_position_invariance", False)      # back-compat fills, video_sample.py:25-28,557-559
            cfg.setdefault("cond_emb_type", "channel")
            holder["sd"] = data["state_dict"]
        else:
            cfg = dict(defaults, T=args.max_frames, image_size=args.image_size, num_channels=args.num_channels,
                       num_res_blocks=args.num_res_blocks, rp_alpha=args.max_frames, rp_beta=args.max_frames,
                       rp_gamma=args.max_frames)
        cfg["timestep_respacing"] = args.timestep_respacing
    else:
        cfg = None
    cfg = vdist.broadcast_object(cfg, src=0)
    ns = argparse.Namespace(**cfg)
    model, diffusion = create(**args_to_dict(ns, defaults.keys()))
    model.config = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, type(None), list, tuple, dict))}
    model.to(device)
    model.eval()

    def state_dict_fn():                                              # called on rank 0 only
        if "sd" not in holder:
            holder["sd"] = {k: torch.from_numpy(synth_param(k, s)) for k, s in model.param_specs()}
        return holder["sd"]

    vdist.share_weights(model, state_dict_fn, rank)
    return model, diffusion


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint_path", nargs="?", default="")
    ap.add_argument("--synthetic", type=str2bool, nargs="?", const=True, default=True)
    ap.add_argument("--inference_mode", default="autoreg", choices=sorted(inference_util.inference_strategies))
    ap.add_argument("--T", type=int, default=16, help="video length")
    ap.add_argument("--max_frames", type=int, default=10)
    ap.add_argument("--obs_length", type=int, default=4)
    ap.add_argument("--step_size", type=int, default=1)
    ap.add_argument("--batch_size", type=int, default=2)
    ap.add_argument("--num_videos", type=int, default=2)
    ap.add_argument("--timestep_respacing", default="ddim50")
    ap.add_argument("--observed_frames", default="x_0")
    ap.add_argument("--image_size", type=int, default=64)
    ap.add_argument("--num_channels", type=int, default=128)
    ap.add_argument("--num_res_blocks", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--adaptive_distance", default="l2", choices=["l2", "lpips"],
                    help="adaptive-* modes: frame embedding for the farthest-point selection (lpips needs set_lpips_embedder)")
    ap.add_argument("--executor", default="eager", choices=["graph", "eager"],
                    help="eager: one p_sample call per step (default); graph: one captured hipGraph per window shape (executor.py)")
    ap.add_argument("--suffix_skip", type=str2bool, nargs="?", const=True, default=False,
                    help="with --executor graph and observed_frames x_0 / x_t_minus_1: run the network behind its last attention layer "
                         "on the non-observed frames only")
    ap.add_argument("--prefix_cache", type=str2bool, nargs="?", const=True, default=False,
                    help="with --executor graph and observed_frames x_0: compute the observed frames' encoder prefix once per window")
    ap.add_argument("--eval_dir", default=None,
                    help="results directory; default: derived from the checkpoint path and the sampling options "
                         "(test_util.get_model_results_path), 'results/synthetic' without a checkpoint")
    ap.add_argument("--out_dir", default=None, help="alias of --eval_dir (earlier rounds' flag)")
    ap.add_argument("--use_ddim", type=str2bool, nargs="?", const=True, default=False)
    ap.add_argument("--sample_idx", type=int, default=0)
    args = ap.parse_args(argv)
    return run(args)


def run(args, create=None, device=None):
    """The body of `main` (video_sample.py:520-640 of the reference): join the job, load + share the weights, walk this
    rank's tasks, write `samples/sample_%04d-%d.npy` under the reference's results/<...>/<run id>/ naming.
    `create` / `device` let the CPU tests drive the sharding + broadcast path with a stand-in engine."""
    import json
    from . import test_util
    logging.basicConfig(level=logging.INFO)
    from . import dist as vdist
    rank, local_rank, world = vdist.init()
    if device is None:
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    torch.manual_seed(args.seed + rank)
    model, diffusion = load_model(args, device, rank, world, create=create)
    # results/<checkpoint subpath>/<stem>[_<step>][_ddim][_respace<X>]/<mode>_<max_frames>_<step_size>_<T>_<obs_length>/
    # (test_util.py:65-132 of the reference, video_sample.py:600-611): what video_eval.py reads
    # -- derived on rank 0 (a '*latest' checkpoint is opened once more there for its step) and sent to the others
    out_dir = None
    if rank == 0:
        if args.eval_dir is None:
            args.eval_dir = args.out_dir if args.out_dir is not None else (None if args.checkpoint_path else "results/synthetic")
        out_dir = test_util.get_model_results_path(args) / test_util.get_eval_run_identifier(args)
        os.makedirs(out_dir / "samples", exist_ok=True)
        json_path = out_dir / "model_config.json"                         # video_sample.py:620-626
        if not json_path.exists():
            with test_util.Protect(json_path):
                with open(json_path, "w") as f:
                    json.dump(model.config, f, indent=4)
    out_dir = vdist.broadcast_object(out_dir, src=0)
    n_tasks = (args.num_videos + args.batch_size - 1) // args.batch_size
    for task in vdist.task_ids(n_tasks, rank, world):                       # video_sample.py:577-582
        idx = vdist.indices_for_task(task, args.batch_size, args.num_videos)
        g = torch.Generator().manual_seed(1234 + task)
        batch = torch.rand(len(idx), args.T, 3, args.image_size, args.image_size, generator=g) * 2 - 1
        recon, _ = infer_video(args.inference_mode, model, diffusion, batch, args.max_frames, args.obs_length,
                               args.step_size, observed_frames=args.observed_frames, executor=args.executor,
                               adaptive_distance=args.adaptive_distance, prefix_cache=getattr(args, "prefix_cache", False),
                               suffix_skip=getattr(args, "suffix_skip", False))
        for p in save_samples(recon, str(out_dir), first_index=idx[0], sample_idx=args.sample_idx):
            logger.info(f"*** Saved {p} ***")
    vdist.barrier()
    return out_dir


if __name__ == "__main__":
    main()
