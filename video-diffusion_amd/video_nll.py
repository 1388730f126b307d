"""NLL evaluation: the caller of `calc_bpd_loop_subsampled` (contract of scripts/video_nll.py:142-186 of the reference) and the job around it
(`run` / `main`: :31-140,187-352 -- frame-index lists, one metrics pickle per video under elbos/, finished videos skipped).

Per batch item the window is its observed frames followed by its latent frames; items are ragged, so shorter ones are
padded with zero frames that belong to neither mask.  The window is built here as ONE gather over a padded (item, slot) ->
frame table instead of per-item assignments.  Returned: every metric summed over the timestep axis and multiplied by the
window length (the reference reports bits per frame-dim summed, not averaged, over frames), as numpy arrays.  The reference's
script omits `observed_frames` / `x_t_minus_1` from model_kwargs, which its own `CondMargVideoModel.forward`
(unet.py:958-974) then fails to find; here they default to 'x_0' / x0.
"""
import torch
from torch.nn.utils.rnn import pad_sequence


def _window_table(obs_indices, lat_indices, n_items):
    """(frame table [n_items, F] long, observed [n_items, F] bool, latent [n_items, F] bool); F = the longest window."""
    pairs = list(zip(obs_indices, lat_indices))
    pairs += [((), ())] * (n_items - len(pairs))
    rows = [torch.as_tensor(list(o) + list(l), dtype=torch.long) for o, l in pairs]
    table = pad_sequence(rows, batch_first=True)
    slot = torch.arange(table.shape[1])[None, :]
    n_obs = torch.tensor([len(o) for o, _ in pairs])[:, None]
    n_all = torch.tensor([len(r) for r in rows])[:, None]
    return table, slot < n_obs, (slot >= n_obs) & (slot < n_all)


@torch.no_grad()
def run_bpd_evaluation(model, diffusion, batch, clip_denoised, obs_indices, lat_indices, t_seq=None, observed_frames="x_0"):
    dev = model.device
    table, observed, latent = _window_table(obs_indices, lat_indices, batch.shape[0])
    n_slots = table.shape[1]
    if n_slots > batch.shape[1]:
        raise ValueError(f"a window of {n_slots} frames does not fit videos of {batch.shape[1]}")
    item = torch.arange(batch.shape[0])[:, None]
    used = (observed | latent).to(batch.dtype)[:, :, None, None, None]
    x0 = (batch[item, table.to(batch.device)] * used.to(batch.device)).to(dev)
    as_mask = lambda m: m.to(device=dev, dtype=x0.dtype)[:, :, None, None, None]
    lat_mask = as_mask(latent)
    model_kwargs = dict(frame_indices=table.to(dev), x0=x0, obs_mask=as_mask(observed), latent_mask=lat_mask,
                        kinda_marg_mask=torch.zeros_like(lat_mask), observed_frames=observed_frames, x_t_minus_1=x0)
    per_t = diffusion.calc_bpd_loop_subsampled(model, x0, clip_denoised=clip_denoised, model_kwargs=model_kwargs,
                                               latent_mask=lat_mask, t_seq=t_seq)
    out = {}
    for name, v in per_t.items():
        total = v.sum(dim=1) if v.ndim > 1 else v
        out[name] = (total * n_slots).detach().cpu().numpy()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The job around it: scripts/video_nll.py:31-140,187-352 -- which (observed, latent) index lists every test video is scored on, one
# pickle of metrics per video under <eval_dir>/elbos/, finished videos skipped before any network call.
def get_eval_frame_indices(args, batch=None, optimal_schedule_path=None):
    """(obs_indices, lat_indices), each [dataset item][window] -> frame list (video_nll.py:31-84): loaded from `args.indices_path` for an
    `inference_mode` that is not a registered strategy; else every window of the strategy's schedule -- the same lists for every item, or one
    per batch item for `adaptive-*` (which looks at the videos).  Non-adaptive lists are saved to `indices_path` on first use and CHECKED
    against it afterwards, so that the workers of one job score the same windows."""
    import os
    import time
    from . import inference_util
    if args.inference_mode not in inference_util.inference_strategies:
        return torch.load(args.indices_path)
    adaptive = "adaptive" in args.inference_mode
    schedule = inference_util.inference_strategies[args.inference_mode](
        video_length=args.T, num_obs=args.obs_length, max_frames=args.max_frames, step_size=args.step_size,
        optimal_schedule_path=optimal_schedule_path, **(dict(distance=getattr(args, "adaptive_distance", "lpips")) if adaptive else {}))
    if adaptive:
        schedule.set_videos(batch)
    windows = list(schedule)
    if adaptive:                                   # [window][side][item] -> [item][window]
        n = len(batch)
        return ([[w[0][j] for w in windows] for j in range(n)], [[w[1][j] for w in windows] for j in range(n)])
    per_item = ([w[0] for w in windows], [w[1] for w in windows])
    obs_indices, lat_indices = ([per_item[0]] * args.test_set_size, [per_item[1]] * args.test_set_size)
    if os.path.exists(args.indices_path):
        try:
            saved = torch.load(args.indices_path)
        except EOFError:                           # another worker is still writing it
            time.sleep(5)
            saved = torch.load(args.indices_path)
        if list(saved[0]) != obs_indices or list(saved[1]) != lat_indices:
            raise AssertionError(f"frame indices differ from the ones saved at {args.indices_path}")
    else:
        torch.save((obs_indices, lat_indices), args.indices_path)
    return obs_indices, lat_indices


def run(args, create=None, device=None):
    """Body of scripts/video_nll.py:87-140,262-352 on the engine: load + share the weights, name the run directory
    (test_util.py:65-132), deal the batches of the selected dataset items to the ranks, and per batch: skip it if every
    `elbos/elbo_<dataset index><postfix>.pkl` exists, else score every window type (`run_bpd_evaluation`) and pickle
    {metric: array over windows} per video.  Videos come from --videos / --synthetic as in video_sample."""
    import json
    import os
    import pickle
    from pathlib import Path
    import numpy as np
    from . import dist as vdist
    from . import test_util
    from .video_sample import load_model, open_videos
    rank, local_rank, world = vdist.init(device_index=device.index if device is not None and device.type == "cuda" else None)
    if device is None:
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    torch.manual_seed(getattr(args, "seed", 0) + rank)
    adaptive = args.adaptive = "adaptive" in args.inference_mode
    model, diffusion = load_model(args, device, rank, world, create=create)
    if args.max_frames is None:                                         # video_nll.py:288-290: BEFORE the run directory is named
        args.max_frames = model.config.get("max_frames") or model.config["T"]
    run_id = test_util.get_eval_run_identifier(args)                    # :293-296: named here -- an unset --T reads 'None', it is resolved below (:301-302)
    dataset = open_videos(args)
    if args.T is None:
        args.T = int(dataset[0][0].shape[0])
    args.test_set_size = len(dataset)
    out_dir = None
    if rank == 0:
        if args.eval_dir is None and not args.checkpoint_path:
            args.eval_dir = "results/synthetic"
        out_dir = test_util.get_model_results_path(args) / run_id
        os.makedirs(out_dir / "elbos", exist_ok=True)
        json_path = out_dir / "model_config.json"
        if not json_path.exists():
            with test_util.Protect(json_path):
                with open(json_path, "w") as f:
                    json.dump(model.config, f, indent=4)
    out_dir = Path(vdist.broadcast_object(out_dir, src=0))
    # the items of this job (video_nll.py:307-318): one batch-sized block for a task id (--task_id or SLURM_ARRAY_TASK_ID), else everything;
    # an explicit --indices list is refused there too
    task_id = getattr(args, "task_id", None)
    if task_id is None and "SLURM_ARRAY_TASK_ID" in os.environ:
        task_id = int(os.environ["SLURM_ARRAY_TASK_ID"])
    if getattr(args, "indices", None) is not None:
        raise NotImplementedError("video_nll: --indices (the reference raises here too: use --task_id)")
    args.indices = list(range(task_id * args.batch_size, (task_id + 1) * args.batch_size)) if task_id is not None else list(range(len(dataset)))
    if getattr(args, "indices_path", None) is None:
        args.indices_path = out_dir / "frame_indices.pt"
    postfix = ("_ddim" if args.use_ddim else "") + (f"_respace{args.timestep_respacing}" if args.timestep_respacing != "" else "")
    optimal_schedule_path = None if getattr(args, "optimality", None) is None else out_dir / "optimal_schedule.pt"
    batches = [args.indices[k:k + args.batch_size] for k in range(0, len(args.indices), args.batch_size)]
    shared = None
    for task in vdist.task_ids(len(batches), rank, world):
        ids = batches[task]
        names = [out_dir / "elbos" / f"elbo_{i}{postfix}.pkl" for i in ids]
        if all(p.exists() for p in names):
            print("Already exist. Skipping", names)
            continue
        batch = torch.stack([dataset[i][0] for i in ids])[:, :args.T]
        if adaptive:
            obs_b, lat_b = get_eval_frame_indices(args, batch=batch, optimal_schedule_path=optimal_schedule_path)
        else:
            if shared is None:
                shared = get_eval_frame_indices(args, optimal_schedule_path=optimal_schedule_path)
            obs_b, lat_b = [shared[0][i] for i in ids], [shared[1][i] for i in ids]
        per_window = [run_bpd_evaluation(model, diffusion, batch, args.clip_denoised, [o[w] for o in obs_b], [l[w] for l in lat_b])
                      for w in range(len(obs_b[0]))]
        stacked = {k: np.stack([r[k] for r in per_window], axis=1) for k in per_window[0]}
        for j, path in enumerate(names):
            with open(path, "wb") as f:
                pickle.dump({k: v[j] for k, v in stacked.items()}, f)
            print("Saved to", path)
    vdist.barrier()
    return out_dir


def main(argv=None):
    import argparse
    from .script_util import str2bool
    from .video_sample import add_job_arguments
    ap = add_job_arguments(argparse.ArgumentParser())
    ap.add_argument("--indices_path", default=None, help="saved (obs_indices, lat_indices); default <eval_dir>/frame_indices.pt")
    ap.add_argument("--clip_denoised", type=str2bool, default=True)
    ap.add_argument("--adaptive_distance", default="l2", choices=["l2", "lpips"])
    return run(ap.parse_args(argv))


if __name__ == "__main__":
    main()
