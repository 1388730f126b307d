"""Windowed conditional sampling: the caller of the hot path (scripts/video_sample.py:31-47, 50-190,
266-271 of the reference), driving the HIP engine.

    python -m video_diffusion_amd.video_sample --synthetic --inference_mode autoreg --T 32 \
        --obs_length 4 --max_frames 10 --step_size 2 --timestep_respacing ddim50 --out_dir /tmp/samples

`infer_video` keeps the reference's contract: `batch` (B, T, C, H, W) in [-1, 1]; the first
`obs_length` frames are observed; each window starts from `x0.clone()` (observed frames + whatever
the latent slots currently hold, zeros at first -- SURVEY F5), walks every respaced timestep through
`diffusion.p_sample`, and writes the last `n_latent` frames back.  One H2D transfer of the window's
inputs and one D2H of its result per window; nothing crosses the host inside the step loop.
No datasets or checkpoints ship with the reference: the CLI reads its test videos from a `.npy` file
(`--videos`) or makes synthetic ones (`--synthetic`), and takes either a checkpoint given on the command line or the
closed-form weights.  The job around `infer_video` -- which videos (`--indices`, `--task_id`, `--subset_size`), how many
samples of each (`--num_samples`, `--sample_idx`), what is already on disk and therefore skipped BEFORE any denoise
step runs -- follows scripts/video_sample.py:192-239,570-640; `run()` is that body.
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
                adaptive_distance="lpips", prefix_cache=False, suffix_skip=False, save_all_timesteps=False):
    """video_sample.py:50-190.  Returns (samples ndarray (B,T,C,H,W), all_timestep_samples): the second is the
    (B, num_timesteps, T, C, H, W) record of every step's output when `save_all_timesteps` (the reference's
    `args.save_all_timesteps`, :84-91,168-186; eager executor only -- the graph keeps a window on the device), else
    the reference's one-element placeholder.

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
    if save_all_timesteps:
        all_timestep_samples = torch.zeros([B, diffusion.num_timesteps, T, C, H, W])
        all_timestep_samples[:, :, :obs_length] = samples[:, :obs_length].unsqueeze(1)
    else:
        all_timestep_samples = torch.zeros([1])
    use_graph = (executor == "graph" and observed_frames in ("x_0", "x_t", "x_t_minus_1") and not use_gradient_method
                 and not save_all_timesteps)
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

        def write_back(local, every_step=None):
            if adaptive:
                for i, li in enumerate(latent_frame_indices):
                    samples[i, li] = local[i, n_obs_w:].cpu()
                    if every_step is not None:
                        all_timestep_samples[i, :, li] = every_step[i, :, n_obs_w:].cpu()
            else:
                samples[:, latent_frame_indices] = local[:, -n_latent:].cpu()
                if every_step is not None:
                    all_timestep_samples[:, :, latent_frame_indices] = every_step[:, :, -n_latent:].cpu()
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
        trace = [] if save_all_timesteps else None
        for timestep in timesteps:
            if sampler == "p_sample":
                local_samples = diffusion.p_sample(model, local_samples, t=t_tensors[timestep], clip_denoised=True,
                                                   model_kwargs=model_kwargs, return_attn_weights=False,
                                                   use_gradient_method=use_gradient_method)["sample"]
            else:
                local_samples = diffusion.ddim_sample(model, local_samples, t=t_tensors[timestep], clip_denoised=True,
                                                      model_kwargs=model_kwargs, eta=eta)["sample"]
            if trace is not None:
                trace.append(local_samples.clone())
        write_back(local_samples, None if trace is None else torch.stack(trace, dim=1))
        model.check_device_errors()    # once per window (write_back has synchronised): a non-finite network output of any of its steps raises here
    return samples.numpy(), all_timestep_samples.numpy()


def to_uint8(recon):
    """video_sample.py:266-268: ((x+1)/2*255) truncated to uint8."""
    return ((recon - drange[0]) / (drange[1] - drange[0]) * 255).astype(np.uint8)


class SyntheticVideos:
    """Stand-in for `get_test_dataset` (image_datasets.py; no dataset ships offline): item i is a U[-1, 1] video drawn
    from a generator seeded by i alone, so a video does not depend on how the job was cut into batches or ranks."""

    def __init__(self, n, T, size, channels=3):
        self.n, self.shape = int(n), (int(T), channels, int(size), int(size))

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise IndexError(i)
        g = torch.Generator().manual_seed(1234 + int(i))
        return torch.rand(*self.shape, generator=g) * 2 - 1, {}


class ArrayVideos:
    """`--videos file.npy`: (N, T, 3, H, W); float arrays are taken as already in [-1, 1] (what the reference's datasets
    hand over), uint8 arrays -- the format this tool writes -- are mapped back by x / 255 * 2 - 1."""

    def __init__(self, path):
        a = np.load(path, mmap_mode="r")
        if a.ndim != 5:
            raise ValueError(f"{path}: expected (N, T, C, H, W), got {a.shape}")
        self.a = a

    def __len__(self):
        return self.a.shape[0]

    def __getitem__(self, i):
        v = np.array(self.a[i])
        if v.dtype == np.uint8:
            return torch.from_numpy(v).float() / 255 * (drange[1] - drange[0]) + drange[0], {}
        return torch.from_numpy(v).float(), {}


def open_videos(args):
    path = getattr(args, "videos", None)
    if path:
        return ArrayVideos(path)
    if not getattr(args, "synthetic", True):
        raise ValueError("no dataset ships with this tool: give --videos <file.npy> or --synthetic")
    return SyntheticVideos(args.num_videos, args.T if args.T is not None else 16, args.image_size)


def resolve_indices(args, dataset_len):
    """Which dataset items the job covers (video_sample.py:570-590): --indices as given; else --task_id -> one batch worth
    of consecutive items (NOT clipped to the dataset: `Subset` then fails on the first missing item, as there); else the
    first --subset_size items; else everything.  --task_id refuses --subset_size as the reference asserts."""
    indices, task_id = getattr(args, "indices", None), getattr(args, "task_id", None)
    subset_size = getattr(args, "subset_size", None)
    if indices is None and task_id is not None:
        assert subset_size is None
        logger.info(f"Only generating predictions for the batch #{task_id}.")
        return list(range(task_id * args.batch_size, (task_id + 1) * args.batch_size))
    if subset_size is not None:
        logger.info(f"Only generating predictions for the first {subset_size} videos of the dataset.")
        return list(range(subset_size))
    if indices is None:
        logger.info("Generating predictions for the whole dataset.")
        return list(range(dataset_len))
    return [int(i) for i in indices]


def sample_names(out_dir, dataset_ids, sample_idx, prefix="sample"):
    """samples/<prefix>_%04d-%d.npy for each dataset item of a batch (video_sample.py:209-230)."""
    return [out_dir / "samples" / f"{prefix}_{i:04d}-{sample_idx}.npy" for i in dataset_ids]


def q_sample_every_timestep(diffusion, model, batch):
    """video_sample.py:245-254: the clean videos noised to every timestep, (B, num_timesteps, T, C, H, W)."""
    out = [diffusion.q_sample(batch, t=torch.full((batch.shape[0],), ts, dtype=torch.long), model=model).cpu()
           for ts in range(diffusion.num_timesteps)]
    return torch.stack(out, dim=1).numpy()


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
            mf = args.max_frames if args.max_frames is not None else 10   # video_train.py:164's default
            cfg = dict(defaults, T=mf, max_frames=mf, image_size=args.image_size, num_channels=args.num_channels,
                       num_res_blocks=args.num_res_blocks, rp_alpha=mf, rp_beta=mf, rp_gamma=mf)
        cfg.update(use_ddim=bool(getattr(args, "use_ddim", False)), timestep_respacing=args.timestep_respacing)   # video_sample.py:551-554
        if getattr(args, "override_dataset", None) is not None:
            cfg["dataset"] = args.override_dataset                                                                # :555-556
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




def add_job_arguments(ap):
    """The options the two sampling CLIs share; names, defaults and meaning of scripts/video_sample.py:405-528 where the
    reference has the option (its dataset options are replaced by --videos / --synthetic)."""
    ap.add_argument("checkpoint_path", nargs="?", default="")
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--eval_dir", default=None,
                    help="results directory; default: derived from the checkpoint path and the sampling options "
                         "(test_util.get_model_results_path), 'results/synthetic' without a checkpoint")
    ap.add_argument("--out_dir", default=None, help="alias of --eval_dir (earlier rounds' flag)")
    ap.add_argument("--videos", default=None, help=".npy file of test videos (N, T, 3, H, W): float in [-1, 1] or uint8")
    ap.add_argument("--synthetic", type=str2bool, nargs="?", const=True, default=True,
                    help="without --videos: --num_videos synthetic U[-1, 1] videos, item i seeded by i")
    ap.add_argument("--num_videos", type=int, default=2, help="size of the synthetic dataset")
    ap.add_argument("--dataset_partition", default="test", choices=["train", "test", "variable_length"],
                    help="names the run directory as the reference does ('trainset_' prefix, 'variable_length/' subdirectory); the videos themselves "
                         "come from --videos / --synthetic")
    ap.add_argument("--override_dataset", default=None, help="'<name>_' prefix of the run directory and `dataset` of model_config.json (video_sample.py:555-556)")
    ap.add_argument("--use_gradient_method", action="store_true")
    ap.add_argument("--inference_mode", default="autoreg", choices=sorted(inference_util.inference_strategies))
    ap.add_argument("--max_frames", type=int, default=None,
                    help="frames (observed or latent) per window; defaults to what the model was trained with")
    ap.add_argument("--obs_length", type=int, default=36)
    ap.add_argument("--step_size", type=int, default=1)
    ap.add_argument("--indices", type=int, nargs="*", default=None, help="only these dataset items")
    ap.add_argument("--use_ddim", type=str2bool, nargs="?", const=True, default=False)
    ap.add_argument("--timestep_respacing", default="")
    ap.add_argument("--T", type=int, default=None, help="video length; default: the dataset's (16 for --synthetic)")
    ap.add_argument("--subset_size", type=int, default=None, help="only the first N dataset items")
    ap.add_argument("--num_samples", type=int, default=1, help="samples per test video")
    ap.add_argument("--sample_idx", type=int, default=None,
                    help="write exactly this sample index (--num_samples is then ignored)")
    ap.add_argument("--task_id", type=int, default=None,
                    help="only the batch-sized block of dataset items #task_id (composes with the rank shard)")
    ap.add_argument("--optimality", default=None,
                    choices=["linspace-t", "random-t", "linspace-t-force-nearby", "random-t-force-nearby"],
                    help="read <eval_dir>/optimal_schedule.pt (made by the reference's video_optimal_schedule.py)")
    ap.add_argument("--observed_frames", default="x_0", choices=["x_0", "x_t", "x_t_minus_1"])
    ap.add_argument("--save_all_timesteps", action="store_true")
    ap.add_argument("--image_size", type=int, default=64, help="without a checkpoint: the closed-form model's size")
    ap.add_argument("--num_channels", type=int, default=128)
    ap.add_argument("--num_res_blocks", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    return ap


def main(argv=None):
    ap = add_job_arguments(argparse.ArgumentParser())
    ap.add_argument("--adaptive_distance", default="l2", choices=["l2", "lpips"],
                    help="adaptive-* modes: frame embedding for the farthest-point selection (lpips needs set_lpips_embedder)")
    ap.add_argument("--executor", default="eager", choices=["graph", "eager"],
                    help="eager: one p_sample call per step (default); graph: one captured hipGraph per window shape (executor.py)")
    ap.add_argument("--suffix_skip", type=str2bool, nargs="?", const=True, default=False,
                    help="with --executor graph and observed_frames x_0 / x_t_minus_1: run the network behind its last attention layer "
                         "on the non-observed frames only")
    ap.add_argument("--prefix_cache", type=str2bool, nargs="?", const=True, default=False,
                    help="with --executor graph and observed_frames x_0: compute the observed frames' encoder prefix once per window")
    args = ap.parse_args(argv)
    return run(args)


def _default_infer(args, model, diffusion, batch, optimal_schedule_path):
    return infer_video(args.inference_mode, model, diffusion, batch, args.max_frames, args.obs_length, args.step_size,
                       optimal_schedule_path, use_gradient_method=getattr(args, "use_gradient_method", False),
                       observed_frames=args.observed_frames, executor=getattr(args, "executor", "eager"),
                       adaptive_distance=getattr(args, "adaptive_distance", "l2"),
                       prefix_cache=getattr(args, "prefix_cache", False), suffix_skip=getattr(args, "suffix_skip", False),
                       save_all_timesteps=getattr(args, "save_all_timesteps", False))


def run(args, create=None, device=None, infer=None):
    """The body of the reference's script (video_sample.py:192-298, 530-640): join the job, load + share the weights,
    decide which dataset items this job covers, and for every batch this rank owns and every sample index: form the
    output names, look at the disk, and run `infer_video` only if a file is missing (`todo`, :231-239) -- a resumed
    job repeats no denoise step.  Files: samples/sample_%04d-%d.npy (+ all_timestep_sample_ / q_sample_ / error_ with
    --save_all_timesteps) under results/<...>/<run id>/ (test_util.py:65-132).  `create` / `device` let the CPU tests
    drive the sharding + broadcast path with a stand-in engine; `infer(args, model, diffusion, batch, schedule_path)`
    is the sampler (video_sample_full passes its own)."""
    import json
    from pathlib import Path
    from . import test_util
    logging.basicConfig(level=logging.INFO)
    from . import dist as vdist
    # the process group is bound to the device this job computes on (an explicit `device` wins over LOCAL_RANK)
    rank, local_rank, world = vdist.init(device_index=device.index if device is not None and device.type == "cuda" else None)
    if device is None:
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    torch.manual_seed(args.seed + rank)
    infer = infer or _default_infer
    # the run identifier is formed from the options AS GIVEN, before --max_frames / --T take their defaults from the model and
    # the dataset (video_sample.py:530-533 precedes :568-570,612-615: an unset one reads 'None' in the directory name)
    run_id = test_util.get_eval_run_identifier(args)
    model, diffusion = load_model(args, device, rank, world, create=create)
    if args.max_frames is None:                                            # video_sample.py:568-570
        args.max_frames = model.config.get("max_frames") or model.config["T"]
    dataset = open_videos(args)
    if args.T is None:                                                     # :612-615
        args.T = int(dataset[0][0].shape[0])
    # results/<checkpoint subpath>/<stem>[_<step>][_ddim][_respace<X>]/<mode>_<max_frames>_<step_size>_<T>_<obs_length>/
    # (test_util.py:65-132 of the reference, video_sample.py:530-533,600-611): what video_eval.py reads
    # -- derived on rank 0 (a '*latest' checkpoint is opened once more there for its step) and sent to the others
    out_dir = None
    if rank == 0:
        if args.eval_dir is None:
            alias = getattr(args, "out_dir", None)
            args.eval_dir = alias if alias is not None else (None if args.checkpoint_path else "results/synthetic")
        out_dir = test_util.get_model_results_path(args) / run_id
        if getattr(args, "dataset_partition", None) == "variable_length":                                       # video_sample.py:603-604
            out_dir = out_dir / "variable_length"
        os.makedirs(out_dir / "samples", exist_ok=True)
        json_path = out_dir / "model_config.json"                         # video_sample.py:620-626
        if not json_path.exists():
            with test_util.Protect(json_path):
                with open(json_path, "w") as f:
                    json.dump(model.config, f, indent=4)
    out_dir = Path(vdist.broadcast_object(out_dir, src=0))
    optimal_schedule_path = None if getattr(args, "optimality", None) is None else out_dir / "optimal_schedule.pt"   # :199-200
    args.indices = resolve_indices(args, len(dataset))
    batches = [args.indices[k:k + args.batch_size] for k in range(0, len(args.indices), args.batch_size)]   # DataLoader(Subset(..)), no shuffle
    one_idx = getattr(args, "sample_idx", None)
    sample_ids = range(getattr(args, "num_samples", 1)) if one_idx is None else [one_idx]
    every = getattr(args, "save_all_timesteps", False)
    for task in vdist.task_ids(len(batches), rank, world):                  # one process per GPU: rank r owns batches r, r+R, ...
        ids = batches[task]
        batch = None
        for sample_idx in sample_ids:
            names = sample_names(out_dir, ids, sample_idx)
            todo = [not p.exists() for p in names]
            if not any(todo):
                logger.info(f"Nothing to do for the videos {ids[0]} - {ids[-1]}, sample #{sample_idx}.")
                continue
            if batch is None:
                batch = torch.stack([dataset[i][0] for i in ids])[:, :args.T].to(device)
            q_all = q_sample_every_timestep(diffusion, model, batch) if every else None
            recon, recon_all = infer(args, model, diffusion, batch, optimal_schedule_path)
            outputs = [(names, to_uint8(recon))]
            if every:
                outputs += [(sample_names(out_dir, ids, sample_idx, "q_sample"), q_all),
                            (sample_names(out_dir, ids, sample_idx, "error"), q_all - recon_all),
                            (sample_names(out_dir, ids, sample_idx, "all_timestep_sample"), to_uint8(recon_all))]
            for paths, arrays in outputs:
                for i, path in enumerate(paths):
                    if todo[i]:
                        np.save(path, arrays[i])
                        logger.info(f"*** Saved {path} ***")
                    else:
                        logger.info(f"Skipped {path}")
    vdist.barrier()
    return out_dir


if __name__ == "__main__":
    main()
