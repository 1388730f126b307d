#!/usr/bin/env python3
"""Throughput of the vertical / horizontal sampler (scripts/video_sample_full.py of the reference) through
`video_diffusion_amd.video_sample_full.infer_video`: denoise steps per second INCLUDING the host loop that re-assembles a
window per (timestep, window) pair, beside the raw step rate of the same window shape.  One JSON line.
    python tools/full_sampler_bench.py [--batch 8] [--T 32] [--max-frames 16] [--obs 4] [--step-size 12] [--respacing ddim20]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_diffusion_amd as vda  # noqa: E402
from video_diffusion_amd import inference_util, video_sample_full  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--T", type=int, default=28)
    ap.add_argument("--max-frames", type=int, default=16)
    ap.add_argument("--obs", type=int, default=4)
    ap.add_argument("--step-size", type=int, default=12)
    ap.add_argument("--respacing", default="ddim20")
    ap.add_argument("--vertical", type=int, default=0)
    args = ap.parse_args()
    cfg = vda.video_model_and_diffusion_defaults()
    cfg.update(T=args.max_frames, image_size=64, rp_alpha=args.max_frames, rp_beta=args.max_frames, rp_gamma=args.max_frames,
               timestep_respacing=args.respacing)
    model, diff = vda.create_video_model_and_diffusion(**cfg)
    model.load_state_dict({k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in model.param_specs()})
    model.to("cuda").eval()
    batch = (torch.rand(args.batch, args.T, 3, 64, 64, generator=torch.Generator().manual_seed(3)) * 2 - 1).cuda()
    windows = list(inference_util.inference_strategies["autoreg"](video_length=args.T, num_obs=args.obs, max_frames=args.max_frames,
                                                                 step_size=args.step_size, optimal_schedule_path=None))
    n_steps = len(windows) * diff.num_timesteps

    def run():
        torch.manual_seed(0)
        return video_sample_full.infer_video("autoreg", model, diff, batch, args.max_frames, args.obs, args.step_size,
                                             vertical_steps=args.vertical)

    run()                                                  # warm-up (workspace, kernel attributes)
    torch.cuda.synchronize()
    t0 = time.time()
    out, _ = run()
    torch.cuda.synchronize()
    dt = time.time() - t0
    # raw step rate of the largest window shape, no host loop
    Tw = max(len(o) + len(l) for o, l in windows)
    x = torch.randn(args.batch, Tw, 3, 64, 64, device="cuda")
    obs = torch.zeros(args.batch, Tw, 1, 1, 1, device="cuda"); obs[:, :args.obs] = 1
    kw = dict(frame_indices=torch.arange(Tw, device="cuda").view(1, Tw).repeat(args.batch, 1), x0=x, obs_mask=obs, latent_mask=1 - obs,
              kinda_marg_mask=torch.zeros_like(obs), x_t_minus_1=x, observed_frames="x_0")
    t = torch.tensor([5] * args.batch, device="cuda")
    for _ in range(3):
        diff.p_sample(model, x, t, model_kwargs=kw)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        diff.p_sample(model, x, t, model_kwargs=kw)
    torch.cuda.synchronize()
    raw = 10 / (time.time() - t0)
    print(json.dumps({"metric": "denoise-steps/sec through video_sample_full.infer_video (horizontal sampler, host loop included)",
                      "value": round(n_steps / dt, 3), "unit": "denoise-steps/sec", "steps": n_steps, "windows": len(windows),
                      "window_frames": [len(o) + len(l) for o, l in windows], "seconds": round(dt, 3),
                      "raw_p_sample_steps_per_sec_largest_window": round(raw, 3),
                      "config": {"workload": f"64x64 default model, B={args.batch}, video T={args.T}, autoreg max_frames={args.max_frames} "
                                             f"obs={args.obs} step_size={args.step_size}, {args.respacing}, vertical_steps={args.vertical}"},
                      "finite": bool(torch.isfinite(torch.from_numpy(out)).all())}))


if __name__ == "__main__":
    main()
