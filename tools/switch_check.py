#!/usr/bin/env python3
"""SHA-1 of the network output under the process' A/B switches (VD_NO_RPE_ALL, VD_NO_CONV_ACT, VD_HEAD_GENERIC, ...: read once per process, so a
test compares child processes).  Prints one JSON line: {"tiny": sha, "full64": sha, "version": ...}.
  tiny    tests/golden/unet_tiny.npz case 0 (32 base channels: generic kernels, RPE nets)
  full64  the default 116 M model, one 16-frame 64 x 64 clip, seeded (conv_wino_z128.hip's activating form serves its 64 x 64 and 32 x 32 levels)"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import video_diffusion_amd as vda  # noqa: E402


def sd_of(model):
    return {n: torch.from_numpy(vda.weights_init.synth_param(n, s)) for n, s in model.param_specs()}


def eps_sha(cfg, x, t, kw):
    model, diff = vda.create_video_model_and_diffusion(**{k: cfg[k] for k in vda.video_model_and_diffusion_defaults()})
    model.load_state_dict(sd_of(model))
    model.to("cuda").eval()
    eps, _ = diff._wrap_model(model)(x.cuda(), t.cuda(), **{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()})
    torch.cuda.synchronize()
    model.check_device_errors()
    return hashlib.sha1(eps.cpu().numpy().tobytes()).hexdigest()


def main():
    rec = dict(np.load(os.path.join(ROOT, "tests", "golden", "unet_tiny.npz"), allow_pickle=False))
    cfg = json.loads(str(rec["cfg_json"]))
    c = {k[3:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("c0_") and k != "c0_observed_frames"}
    kw = dict(frame_indices=c["frame_indices"], x0=c["x0"], obs_mask=c["obs_mask"], latent_mask=c["latent_mask"], kinda_marg_mask=c["kinda_marg_mask"],
              x_t_minus_1=c["x0"], observed_frames="x_0")
    out = {"tiny": eps_sha(cfg, c["x"], c["t"], kw)}
    cfg = {**vda.video_model_and_diffusion_defaults(), **dict(T=16, image_size=64, rp_alpha=16, rp_beta=16, rp_gamma=16, timestep_respacing="ddim250")}
    g = torch.Generator().manual_seed(9)
    B, T, S, n_obs = 1, 16, 64, 4
    x0 = torch.rand(B, T, 3, S, S, generator=g) * 2 - 1
    x0[:, n_obs:] = 0
    x = torch.randn(B, T, 3, S, S, generator=g)
    obs = torch.zeros(B, T, 1, 1, 1)
    obs[:, :n_obs] = 1
    kw = dict(frame_indices=torch.arange(T).view(1, T), x0=x0, obs_mask=obs, latent_mask=1 - obs, kinda_marg_mask=torch.zeros(B, T, 1, 1, 1), x_t_minus_1=x0,
              observed_frames="x_0")
    out["full64"] = eps_sha(cfg, x, torch.tensor([200]), kw)
    out["version"] = vda._lib.lib().vd_version().decode()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
