#!/usr/bin/env python3
"""Round-5 golden vectors from the IMPORTED reference (build container only; the reference never travels).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 /root/repo/tools/gen_golden_r5.py

  tests/golden/nll_xstart_tiny.npz    the NLL path with predict_xstart=True (ModelMeanType.START_X): _vb_terms_bpd
                                      (gaussian_diffusion.py:750-790) at t = 4, 2, 0, clip on / off, masked and unmasked, and
                                      calc_bpd_loop_subsampled (:928-1002) from a seeded global generator (draw order pinned)
  tests/golden/attn_denoised_tiny.npz return_attn_weights TOGETHER with denoised_fn (:274-324 allows it): p_sample and
                                      p_mean_variance dicts + the per-block head-averaged attention maps
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")

spec = importlib.util.spec_from_file_location("weights_init", os.path.join(REPO, "video-diffusion_amd", "weights_init.py"))
weights_init = importlib.util.module_from_spec(spec)
spec.loader.exec_module(weights_init)

lp = types.ModuleType("lpips")
lp.LPIPS = type("LPIPS", (torch.nn.Module,), {})
lp.normalize_tensor = lambda x: x
sys.modules["lpips"] = lp

from improved_diffusion import script_util as su  # noqa: E402

torch.set_num_threads(8)


def tiny_cfg(**over):
    d = su.video_model_and_diffusion_defaults()
    d.update(T=4, image_size=32, num_channels=32, num_res_blocks=1, rp_alpha=4, rp_beta=4, rp_gamma=4, timestep_respacing="ddim5")
    d.update(over)
    return d


def build(cfg):
    model, diff = su.create_video_model_and_diffusion(**cfg)
    sd = model.state_dict()
    model.load_state_dict({k: torch.from_numpy(weights_init.synth_param(k, tuple(v.shape))) for k, v in sd.items()})
    model.eval()
    return model, diff


def make_inputs(B, T, S, n_obs, seed, fidx_rows):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(B, T, 3, S, S, generator=g) * 2 - 1
    x = torch.randn(B, T, 3, S, S, generator=g)
    noise = torch.randn(B, T, 3, S, S, generator=g)
    obs = torch.zeros(B, T, 1, 1, 1)
    obs[:, :n_obs] = 1
    return dict(x=x, x0=x0, noise=noise, obs_mask=obs, latent_mask=1 - obs, kinda_marg_mask=torch.zeros(B, T, 1, 1, 1),
                frame_indices=torch.tensor(fidx_rows, dtype=torch.int64))


def kwargs_of(inp):
    return dict(frame_indices=inp["frame_indices"], x0=inp["x0"], obs_mask=inp["obs_mask"], latent_mask=inp["latent_mask"],
                kinda_marg_mask=inp["kinda_marg_mask"], x_t_minus_1=inp["x0"], observed_frames="x_0")


class FixedNoise:
    def __init__(self, *tensors):
        self.q = list(tensors)

    def __enter__(self):
        self.real = torch.randn_like
        torch.randn_like = lambda x, **k: self.q.pop(0).clone()
        return self

    def __exit__(self, *exc):
        torch.randn_like = self.real


def denoised_fn(x):
    return 1.3 * torch.tanh(1.5 * x) + 0.05


def gen_nll_xstart():
    cfg = tiny_cfg(predict_xstart=True)
    model, diff = build(cfg)
    inp = make_inputs(2, 4, 32, 2, seed=52, fidx_rows=[[0, 1, 2, 3], [2, 3, 6, 7]])
    x0 = inp["x0"]
    B = x0.shape[0]
    rec = dict(cfg_json=np.array(json.dumps(cfg)), **{k: v.numpy() for k, v in inp.items()})
    kw = kwargs_of(inp)
    with torch.no_grad():
        for tv in (4, 2, 0):
            t = torch.tensor([tv] * B)
            x_t = diff.q_sample(x0, t, noise=inp["noise"])
            rec[f"t{tv}_x_t"] = x_t.numpy()
            for clip in (True, False):
                vb = diff._vb_terms_bpd(model, x_start=x0, x_t=x_t, t=t, clip_denoised=clip, model_kwargs=dict(kw), latent_mask=inp["latent_mask"])
                rec[f"t{tv}_vb_clip{int(clip)}"] = vb["output"].numpy()
                rec[f"t{tv}_pred_xstart_clip{int(clip)}"] = vb["pred_xstart"].numpy()
            rec[f"t{tv}_vb_nomask"] = diff._vb_terms_bpd(model, x_start=x0, x_t=x_t, t=t, clip_denoised=True, model_kwargs=dict(kw))["output"].numpy()
        torch.manual_seed(311)
        m = diff.calc_bpd_loop_subsampled(model, x0, clip_denoised=True, model_kwargs=dict(kw), latent_mask=inp["latent_mask"])
        rec["bpd_seed"] = np.array(311)
        for k, v in m.items():
            rec[f"bpd_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "nll_xstart_tiny.npz"), **rec)


def gen_attn_denoised():
    cfg = tiny_cfg(timestep_respacing="ddim250")
    model, diff = build(cfg)
    inp = make_inputs(2, 4, 32, 2, seed=53, fidx_rows=[[0, 1, 2, 3], [5, 6, 9, 12]])
    inp["x0"][:, 2:] = 0
    rec = dict(cfg_json=np.array(json.dumps(cfg)), **{k: v.numpy() for k, v in inp.items()})
    with torch.no_grad():
        t = torch.tensor([120, 120])
        with FixedNoise(inp["noise"]):
            o = diff.p_sample(model, inp["x"], t, clip_denoised=True, denoised_fn=denoised_fn, model_kwargs=kwargs_of(inp), return_attn_weights=True)
        rec["psample"], rec["pred_xstart"] = o["sample"].numpy(), o["pred_xstart"].numpy()
        for kind in ("temporal", "spatial"):
            rec[f"n_{kind}"] = np.array(len(o["attn"][kind]))
            for i, a in enumerate(o["attn"][kind]):
                rec[f"{kind}_{i}_shape"] = np.array(a.shape)
                rec[f"{kind}_{i}"] = (a[:, ::8] if a.shape[1] > 64 else a).numpy()
        pm = diff.p_mean_variance(model, inp["x"], t, clip_denoised=True, denoised_fn=denoised_fn, model_kwargs=kwargs_of(inp), return_attn_weights=True)
        rec["pmv_mean"] = pm["mean"].numpy()
        assert len(pm["attn"]["temporal"]) == len(o["attn"]["temporal"])
    np.savez_compressed(os.path.join(OUT, "attn_denoised_tiny.npz"), **rec)


if __name__ == "__main__":
    gen_nll_xstart()
    gen_attn_denoised()
    for f in ("nll_xstart_tiny.npz", "attn_denoised_tiny.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))
