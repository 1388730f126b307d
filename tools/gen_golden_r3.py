#!/usr/bin/env python3
"""Round-3 golden vectors from the IMPORTED reference (build container only; the reference never travels).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 /root/repo/tools/gen_golden_r3.py [what ...]

what (default: all):
  denoised   tests/golden/denoised_fn_tiny.npz   p_sample / ddim_sample / p_mean_variance with a `denoised_fn`
                                                 (gaussian_diffusion.py:319-324)
  grad       tests/golden/grad_tiny.npz          p_mean_variance / p_sample with use_gradient_method=True
                                                 (gaussian_diffusion.py:264-271,350-364): x.grad, mean, sample
  attn       tests/golden/attn_tiny.npz          return_attn_weights=True: the {'temporal': [...], 'spatial': [...]} lists
                                                 (unet.py:457-466,799-836)
  variants   tests/golden/variants_tiny.npz      cond_emb_type duplicate / all-initzero / t=0 and learn_sigma=True
                                                 (unet.py:932-947,1014-1019; gaussian_diffusion.py:277-298)
  full       tests/golden/unet_full64.npz, unet_full128.npz   eps of the DEFAULT 116 M (64x64, T=16) and 119 M (128x128,
                                                 T=8) models for one clip, plus reference-vs-oracle seconds per step (the
                                                 `cpu_baseline.kind: "port"` equivalence), printed and stored
"""
import importlib.util
import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

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
    d.update(T=4, image_size=32, num_channels=32, num_res_blocks=1, rp_alpha=4, rp_beta=4, rp_gamma=4,
             timestep_respacing="ddim250")
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
    x0[:, n_obs:] = 0
    x = torch.randn(B, T, 3, S, S, generator=g)
    noise = torch.randn(B, T, 3, S, S, generator=g)
    noise2 = torch.randn(B, T, 3, S, S, generator=g)
    obs = torch.zeros(B, T, 1, 1, 1)
    obs[:, :n_obs] = 1
    km = torch.zeros(B, T, 1, 1, 1)
    return dict(x=x, x0=x0, noise=noise, noise2=noise2, obs_mask=obs, latent_mask=1 - obs, kinda_marg_mask=km,
                frame_indices=torch.tensor(fidx_rows, dtype=torch.int64))


def kwargs_of(inp, observed_frames="x_0", xtm1=None):
    return dict(frame_indices=inp["frame_indices"], x0=inp["x0"], obs_mask=inp["obs_mask"], latent_mask=inp["latent_mask"],
                kinda_marg_mask=inp["kinda_marg_mask"], x_t_minus_1=inp["x0"] if xtm1 is None else xtm1,
                observed_frames=observed_frames)


class FixedNoise:
    """`th.randn_like` replaced by a queue of recorded tensors: the draw ORDER is part of the fixture."""

    def __init__(self, *tensors):
        self.q = list(tensors)

    def __enter__(self):
        self.real = torch.randn_like
        torch.randn_like = lambda x, **k: self.q.pop(0).clone()
        return self

    def __exit__(self, *exc):
        torch.randn_like = self.real


def denoised_fn(x):
    """An arbitrary but smooth map that leaves [-1, 1] sometimes, so that the clamp behind it matters."""
    return 1.3 * torch.tanh(1.5 * x) + 0.05


def gen_denoised():
    cfg = tiny_cfg()
    model, diff = build(cfg)
    inp = make_inputs(2, 4, 32, 2, 11, [[0, 1, 2, 3], [5, 6, 9, 12]])
    rec = dict(cfg_json=json.dumps(cfg), **{k: v.numpy() for k, v in inp.items()})
    with torch.no_grad():
        for t_val in [249, 120, 0]:
            t = torch.tensor([t_val] * 2)
            for clip in (True, False):
                tag = f"t{t_val}_clip{int(clip)}"
                with FixedNoise(inp["noise"]):
                    o = diff.p_sample(model, inp["x"], t, clip_denoised=clip, denoised_fn=denoised_fn, model_kwargs=kwargs_of(inp))
                rec[tag + "_psample"], rec[tag + "_pred_xstart"] = o["sample"].numpy(), o["pred_xstart"].numpy()
                pm = diff.p_mean_variance(model, inp["x"], t, clip_denoised=clip, denoised_fn=denoised_fn, model_kwargs=kwargs_of(inp))
                rec[tag + "_mean"] = pm["mean"].numpy()
                for eta in (0.0, 1.0):
                    with FixedNoise(inp["noise"]):
                        o = diff.ddim_sample(model, inp["x"], t, clip_denoised=clip, denoised_fn=denoised_fn,
                                             model_kwargs=kwargs_of(inp), eta=eta)
                    rec[tag + f"_ddim_eta{int(eta)}"] = o["sample"].numpy()
    np.savez_compressed(os.path.join(OUT, "denoised_fn_tiny.npz"), **rec)
    print("denoised_fn_tiny.npz written")


def gen_grad():
    """The guidance of gaussian_diffusion.py:264-271,350-364: all frames are fed as latent, a sample of x_{t-1} is drawn
    inside p_mean_variance, its squared distance to the observed frames' x_{t-1} is back-propagated to x_t, and the mean
    moves by -10 * alpha_t * grad / 2.  Cases: the tiny config (32 channels: the generic kernels), a 64-channel one (the
    Winograd / split-GEMM kernels take Cout % 64 == 0) and one without scale-shift norm / with the bucket table."""
    out = {}
    cases = [("c32", tiny_cfg(), 2, 4, 2, [[0, 1, 2, 3], [5, 6, 9, 12]]),
             ("c64", tiny_cfg(num_channels=64, T=6, rp_alpha=6, rp_beta=6, rp_gamma=6), 1, 6, 3, [[0, 1, 2, 3, 4, 5]]),
             ("c64tab", tiny_cfg(num_channels=64, use_rpe_net=False, use_scale_shift_norm=False), 1, 4, 1, [[0, 1, 2, 3]])]
    for name, cfg, B, T, n_obs, fidx in cases:
        model, diff = build(cfg)
        inp = make_inputs(B, T, 32, n_obs, 21 + len(name), fidx)
        xtm1 = inp["x0"] + 0.3 * inp["noise2"] * inp["obs_mask"]          # some "x_{t-1} of the observed frames"
        out[name + "_cfg_json"] = json.dumps(cfg)
        for k, v in inp.items():
            out[f"{name}_{k}"] = v.numpy()
        out[name + "_x_t_minus_1"] = xtm1.numpy()
        for t_val in [249, 100, 1, 0]:
            t = torch.tensor([t_val] * B)
            x = inp["x"].clone()
            with FixedNoise(inp["noise"]):
                pm = diff.p_mean_variance(model, x, t, clip_denoised=True, model_kwargs=kwargs_of(inp, xtm1=xtm1),
                                          use_gradient_method=True)
            tag = f"{name}_t{t_val}"
            out[tag + "_grad"] = x.grad.detach().numpy().copy()
            out[tag + "_mean"] = pm["mean"].detach().numpy()
            out[tag + "_pred_xstart"] = pm["pred_xstart"].detach().numpy()
            x = inp["x"].clone()
            with FixedNoise(inp["noise"], inp["noise2"]):                # p_mean_variance draws first, then p_sample
                o = diff.p_sample(model, x, t, clip_denoised=True, model_kwargs=kwargs_of(inp, xtm1=xtm1),
                                  use_gradient_method=True)
            out[tag + "_psample"] = o["sample"].detach().numpy()
            print(tag, "|grad| max", float(np.abs(out[tag + "_grad"]).max()), "mean shift max",
                  float((5 * np.abs(out[tag + "_grad"])).max()))
    np.savez_compressed(os.path.join(OUT, "grad_tiny.npz"), **out)
    print("grad_tiny.npz written")


def gen_attn():
    cfg = tiny_cfg()
    model, diff = build(cfg)
    inp = make_inputs(2, 4, 32, 2, 31, [[0, 1, 2, 3], [5, 6, 9, 12]])
    rec = dict(cfg_json=json.dumps(cfg), **{k: v.numpy() for k, v in inp.items()})
    with torch.no_grad():
        t = torch.tensor([100, 100])
        with FixedNoise(inp["noise"]):
            o = diff.p_sample(model, inp["x"], t, clip_denoised=True, model_kwargs=kwargs_of(inp), return_attn_weights=True)
        rec["psample"] = o["sample"].numpy()
        for kind in ("temporal", "spatial"):
            rec[f"n_{kind}"] = len(o["attn"][kind])
            for i, a in enumerate(o["attn"][kind]):
                rec[f"{kind}_{i}_shape"] = np.array(a.shape)
                # the 256 x 256 spatial maps are stored every 8th query row (2 MB each otherwise)
                rec[f"{kind}_{i}"] = (a[:, ::8] if a.shape[1] > 64 else a).numpy()
                print(kind, i, tuple(a.shape))
        pm = diff.p_mean_variance(model, inp["x"], t, model_kwargs=kwargs_of(inp), return_attn_weights=True)
        assert len(pm["attn"]["temporal"]) == len(o["attn"]["temporal"])
    np.savez_compressed(os.path.join(OUT, "attn_tiny.npz"), **rec)
    print("attn_tiny.npz written")


def gen_variants():
    """cond_emb_type in {duplicate, all-initzero, t=0} (unet.py:932-947,1014-1019) and learn_sigma=True (LEARNED_RANGE variance,
    gaussian_diffusion.py:277-298; script_util.py:129-131,424-428): eps at Boundary A, p_sample / ddim_sample / p_mean_variance."""
    rec = {}
    for name, over in [("dup", dict(cond_emb_type="duplicate")), ("allz", dict(cond_emb_type="all-initzero")), ("t0", dict(cond_emb_type="t=0")),
                       ("ls", dict(learn_sigma=True))]:
        cfg = tiny_cfg(**over)
        model, diff = build(cfg)
        inp = make_inputs(2, 4, 32, 2, 41 + len(name), [[0, 1, 2, 3], [5, 6, 9, 12]])
        inp["obs_mask"][1] = 0                       # batch item 1 has no observed frame ('t=0' writes -1 through an expanded tensor:
        inp["latent_mask"][1] = 1                    # a whole batch item gets it as soon as one of its frames is observed)
        inp["x0"][1] = 0
        rec[name + "_cfg_json"] = json.dumps(cfg)
        for k, v in inp.items():
            if k != "noise2":
                rec[f"{name}_{k}"] = v.numpy()
        with torch.no_grad():
            for t_val in [100, 0]:
                t = torch.tensor([t_val] * 2)
                tag = f"{name}_t{t_val}"
                out, _ = diff._wrap_model(model)(inp["x"], t, **kwargs_of(inp))
                rec[tag + "_out"] = out.numpy()
                if cfg["learn_sigma"]:
                    # the reference cannot sample with a learned variance on video tensors: its own assert fails
                    # (gaussian_diffusion.py:283, C = x.shape[1] = T); record that, it is the behaviour to mirror
                    try:
                        diff.p_sample(model, inp["x"], t, clip_denoised=True, model_kwargs=kwargs_of(inp))
                        rec[tag + "_psample_error"] = "none"
                    except AssertionError:
                        rec[tag + "_psample_error"] = "AssertionError"
                    continue
                with FixedNoise(inp["noise"]):
                    o = diff.p_sample(model, inp["x"], t, clip_denoised=True, model_kwargs=kwargs_of(inp))
                rec[tag + "_psample"], rec[tag + "_pred_xstart"] = o["sample"].numpy(), o["pred_xstart"].numpy()
                with FixedNoise(inp["noise"]):
                    o = diff.ddim_sample(model, inp["x"], t, clip_denoised=True, model_kwargs=kwargs_of(inp), eta=1.0)
                rec[tag + "_ddim_eta1"] = o["sample"].numpy()
        print(name, "out", rec[f"{name}_t100_out"].shape, "var_type", diff.model_var_type)
    np.savez_compressed(os.path.join(OUT, "variants_tiny.npz"), **rec)
    print("variants_tiny.npz written")


def gen_full():
    from oracle.unet_ref import UNetRef
    timing = {}
    for name, size, T, n_obs, seed in [("unet_full64.npz", 64, 16, 4, 9), ("unet_full128.npz", 128, 8, 4, 19)]:
        cfg = su.video_model_and_diffusion_defaults()
        cfg.update(T=T if size == 128 else 16, image_size=size, rp_alpha=16, rp_beta=16, rp_gamma=16, timestep_respacing="ddim250")
        if size == 128:
            cfg.update(T=16)
        model, diff = build(cfg)
        n_par = sum(p.numel() for p in model.parameters())
        # the same seeded window the GPU tests build (tests/test_gpu_engine.py::_rand_window)
        g = torch.Generator().manual_seed(seed)
        x0 = torch.rand(1, T, 3, size, size, generator=g) * 2 - 1
        x0[:, n_obs:] = 0
        x = torch.randn(1, T, 3, size, size, generator=g)
        obs = torch.zeros(1, T, 1, 1, 1)
        obs[:, :n_obs] = 1
        fidx = torch.arange(T, dtype=torch.int64).view(1, T)
        kw = dict(frame_indices=fidx, x0=x0, obs_mask=obs, latent_mask=1 - obs, kinda_marg_mask=torch.zeros(1, T, 1, 1, 1),
                  x_t_minus_1=x0, observed_frames="x_0")
        t = torch.tensor([200])
        wrapped = diff._wrap_model(model)
        with torch.no_grad():
            wrapped(x, t, **kw)                                              # warm-up
            t0 = time.time()
            eps, _ = wrapped(x, t, **kw)
            t_ref = time.time() - t0
            sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
            ora = UNetRef(cfg, sd)
            tm = torch.tensor([float(diff.timestep_map[200]) * (1000.0 / diff.original_num_steps)]) if diff.rescale_timesteps \
                else torch.tensor([float(diff.timestep_map[200])])
            ora(x, tm, **kw)
            t0 = time.time()
            eps_o = ora(x, tm, **kw)
            t_ora = time.time() - t0
        d = float((eps - eps_o).abs().max())
        timing[name] = dict(params=n_par, reference_s_per_step=t_ref, oracle_s_per_step=t_ora, max_abs_eps_diff=d, threads=8,
                            shape=[1, T, 3, size, size])
        print(name, timing[name])
        np.savez_compressed(os.path.join(OUT, name), eps=eps.numpy(), t=np.array([200]), seed=np.array([seed]), n_obs=np.array([n_obs]),
                            T=np.array([T]), x_checksum=np.array([float(x.double().sum()), float(x0.double().sum())]),
                            cfg_json=json.dumps(cfg), n_params=np.array([n_par]))
    json.dump(timing, open(os.path.join(OUT, "full_size_reference_vs_oracle.json"), "w"), indent=1)


if __name__ == "__main__":
    what = sys.argv[1:] or ["denoised", "grad", "attn", "variants", "full"]
    for w in what:
        {"denoised": gen_denoised, "grad": gen_grad, "attn": gen_attn, "variants": gen_variants, "full": gen_full}[w]()
