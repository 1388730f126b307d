#!/usr/bin/env python3
"""Error report of the DECLARED reduced mode VD_MATH=bf16x3 (three of the six bf16 piece products; never the default, never
the headline).  Run with that variable set (tests/test_gpu_ops.py and bench.py start it as a child process); one JSON line:
  linear / conv : max and mean |error| against an fp64 product, next to the same figures of the exact mode's bound
  eps_tiny      : max |eps - reference golden| on tests/golden/unet_tiny.npz (32 base channels, 32x32)
  eps_full64    : the same on the default 116 M model (tests/golden/unet_full64.npz)"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import video_diffusion_amd as vda  # noqa: E402
from video_diffusion_amd import _lib  # noqa: E402
from helpers import case_inputs, load_npz, n_cases, synth_sd  # noqa: E402


def main():
    assert os.environ.get("VD_MATH") == "bf16x3", "run with VD_MATH=bf16x3"
    L = _lib.lib()
    out = {"version": L.vd_version().decode()}
    g = torch.Generator().manual_seed(0)
    # ---- linear: M x K @ K x N
    M, K, N = 4096, 512, 384
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * (3.0 / K) ** 0.5; b = torch.randn(N, generator=g) * 0.1
    ws = torch.empty(3 * N * K, dtype=torch.int16)
    _lib.check(L.vd_pack_linear_split(_lib.ptr(w.contiguous()), _lib.ptr(ws), N, K))
    o = torch.empty(M, N, device="cuda")
    ad, bd, wd = a.cuda(), b.cuda(), ws.cuda()
    _lib.check(L.vd_op_linear_split(_lib.ptr(ad), M, K, _lib.ptr(wd), _lib.ptr(bd), None, 0, _lib.ptr(o), N, _lib.current_stream()))
    ref = a.double() @ w.double().t() + b.double()
    e = (o.cpu().double() - ref).abs()
    out["linear"] = {"max_err": float(e.max()), "mean_err": float(e.mean()), "ref_rms": float(ref.pow(2).mean().sqrt())}
    # ---- conv 3x3: 128 -> 128 at 16x16, 8 frames
    Cin, Cout, H, nfr = 128, 128, 16, 8
    x = torch.randn(nfr, H, H, Cin, generator=g); wc = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    wp = torch.empty(48 * Cout * Cin, dtype=torch.int16)
    _lib.check(L.vd_pack_conv3_wino_s64(_lib.ptr(wc.contiguous()), _lib.ptr(wp), Cout, Cin))
    oc = torch.empty(nfr, H, H, Cout, device="cuda")
    xd, wpd = x.cuda(), wp.cuda()
    _lib.check(L.vd_op_conv_wino_r64(_lib.ptr(xd), Cin, nfr, H, H, 0, _lib.ptr(wpd), None, None, None, 0, _lib.ptr(oc), Cout, None,
                                     _lib.current_stream()))
    refc = F.conv2d(x.permute(0, 3, 1, 2).double(), wc.double(), padding=1).permute(0, 2, 3, 1)
    e = (oc.cpu().double() - refc).abs()
    out["conv"] = {"max_err": float(e.max()), "mean_err": float(e.mean()), "ref_rms": float(refc.pow(2).mean().sqrt())}
    # ---- whole network against the reference's goldens
    for tag, name in [("eps_tiny", "unet_tiny.npz"), ("eps_full64", "unet_full64.npz")]:
        rec = load_npz(name)
        cfg = json.loads(str(rec["cfg_json"]))
        model, diff = vda.create_video_model_and_diffusion(**{k: cfg[k] for k in vda.video_model_and_diffusion_defaults()})
        model.load_state_dict(synth_sd(model.param_specs()))
        model.to("cuda").eval()
        errs = []
        if tag == "eps_tiny":
            for ci in range(n_cases(rec)):
                c = case_inputs(rec, ci)
                kw = dict(frame_indices=c["frame_indices"].cuda(), x0=c["x0"].cuda(), obs_mask=c["obs_mask"].cuda(), latent_mask=c["latent_mask"].cuda(),
                          kinda_marg_mask=c["kinda_marg_mask"].cuda(), x_t_minus_1=c["x0"].cuda(), observed_frames=c["observed_frames"])
                eps, _ = diff._wrap_model(model)(c["x"].cuda(), c["t"].cuda(), **kw)
                errs.append(float((eps.cpu() - c["eps"]).abs().max()))
            scale = float(np.abs(rec["c0_eps"]).max())
        else:
            T, n_obs, S = int(rec["T"][0]), int(rec["n_obs"][0]), cfg["image_size"]
            gg = torch.Generator().manual_seed(int(rec["seed"][0]))
            x0 = torch.rand(1, T, 3, S, S, generator=gg) * 2 - 1
            x0[:, n_obs:] = 0
            xx = torch.randn(1, T, 3, S, S, generator=gg)
            obs = torch.zeros(1, T, 1, 1, 1); obs[:, :n_obs] = 1
            kw = dict(frame_indices=torch.arange(T).view(1, T).cuda(), x0=x0.cuda(), obs_mask=obs.cuda(), latent_mask=(1 - obs).cuda(),
                      kinda_marg_mask=torch.zeros(1, T, 1, 1, 1).cuda(), x_t_minus_1=x0.cuda(), observed_frames="x_0")
            eps, _ = diff._wrap_model(model)(xx.cuda(), torch.tensor([int(rec["t"][0])]).cuda(), **kw)
            errs.append(float((eps.cpu() - torch.from_numpy(rec["eps"])).abs().max()))
            scale = float(np.abs(rec["eps"]).max())
        out[tag] = {"max_err": max(errs), "eps_max": scale}
        del model
    print(json.dumps(out))


if __name__ == "__main__":
    main()
