#!/usr/bin/env python3
"""One arithmetic mode of the library, end to end, in ONE process (VD_MATH is read once per process): run with
VD_MATH=f16x3 | bf16x6 | fp32 (tests/test_gpu_ops.py starts it as a child for every mode that is not the test process' own;
bench.py for the modes it reports beside the headline).  One JSON line:
  linear / conv / conv_s2 : |error| against an fp64 product -- max, mean, SIGNED mean, the output's RMS -- and the same
                            figures of the fp32-MFMA kernel (gemm_frag.hip / conv_wino.hip / the generic kernel) on the same inputs
  eps_tiny, eps_full64    : elements outside 1e-4 + 1e-4 |ref| and max |eps - golden| on tests/golden/unet_tiny.npz (10 cases) and
                            unet_full64.npz (the default 116 M model), both minted from the imported reference
  psample_tiny            : the same for p_sample at t = 249, 248, 1, 0 on tests/golden/psample_tiny.npz"""
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


def stats(got, ref64):
    d = got.double() - ref64
    return {"max_err": float(d.abs().max()), "mean_err": float(d.abs().mean()), "signed_mean_err": float(d.mean()),
            "ref_rms": float(ref64.pow(2).mean().sqrt())}


def outside(got, ref, atol=1e-4, rtol=1e-4):
    err = (got - ref).abs()
    return int((err > atol + rtol * ref.abs()).sum()), float(err.max())


def main():
    L = _lib.lib()
    out = {"version": L.vd_version().decode(), "mode": {0: "f16x3", 1: "bf16x6", 2: "fp32"}[L.vd_math_mode()]}
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(0)
    # ---- linear: M x K @ K x N on gemm_split.hip, and on the fp32-MFMA kernel
    M, K, N = 4096, 512, 384
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * (3.0 / K) ** 0.5; b = torch.randn(N, generator=g) * 0.1
    ws = torch.empty(L.vd_split_image_u16(N, K), dtype=torch.int16)
    _lib.check(L.vd_pack_linear_split(_lib.ptr(w.contiguous()), _lib.ptr(ws), N, K))
    wf = torch.empty(N * K)
    _lib.check(L.vd_pack_linear_frag(_lib.ptr(w.contiguous()), _lib.ptr(wf), N, K))
    o, of = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ad, bd, wd, wfd = a.cuda(), b.cuda(), ws.cuda(), wf.cuda()
    _lib.check(L.vd_op_linear_split(_lib.ptr(ad), M, K, _lib.ptr(wd), _lib.ptr(bd), None, 0, _lib.ptr(o), N, st))
    _lib.check(L.vd_op_conv(_lib.ptr(ad), None, K, K, M, 1, 1, 0, 1, 0, 1, None, _lib.ptr(wfd), None, _lib.ptr(bd), None, None, 0, None, None, 0,
                            _lib.ptr(of), N, st))
    ref = a.double() @ w.double().t() + b.double()
    out["linear"] = {**stats(o.cpu(), ref), "fp32_kernel": stats(of.cpu(), ref)}
    # ---- conv 3x3 stride 1: 128 -> 128 at 16x16, 8 frames, on conv_wino_r64.hip and on conv_wino.hip
    Cin, Cout, H, nfr = 128, 128, 16, 8
    x = torch.randn(nfr, H, H, Cin, generator=g); wc = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    wp = torch.empty(L.vd_split_image_u16(Cout, 16 * Cin), dtype=torch.int16)
    _lib.check(L.vd_pack_conv3_wino_split(_lib.ptr(wc.contiguous()), _lib.ptr(wp), Cout, Cin))
    ww = torch.empty(16 * Cout * Cin)
    _lib.check(L.vd_pack_conv3_wino(_lib.ptr(wc.contiguous()), _lib.ptr(ww), Cout, Cin))
    oc, ocf = torch.empty(nfr, H, H, Cout, device="cuda"), torch.empty(nfr, H, H, Cout, device="cuda")
    xd, wpd, wwd = x.cuda(), wp.cuda(), ww.cuda()
    _lib.check(L.vd_op_conv_wino_split(_lib.ptr(xd), Cin, nfr, H, H, 0, _lib.ptr(wpd), None, None, None, 0, _lib.ptr(oc), Cout, None, st))
    _lib.check(L.vd_op_conv(_lib.ptr(xd), None, Cin, Cin, nfr, H, H, 0, 1, 1, 3, None, None, _lib.ptr(wwd), None, None, None, 0, None, None, 0,
                            _lib.ptr(ocf), Cout, st))
    refc = F.conv2d(x.permute(0, 3, 1, 2).double(), wc.double(), padding=1).permute(0, 2, 3, 1)
    out["conv"] = {**stats(oc.cpu(), refc), "fp32_kernel": stats(ocf.cpu(), refc)}
    # ---- whole network against the reference's goldens
    for tag, name in [("eps_tiny", "unet_tiny.npz"), ("eps_full64", "unet_full64.npz")]:
        rec = load_npz(name)
        cfg = json.loads(str(rec["cfg_json"]))
        model, diff = vda.create_video_model_and_diffusion(**{k: cfg[k] for k in vda.video_model_and_diffusion_defaults()})
        model.load_state_dict(synth_sd(model.param_specs()))
        model.to("cuda").eval()
        bad, worst = 0, 0.0
        if tag == "eps_tiny":
            for ci in range(n_cases(rec)):
                c = case_inputs(rec, ci)
                kw = dict(frame_indices=c["frame_indices"].cuda(), x0=c["x0"].cuda(), obs_mask=c["obs_mask"].cuda(), latent_mask=c["latent_mask"].cuda(),
                          kinda_marg_mask=c["kinda_marg_mask"].cuda(), x_t_minus_1=c["x0"].cuda(), observed_frames=c["observed_frames"])
                eps, _ = diff._wrap_model(model)(c["x"].cuda(), c["t"].cuda(), **kw)
                nb, mx = outside(eps.cpu(), c["eps"])
                bad += nb; worst = max(worst, mx)
            scale = float(np.abs(rec["c0_eps"]).max())
            # p_sample with the recorded noise (tests/test_gpu_engine.py::test_p_sample_and_ddim_match_reference_golden)
            pr = load_npz("psample_tiny.npz")
            c = {k: torch.from_numpy(pr[k]) for k in ["x", "x0", "noise", "obs_mask", "latent_mask", "kinda_marg_mask", "frame_indices"]}
            kw = dict(frame_indices=c["frame_indices"].cuda(), x0=c["x0"].cuda(), obs_mask=c["obs_mask"].cuda(), latent_mask=c["latent_mask"].cuda(),
                      kinda_marg_mask=c["kinda_marg_mask"].cuda(), x_t_minus_1=c["x0"].cuda(), observed_frames="x_0")
            pb, pw = 0, 0.0
            for t_val in [249, 248, 1, 0]:
                t = torch.tensor([t_val] * c["x"].shape[0], device="cuda")
                sample, _ = diff._step(0, model, c["x"].cuda(), t, True, None, kw, 0.0, c["noise"])
                nb, mx = outside(sample.cpu(), torch.from_numpy(pr[f"t{t_val}_psample"]))
                pb += nb; pw = max(pw, mx)
            out["psample_tiny"] = {"outside_tol": pb, "max_err": pw}
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
            bad, worst = outside(eps.cpu(), torch.from_numpy(rec["eps"]))
            scale = float(np.abs(rec["eps"]).max())
        out[tag] = {"outside_tol": bad, "max_err": worst, "eps_max": scale}
        del model
    print(json.dumps(out))


if __name__ == "__main__":
    main()
