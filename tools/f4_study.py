#!/usr/bin/env python3
"""CPU study asked for by the round-3 review before any F(4x4,3x3) kernel is built (zero GPU minutes): what would Winograd
F(4x4,3x3) cost in accuracy?  Emulates, in torch on the CPU, the arithmetic a kernel would run -- input transform B^T d B and
output transform A^T m A in fp32, U = G g G^T in fp64 rounded once to fp32, the element products summed over channels in fp32
(optionally with the operands carried as two fp16 pieces and three piece products, the library's default arithmetic) -- for
F(2x2,3x3) and F(4x4,3x3), against an fp64 direct convolution:
  1. per layer on the conv census' shapes, error relative to a plain fp32 direct conv;
  2. end to end: the oracle UNet (oracle/unet_ref.py) with every 3x3 stride-1 conv replaced, eps of the default 116 M model
     against the reference's golden tests/golden/unet_full64.npz (tolerance of the tier: 1e-4 + 1e-4 |ref|).
Writes profiles/r04_f4_study.json.     python tools/f4_study.py [--no-net]"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.set_num_threads(8)

MATS = {
    2: dict(G=[[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]],
            Bt=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
            At=[[1, 1, 1, 0], [0, 1, -1, -1]]),
    4: dict(G=[[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
            Bt=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
            At=[[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}


def f16_pieces(v):
    a0 = v.half().float()
    return a0, ((v - a0) * 4096).half().float()


def wino_conv(x, w, m, pieces=False):
    """x [N][C][H][W] fp32, w [O][C][3][3] fp32, F(m x m, 3x3), padding 1.  fp32 throughout except U (fp64 -> fp32)."""
    M = MATS[m]
    G, Bt, At = (torch.tensor(M[k], dtype=torch.float64) for k in ("G", "Bt", "At"))
    N, C, H, W = x.shape
    O = w.shape[0]
    a = m + 2
    th, tw = H // m, W // m
    U = torch.einsum("ik,ockl,jl->ijoc", G, w.double(), G).float()                      # [a][a][O][C]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                                          # [N][C][th][tw][a][a]
    Btf = Bt.float()
    V = torch.einsum("ik,ncyxkl,jl->ijnyxc", Btf, tiles, Btf)                           # fp32 transform
    V = V.reshape(a, a, N * th * tw, C)
    if pieces:
        mx = U.abs().amax((0, 1, 3))                                                     # per-cout scale (split_pack.hip)
        s = torch.where(mx > 0, torch.ldexp(torch.ones(O), 14 - torch.frexp(mx)[1]), torch.ones(O))
        Us = U * s[None, None, :, None]
        b0 = Us.half().float(); b1 = (Us - b0).half().float(); b2 = (b0 / 4096).half().float()
        a0, a1 = f16_pieces(V)
        A = torch.cat([a0, a0, a1], -1)                                                  # one fp32 accumulation chain, as the kernel's
        B = torch.cat([b0, b1, b2], -1)
        Mm = torch.einsum("ijtk,ijok->ijto", A, B) / s[None, None, None, :]
    else:
        Mm = torch.einsum("ijtc,ijoc->ijto", V, U)
    Atf = At.float()
    Y = torch.einsum("pi,ijto,qj->topq", Atf, Mm, Atf)                                   # [tiles][O][m][m]
    Y = Y.reshape(N, th, tw, O, m, m).permute(0, 3, 1, 4, 2, 5).reshape(N, O, H, W)
    return Y


def layer_study():
    out = []
    g = torch.Generator().manual_seed(0)
    for (C, O, H) in [(128, 128, 64), (256, 256, 32), (384, 384, 16), (512, 512, 8), (1024, 512, 8)]:
        N = 2
        x = F.silu(torch.randn(N, C, H, H, generator=g) * 1.5)                          # what a conv input looks like: SiLU(GroupNorm(.))
        w = torch.randn(O, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
        ref = F.conv2d(x.double(), w.double(), padding=1)
        rms = float(ref.pow(2).mean().sqrt())
        e = lambda y: (float((y.double() - ref).abs().max()), float((y.double() - ref).abs().mean()))  # noqa: E731
        d32 = e(F.conv2d(x, w, padding=1))
        row = {"shape": f"{C}->{O}@{H}", "ref_rms": rms, "direct_fp32": d32}
        for m in (2, 4):
            for pc in (False, True):
                mx, mn = e(wino_conv(x, w, m, pc))
                row[f"F{m}{'_f16x3' if pc else '_fp32'}"] = {"max": mx, "mean": mn, "max_over_direct": mx / d32[0], "mean_over_direct": mn / d32[1]}
        out.append(row)
        print(json.dumps(row), flush=True)
    return out


def net_study():
    from helpers import load_npz, synth_sd
    import video_diffusion_amd as vda
    from oracle.unet_ref import UNetRef
    rec = load_npz("unet_full64.npz")
    cfg = json.loads(str(rec["cfg_json"]))
    model = vda.create_video_model_and_diffusion(**{k: cfg[k] for k in vda.video_model_and_diffusion_defaults()})[0]
    sd = synth_sd(model.param_specs())
    T, n_obs, S = int(rec["T"][0]), int(rec["n_obs"][0]), cfg["image_size"]
    gg = torch.Generator().manual_seed(int(rec["seed"][0]))
    x0 = torch.rand(1, T, 3, S, S, generator=gg) * 2 - 1
    x0[:, n_obs:] = 0
    xx = torch.randn(1, T, 3, S, S, generator=gg)
    obs = torch.zeros(1, T, 1, 1, 1); obs[:, :n_obs] = 1
    from oracle.sampler_ref import SamplerRef
    from oracle.schedule_ref import ScheduleRef
    gold = torch.from_numpy(rec["eps"])
    res = {}
    real_conv = UNetRef.conv
    for tag, m, pc in [("oracle_direct_fp32", 0, False), ("F2_fp32", 2, False), ("F2_f16x3", 2, True), ("F4_fp32", 4, False), ("F4_f16x3", 4, True)]:
        def conv(self, x, pre, stride=1, pad=1, _m=m, _pc=pc):
            w = self.p(pre + ".weight")
            if _m and stride == 1 and w.shape[-1] == 3 and x.shape[-1] % 4 == 0 and x.shape[-1] >= 8 and w.shape[0] % 64 == 0 and w.shape[1] % 32 == 0:
                return wino_conv(x, w, _m, _pc) + self.p(pre + ".bias")[None, :, None, None]
            return real_conv(self, x, pre, stride, pad)
        UNetRef.conv = conv
        ora = SamplerRef(ScheduleRef(cfg["diffusion_steps"], cfg["noise_schedule"], cfg["timestep_respacing"], cfg["sigma_small"],
                                     cfg["rescale_timesteps"]), UNetRef(cfg, sd))
        kw = dict(frame_indices=torch.arange(T).view(1, T), x0=x0, obs_mask=obs, latent_mask=1 - obs, kinda_marg_mask=torch.zeros(1, T, 1, 1, 1))
        eps = ora.eps(xx, torch.tensor([int(rec["t"][0])]), kw)
        err = (eps - gold).abs()
        lim = 1e-4 + 1e-4 * gold.abs()
        res[tag] = {"max_err": float(err.max()), "mean_err": float(err.mean()), "outside_tol": int((err > lim).sum()), "worst_over_tol": float((err / lim).max())}
        print(tag, json.dumps(res[tag]), flush=True)
    UNetRef.conv = real_conv
    return res


if __name__ == "__main__":
    out = {"layers": layer_study()}
    if "--no-net" not in sys.argv:
        out["network_unet_full64"] = net_study()
    with open(os.path.join(ROOT, "profiles", "r04_f4_study.json"), "w") as f:
        json.dump(out, f, indent=1)
