#!/usr/bin/env python3
"""Measured error of the split kernels (gemm_split.hip, conv_wino_r64.hip) in the process' arithmetic (VD_MATH) against fp64, next
to the fp32-MFMA kernels (gemm_frag.hip, conv_wino.hip) on the same inputs: per shape max, mean and SIGNED mean error and the
ratios the tests bound (tests/test_gpu_ops.py::check_vs_fp32_kernel).  One JSON object on stdout; profiles/r04_split_accuracy.json
holds the f16x3 and bf16x6 runs.   VD_MATH=f16x3|bf16x6 python tools/split_accuracy.py"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402


def stats(o_s, o_f, ref):
    ds, df = o_s.double() - ref, o_f.double() - ref
    return {"split_max": float(ds.abs().max()), "split_mean": float(ds.abs().mean()), "split_signed_mean": float(ds.mean()),
            "fp32_max": float(df.abs().max()), "fp32_mean": float(df.abs().mean()), "fp32_signed_mean": float(df.mean()),
            "ratio_max": float(ds.abs().max() / df.abs().max()), "ratio_mean": float(ds.abs().mean() / df.abs().mean()),
            "ref_rms": float(ref.pow(2).mean().sqrt())}


def main():
    L = _lib.lib()
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(0)
    out = {"version": L.vd_version().decode(), "linear": {}, "conv3x3": {}}
    for scale_a in (1.0, 1e-3, 100.0):
        for M, K, N in [(4096, 32, 128), (4096, 64, 128), (4096, 128, 384), (4096, 384, 1152), (2048, 1152, 128), (1024, 4608, 128)]:
            a = (torch.rand(M, K, generator=g) * 2 - 1) * scale_a
            w = (torch.rand(N, K, generator=g) * 2 - 1) * (3.0 / K) ** 0.5
            ws = torch.empty(L.vd_split_image_u16(N, K), dtype=torch.int16)
            _lib.check(L.vd_pack_linear_split(_lib.ptr(w), _lib.ptr(ws), N, K))
            wf = torch.empty(N * K)
            _lib.check(L.vd_pack_linear_frag(_lib.ptr(w), _lib.ptr(wf), N, K))
            ad, wsd, wfd = a.cuda(), ws.cuda(), wf.cuda()
            o_s, o_f = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
            _lib.check(L.vd_op_linear_split(_lib.ptr(ad), M, K, _lib.ptr(wsd), None, None, 0, _lib.ptr(o_s), N, st))
            _lib.check(L.vd_op_conv(_lib.ptr(ad), None, K, K, M, 1, 1, 0, 1, 0, 1, None, _lib.ptr(wfd), None, None, None, None, 0,
                                    None, None, 0, _lib.ptr(o_f), N, st))
            torch.cuda.synchronize()
            out["linear"][f"M{M}_K{K}_N{N}_a{scale_a:g}"] = stats(o_s.cpu(), o_f.cpu(), a.double() @ w.double().t())
    for scale_a in (1.0, 1e-3):
        for nfr, Cin, Cout, H in [(2, 128, 128, 64), (4, 256, 256, 32), (8, 384, 384, 16), (16, 512, 512, 8), (16, 1024, 512, 8)]:
            x = F.silu(torch.randn(nfr, H, H, Cin, generator=g) * 1.5) * scale_a
            wc = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
            wp = torch.empty(L.vd_split_image_u16(Cout, 16 * Cin), dtype=torch.int16)
            _lib.check(L.vd_pack_conv3_wino_split(_lib.ptr(wc), _lib.ptr(wp), Cout, Cin))
            ww = torch.empty(16 * Cout * Cin)
            _lib.check(L.vd_pack_conv3_wino(_lib.ptr(wc), _lib.ptr(ww), Cout, Cin))
            xd, wpd, wwd = x.cuda(), wp.cuda(), ww.cuda()
            o_s, o_f = torch.empty(nfr, H, H, Cout, device="cuda"), torch.empty(nfr, H, H, Cout, device="cuda")
            _lib.check(L.vd_op_conv_wino_split(_lib.ptr(xd), Cin, nfr, H, H, 0, _lib.ptr(wpd), None, None, None, 0, _lib.ptr(o_s), Cout, None, st))
            _lib.check(L.vd_op_conv(_lib.ptr(xd), None, Cin, Cin, nfr, H, H, 0, 1, 1, 3, None, None, _lib.ptr(wwd), None, None, None, 0, None, None, 0,
                                    _lib.ptr(o_f), Cout, st))
            torch.cuda.synchronize()
            ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wc.double(), padding=1).permute(0, 2, 3, 1)
            out["conv3x3"][f"{Cin}to{Cout}_at{H}_a{scale_a:g}"] = stats(o_s.cpu(), o_f.cpu(), ref)
    allr = [v for grp in ("linear", "conv3x3") for v in out[grp].values()]
    out["worst_ratio_max"] = max(v["ratio_max"] for v in allr)
    out["worst_ratio_mean"] = max(v["ratio_mean"] for v in allr)
    out["worst_bias_over_mean_err"] = max(abs(v["split_signed_mean"]) / v["split_mean"] for v in allr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
