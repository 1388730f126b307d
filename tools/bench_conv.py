#!/usr/bin/env python3
"""Micro-benchmark of the conv / linear kernel through vd_op_conv on the layer shapes of the headline
config (B=8, T=16 -> 128 frames).  Prints TFLOP/s per shape; used to iterate on csrc/igemm.hip and
csrc/conv_halo.hip in isolation.   python tools/bench_conv.py [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

SHAPES = [  # name, nfr, C0, C1, Cout, H, ksz, ups, stride, prologue
    ("res64  128->128 3x3 +gn", 128, 128, 0, 128, 64, 3, 0, 1, 1),
    ("res64  256->128 3x3 cat", 128, 128, 128, 128, 64, 3, 0, 1, 1),
    ("up64   256->256 3x3 ups", 128, 256, 0, 256, 32, 3, 1, 1, 0),
    ("res32  256->256 3x3 +gn", 128, 256, 0, 256, 32, 3, 0, 1, 1),
    ("res32  640->256 3x3 cat", 128, 384, 256, 256, 32, 3, 0, 1, 1),
    ("res16  384->384 3x3 +gn", 128, 384, 0, 384, 16, 3, 0, 1, 1),
    ("res16  896->384 3x3 cat", 128, 512, 384, 384, 16, 3, 0, 1, 1),
    ("res8   512->512 3x3 +gn", 128, 512, 0, 512, 8, 3, 0, 1, 1),
    ("res8  1024->512 3x3 cat", 128, 512, 512, 512, 8, 3, 0, 1, 1),
    ("down   256->256 3x3 s2 ", 128, 256, 0, 256, 32, 3, 0, 2, 0),
    ("skip32 640->256 1x1    ", 128, 384, 256, 256, 32, 1, 0, 1, 0),
    ("qkv16  384->1152 lin   ", 128 * 256, 384, 0, 1152, 1, 1, 0, 1, 0),
    ("proj8  512->512 lin    ", 128 * 64, 512, 0, 512, 1, 1, 0, 1, 0),
    ("film   512->15k lin M128", 128, 512, 0, 15360, 1, 1, 0, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    L = _lib.lib()
    tot_fl = tot_ms = 0.0
    for name, nfr, C0, C1, Cout, H, k, ups, stride, pro in SHAPES:
        if args.only and args.only not in name:
            continue
        Cin = C0 + C1
        g = torch.Generator(device="cuda").manual_seed(0)
        x0 = torch.rand(nfr, H, H, C0, device="cuda", generator=g) - 0.5
        x1 = (torch.rand(nfr, H, H, C1, device="cuda", generator=g) - 0.5) if C1 else None
        w = (torch.rand(k * k, Cout, Cin, device="cuda", generator=g) - 0.5) * 0.05
        b = torch.rand(Cout, device="cuda", generator=g)
        A = torch.rand(nfr, Cin, device="cuda", generator=g) + 0.5 if pro else None
        B = torch.rand(nfr, Cin, device="cuda", generator=g) - 0.5 if pro else None
        pad = 1 if k == 3 else 0
        Ho = ((H << ups) + 2 * pad - k) // stride + 1
        out = torch.empty(nfr, Ho, Ho, Cout, device="cuda")
        res = torch.rand(nfr, Ho, Ho, Cout, device="cuda", generator=g) if pro and k == 3 else None

        wf = ww = None
        if k == 3 and stride == 1 and Cout % 64 == 0 and Cin % 32 == 0 and not os.environ.get("VD_NO_WINO"):
            # the engine's form of these layers: GroupNorm+SiLU and the skip concat are materialised by one elementwise
            # pass (norm.hip affine_act_kernel), the Winograd kernel reads one plain tensor
            x0 = torch.rand(nfr, H, H, Cin, device="cuda", generator=g) - 0.5
            x1, C0, A, B, pro = None, Cin, None, None, 0
        if k == 1:
            A = B = None                 # linear layers of the engine use SiLU-only prologues (time_embed, FiLM)
        if stride == 1 and not os.environ.get("VD_NO_HALO"):
            wf = torch.rand(9 * Cout * Cin, device="cuda", generator=g) * 0.05     # timing only: any values
            if k == 3 and Cout % 64 == 0 and not os.environ.get("VD_NO_WINO"):
                ww = torch.rand(16 * Cout * Cin, device="cuda", generator=g) * 0.05

        if k == 1 and x1 is None and os.environ.get("VD_MATH") != "fp32" and Cin % 32 == 0 and Cout % 32 == 0 and not pro:
            ws = torch.randint(-2000, 2000, (3 * Cout * Cin,), device="cuda", dtype=torch.int16)   # timing only: any bf16 bits

            def run():
                _lib.check(L.vd_op_linear_split(_lib.ptr(x0), nfr * H * H, Cin, _lib.ptr(ws), _lib.ptr(b), None, 0, _lib.ptr(out), Cout,
                                                _lib.current_stream()))
            run.__name__ = "split"
        elif (k == 3 and stride == 1 and x1 is None and not pro and Cout % 64 == 0 and Cin % 32 == 0 and H << ups >= 8
              and os.environ.get("VD_MATH") != "fp32" and not os.environ.get("VD_NO_WINO")):
            wsp = torch.randint(-2000, 2000, (48 * Cout * Cin,), device="cuda", dtype=torch.int16)  # timing only

            def run():
                _lib.check(L.vd_op_conv_wino_split(_lib.ptr(x0), Cin, nfr, H, H, ups, _lib.ptr(wsp), _lib.ptr(b), _lib.ptr(res), None, 0,
                                                   _lib.ptr(out), Cout, None, _lib.current_stream()))
        else:
            run = None

        def run_fp32():
            _lib.check(L.vd_op_conv(_lib.ptr(x0), _lib.ptr(x1), C0, Cin, nfr, H, H, ups, stride, pad, k, _lib.ptr(w),
                                    _lib.ptr(wf), _lib.ptr(ww), _lib.ptr(b), _lib.ptr(A), _lib.ptr(B), pro, _lib.ptr(res), None, 0,
                                    _lib.ptr(out), Cout, _lib.current_stream()))
        if run is None:
            run = run_fp32
        run()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(args.reps):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / args.reps
        fl = 2.0 * nfr * Ho * Ho * Cout * Cin * k * k
        tot_fl += fl
        tot_ms += ms
        print(f"{name}  M={nfr * Ho * Ho:7d}  {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    print(f"total {tot_fl / tot_ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
