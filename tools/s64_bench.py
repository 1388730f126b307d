#!/usr/bin/env python3
"""Per-layer timing of the split Winograd kernels (conv_wino_r64.hip / conv_wino_s64.hip) over the 3x3 stride-1 conv census of the headline window
(SURVEY appendix B x 128 frames).  python tools/s64_bench.py [--reps 10] [--frames 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

CENSUS = [  # Cin, Cout, H (output), ups, count
    (128, 128, 64, 0, 7), (256, 256, 64, 1, 1), (256, 128, 64, 0, 2), (384, 128, 64, 0, 1),
    (128, 256, 32, 0, 1), (256, 256, 32, 0, 6), (384, 384, 32, 1, 1), (384, 256, 32, 0, 1), (512, 256, 32, 0, 1), (640, 256, 32, 0, 1),
    (256, 384, 16, 0, 1), (384, 384, 16, 0, 6), (512, 512, 16, 1, 1), (640, 384, 16, 0, 1), (768, 384, 16, 0, 1), (896, 384, 16, 0, 1),
    (384, 512, 8, 0, 1), (512, 512, 8, 0, 10), (896, 512, 8, 0, 1), (1024, 512, 8, 0, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--only", type=int, default=0, help="only layers with this output resolution")
    ap.add_argument("--quick", action="store_true", help="five representative layers only (kernel ablations)")
    ap.add_argument("--kernel", default="split", choices=["split", "s64", "r64"], help="split: the engine's choice per layer")
    args = ap.parse_args()
    L = _lib.lib()
    OPS = {"split": L.vd_op_conv_wino_split, "s64": L.vd_op_conv_wino_s64, "r64": L.vd_op_conv_wino_r64}
    tot_ms = tot_fl = 0.0
    for Cin, Cout, H, ups, cnt in CENSUS:
        if args.only and H != args.only:
            continue
        if args.quick and (Cin, Cout, H) not in [(128, 128, 64), (256, 256, 32), (640, 256, 32), (384, 384, 16), (512, 512, 8)]:
            continue
        nfr, Hs = args.frames, H >> ups
        x0 = torch.rand(nfr, Hs, Hs, Cin, device="cuda") - 0.5
        ws = torch.randint(-2000, 2000, (48 * Cout * Cin,), device="cuda", dtype=torch.int16)
        b = torch.rand(Cout, device="cuda")
        res = torch.rand(nfr, H, H, Cout, device="cuda")
        out = torch.empty(nfr, H, H, Cout, device="cuda")
        split = L.vd_conv_stats_split(H)
        part = torch.empty(nfr, split, Cout, 2, dtype=torch.float64, device="cuda")

        def run():
            OP = OPS[args.kernel]
            _lib.check(OP(_lib.ptr(x0), Cin, nfr, Hs, Hs, ups, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(res), None, 0,
                                             _lib.ptr(out), Cout, _lib.ptr(part), _lib.current_stream()))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(args.reps):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / args.reps
        fl = 2.0 * 9 * nfr * H * H * Cin * Cout
        tot_ms += ms * cnt
        tot_fl += fl * cnt
        print(f"{Cin:4d}->{Cout:4d} @{H:3d}{' ups' if ups else '    '} x{cnt:2d}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s", flush=True)
    print(f"class total {tot_ms:.3f} ms/step, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s direct-equivalent")


if __name__ == "__main__":
    main()
