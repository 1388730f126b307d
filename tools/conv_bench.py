#!/usr/bin/env python3
"""Per-layer timing of conv_wino_r64.hip over the 3x3 stride-1 conv census of the headline window (SURVEY appendix B x 128
frames), through vd_op_conv_wino_split (the process' arithmetic: VD_MATH).  VD_LIB=tools/_timing/<variant>.so times a
kernel-experiment build (tools/build_variant.sh); a -DVD_WINO_TIMING build also prints one work item's cycle stamps.
    python tools/conv_bench.py [--reps 10] [--frames 128] [--quick] [--only H]"""
import argparse
import ctypes
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
QUICK = [(128, 128, 64), (256, 256, 32), (640, 256, 32), (384, 384, 16), (896, 384, 16), (512, 512, 8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--only", type=int, default=0, help="only layers with this output resolution")
    ap.add_argument("--quick", action="store_true", help="six representative layers only (kernel ablations)")
    ap.add_argument("--r64", action="store_true", help="only the layers conv_wino_r64.hip's 64-cout block serves (no Upsample forms)")
    ap.add_argument("--act", action="store_true", help="time conv_wino_z128.hip's activating form (vd_op_conv_wino_act) on the shapes it serves, next to "
                                                       "the materialising pair it replaces (vd_op_affine_act + the plain conv)")
    args = ap.parse_args()
    L = _lib.lib()
    stamps = hasattr(L, "vd_debug_r64_stamps")
    tot_ms = tot_fl = 0.0
    for Cin, Cout, H, ups, cnt in CENSUS:
        if (args.only and H != args.only) or (args.quick and ((Cin, Cout, H) not in QUICK or ups)):
            continue
        nfr, Hs = args.frames, H >> ups
        if args.r64 and (ups or L.vd_conv_wino_block_couts(nfr, H, Cin, Cout) != 64):
            continue
        x0 = torch.rand(nfr, Hs, Hs, Cin, device="cuda") - 0.5
        w = (torch.rand(Cout, Cin, 3, 3) - 0.5) * (12.0 / (9 * Cin)) ** 0.5
        # VD_CONV_BENCH_DATA=zero_x | zero_w | const: the same instruction stream on operands that do not toggle (a kernel that gets faster on
        # them is limited by the board's power management, not by its schedule)
        data = os.environ.get("VD_CONV_BENCH_DATA", "")
        if data == "zero_x": x0.zero_()
        if data == "zero_w": w.zero_()
        if data == "const": x0.fill_(0.25); w.fill_(0.125)
        if data == "f16pairs":   # every fp32 word = two fp16 values of N(0, 1/4): what a pre-split (a0, a1) operand would put on the wires (-DVD_R64_ABL=34 builds)
            x0 = (torch.randn(nfr, Hs, Hs, 2 * Cin, device="cuda") * 0.5).half().view(torch.float32).contiguous()
        b, res = torch.rand(Cout, device="cuda"), torch.rand(nfr, H, H, Cout, device="cuda")
        out = torch.empty(nfr, H, H, Cout, device="cuda")
        part = torch.empty(nfr, L.vd_conv_stats_split(H), Cout, 2, dtype=torch.float64, device="cuda")
        if ups:                                                      # the engine's form: sub-pixel image, no residual
            wp = torch.empty(L.vd_split_image_u16(4 * Cout, 16 * Cin), dtype=torch.int16)
            _lib.check(L.vd_pack_conv3_wino_ups(_lib.ptr(w), _lib.ptr(wp), Cout, Cin))
            part = torch.empty(nfr, L.vd_conv_ups_stats_split(Hs), Cout, 2, dtype=torch.float64, device="cuda")
        else:
            wp = torch.empty(L.vd_split_image_u16(Cout, 16 * Cin), dtype=torch.int16)
            _lib.check(L.vd_pack_conv3_wino_split(_lib.ptr(w), _lib.ptr(wp), Cout, Cin))
        ws = wp.cuda()

        def run():
            if ups:
                _lib.check(L.vd_op_conv_wino_ups(_lib.ptr(x0), Cin, nfr, Hs, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(out), Cout, _lib.ptr(part),
                                                 _lib.current_stream()))
            else:
                _lib.check(L.vd_op_conv_wino_split(_lib.ptr(x0), Cin, nfr, Hs, Hs, 0, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(res), None, 0,
                                                   _lib.ptr(out), Cout, _lib.ptr(part), _lib.current_stream()))
        if args.act:
            if ups or not L.vd_conv_wino_act_ok(nfr, H, Cin, Cout):
                continue
            A, Bc = torch.rand(nfr, Cin, device="cuda") + 0.5, torch.rand(nfr, Cin, device="cuda") - 0.5
            img = torch.empty_like(x0)

            def run_act():
                _lib.check(L.vd_op_conv_wino_act(_lib.ptr(x0), None, Cin, Cin, nfr, Hs, Hs, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(A), _lib.ptr(Bc), _lib.ptr(res),
                                                 _lib.ptr(out), Cout, _lib.ptr(part), _lib.current_stream()))

            def run_pair():
                _lib.check(L.vd_op_affine_act(_lib.ptr(x0), None, Cin, Cin, _lib.ptr(A), _lib.ptr(Bc), nfr, Hs * Hs, 1, _lib.ptr(img), _lib.current_stream()))
                _lib.check(L.vd_op_conv_wino_split(_lib.ptr(img), Cin, nfr, Hs, Hs, 0, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(res), None, 0,
                                                   _lib.ptr(out), Cout, _lib.ptr(part), _lib.current_stream()))
            tms = []
            for fn in (run, run_act, run_pair):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                tms.append(e0.elapsed_time(e1) / args.reps * 1e3)
            print(f"{Cin:5d} -> {Cout:4d} @ {H:2d} x{cnt:2d}: plain conv {tms[0]:7.1f} us | activating conv {tms[1]:7.1f} us | affine_act + plain conv {tms[2]:7.1f} us", flush=True)
            if hasattr(L, "vd_debug_z128_stamps"):
                for nm, fn in (("plain", run), ("activating", run_act)):
                    fn(); torch.cuda.synchronize()
                    st = (ctypes.c_ulonglong * 16)()
                    L.vd_debug_z128_stamps.restype = ctypes.c_int
                    L.vd_debug_z128_stamps.argtypes = [ctypes.c_void_p]
                    assert L.vd_debug_z128_stamps(st) == 0
                    t = list(st)
                    ep = " | ".join(f"n{n}: Z {t[3+2*n]-(t[2] if n == 0 else t[2+2*n])}, rest {t[4+2*n]-t[3+2*n]}" for n in range(4))
                    print(f"      {nm}: item of block 7 (cycles): prologue {t[1]-t[0]}, loop {t[2]-t[1]} = {(t[2]-t[1]) / (Cin // 16):.0f}/chunk, output transform "
                          f"{t[10]-t[2]} [{ep}]; {(t[10]-t[0]) / max(t[15]-t[14], 1) * 0.1:.2f} GHz", flush=True)
            continue
        for _ in range(3):
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
        extra = ""
        if stamps:
            st = (ctypes.c_ulonglong * 16)()
            L.vd_debug_r64_stamps.restype = ctypes.c_int
            L.vd_debug_r64_stamps.argtypes = [ctypes.c_void_p]
            assert L.vd_debug_r64_stamps(st) == 0
            t = list(st)
            ep = " | ".join(f"n{n}: Z {t[3+4*n]-(t[2] if n == 0 else t[6])}, barrier {t[4+4*n]-t[3+4*n]}, sum+store {t[5+4*n]-t[4+4*n]}, stats {t[6+4*n]-t[5+4*n]}"
                            for n in range(2))
            extra = (f"\n      item of block 7 (cycles): prologue {t[1]-t[0]}, loop {t[2]-t[1]} = {(t[2]-t[1]) / (Cin // 16):.0f}/chunk, output transform "
                     f"{t[10]-t[2]} [{ep}]; {(t[10]-t[0]) / max(t[15]-t[14], 1) * 0.1:.2f} GHz")
        if hasattr(L, "vd_debug_z128_stamps") and not ups and L.vd_conv_wino_block_couts(nfr, H, Cin, Cout) == 128:
            st = (ctypes.c_ulonglong * 16)()
            L.vd_debug_z128_stamps.restype = ctypes.c_int
            L.vd_debug_z128_stamps.argtypes = [ctypes.c_void_p]
            assert L.vd_debug_z128_stamps(st) == 0
            t = list(st)
            ep = " | ".join(f"n{n}: Z {t[3+2*n]-(t[2] if n == 0 else t[2+2*n])}, rest {t[4+2*n]-t[3+2*n]}" for n in range(4))
            extra = (f"\n      z128 item of block 7 (cycles): prologue {t[1]-t[0]}, loop {t[2]-t[1]} = {(t[2]-t[1]) / (Cin // 16):.0f}/chunk, output transform "
                     f"{t[10]-t[2]} [{ep}]; {(t[10]-t[0]) / max(t[15]-t[14], 1) * 0.1:.2f} GHz")
        print(f"{Cin:5d} -> {Cout:4d} @ {H:2d}{' ups' if ups else '    '} x{cnt:2d}: {ms * 1e3:8.1f} us  {fl / ms * 1e-9:6.1f} TFLOP/s direct-equivalent{extra}", flush=True)
    print(f"class total {tot_ms:.3f} ms per step, {tot_fl / tot_ms * 1e-9:.1f} TFLOP/s direct-equivalent ({_lib.lib().vd_version().decode()})")


if __name__ == "__main__":
    main()
