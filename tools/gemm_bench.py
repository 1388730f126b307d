#!/usr/bin/env python3
"""Per-shape timing of gemm_split_kernel over the linear / 1x1 / stride-2 shapes of the headline window.
python tools/gemm_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

LIN = [  # M, K, N, res, count
    (32768, 384, 1152, 0, 10), (8192, 512, 1536, 0, 12), (32768, 384, 384, 1, 10), (8192, 512, 512, 1, 12),
    (524288, 256, 128, 0, 2), (524288, 384, 128, 0, 1), (131072, 384, 256, 0, 1), (131072, 512, 256, 0, 1), (131072, 640, 256, 0, 1),
    (131072, 128, 256, 0, 1), (32768, 640, 384, 0, 1), (32768, 768, 384, 0, 1), (32768, 896, 384, 0, 1), (32768, 256, 384, 0, 1),
    (8192, 1024, 512, 0, 2), (8192, 896, 512, 0, 1), (8192, 384, 512, 0, 1), (524288, 64, 128, 0, 1)]
CONV2 = [(128, 128, 64), (256, 256, 32), (384, 384, 16)]     # Cin = Cout, input H (stride 2)


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    L = _lib.lib()
    tot = 0.0
    for M, K, N, res, cnt in LIN:
        a = torch.rand(M, K, device="cuda") - 0.5
        w = (torch.rand(N, K) - 0.5) * (12.0 / K) ** 0.5
        data = os.environ.get("VD_GEMM_BENCH_DATA", "")       # zero_a | zero_w | const: operands that do not toggle (tools/conv_bench.py)
        if data == "zero_a": a.zero_()
        if data == "zero_w": w.zero_()
        if data == "const": a.fill_(0.25); w.fill_(0.125)
        wp = torch.empty(L.vd_split_image_u16(N, K), dtype=torch.int16)
        _lib.check(L.vd_pack_linear_split(_lib.ptr(w), _lib.ptr(wp), N, K))
        ws = wp.cuda()
        b = torch.rand(N, device="cuda")
        r = torch.rand(M, N, device="cuda") if res else None
        out = torch.empty(M, N, device="cuda")
        ms = timeit(lambda: _lib.check(L.vd_op_linear_split(_lib.ptr(a), M, K, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(r), 0, _lib.ptr(out), N,
                                                            _lib.current_stream())), args.reps)
        tot += ms * cnt
        extra = ""
        if hasattr(L, "vd_debug_gs_stamps"):
            import ctypes
            st = (ctypes.c_ulonglong * 8)()
            L.vd_debug_gs_stamps.restype = ctypes.c_int
            L.vd_debug_gs_stamps.argtypes = [ctypes.c_void_p]
            assert L.vd_debug_gs_stamps(st) == 0
            t = list(st)
            extra = f"   one block (us at the 100 MHz clock): prologue {(t[1]-t[0]) / 100:.2f}, K loop {(t[2]-t[1]) / 100:.2f}, epilogue to stores acknowledged {(t[3]-t[2]) / 100:.2f}"
        print(f"lin  M={M:7d} K={K:5d} N={N:5d}{' res' if res else '    '} x{cnt:2d} {ms * 1e3:8.1f} us {2.0 * M * K * N / ms / 1e9:7.1f} TFLOP/s{extra}", flush=True)
    for C, Co, H in CONV2:
        nfr = 128
        x = torch.rand(nfr, H, H, C, device="cuda") - 0.5
        wc = (torch.rand(Co, C, 3, 3) - 0.5) * (12.0 / (9 * C)) ** 0.5
        wp = torch.empty(L.vd_split_image_u16(Co, 9 * C), dtype=torch.int16)
        _lib.check(L.vd_pack_conv3_split(_lib.ptr(wc), _lib.ptr(wp), Co, C))
        ws = wp.cuda()
        b = torch.rand(Co, device="cuda")
        out = torch.empty(nfr, H // 2, H // 2, Co, device="cuda")
        ms = timeit(lambda: _lib.check(L.vd_op_conv_split(_lib.ptr(x), C, nfr, H, H, 2, _lib.ptr(ws), _lib.ptr(b), None, _lib.ptr(out), Co,
                                                          _lib.current_stream())), args.reps)
        tot += ms
        print(f"s2   C={C:4d} H={H:3d}                    x 1 {ms * 1e3:8.1f} us {2.0 * 9 * nfr * (H // 2) ** 2 * C * Co / ms / 1e9:7.1f} TFLOP/s", flush=True)
    print(f"class total {tot:.3f} ms/step")


if __name__ == "__main__":
    main()
