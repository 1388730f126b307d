#!/usr/bin/env python3
"""Sustained timing of the 3x3 Winograd kernels on three long-K layers (VD_WINO_KERNEL=s64|r64 names the kernel; default: the engine's choice):
kernel time over 200 back-to-back launches (clocks ramped), expressed per 16-channel chunk of one work item.
The MFMA floor of a chunk is 96 MFMAs per SIMD x 32 cycles = 1.30 us at 2.36 GHz."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

SHAPES = [(128, 128, 128, 64), (128, 384, 128, 64), (128, 640, 256, 32), (128, 384, 384, 16), (128, 896, 384, 16)]
L = _lib.lib()
op = {"s64": L.vd_op_conv_wino_s64, "r64": L.vd_op_conv_wino_r64}.get(os.environ.get("VD_WINO_KERNEL", ""), L.vd_op_conv_wino_split)
for nfr, Cin, Cout, H in SHAPES:
    x0 = torch.rand(nfr, H, H, Cin, device="cuda") - 0.5
    ws = torch.randint(-2000, 2000, (48 * Cout * Cin,), device="cuda", dtype=torch.int16)
    b = torch.rand(Cout, device="cuda")
    res = torch.rand(nfr, H, H, Cout, device="cuda")
    out = torch.empty(nfr, H, H, Cout, device="cuda")

    def run():
        op(_lib.ptr(x0), Cin, nfr, H, H, 0, _lib.ptr(ws), _lib.ptr(b), (None if os.environ.get("VD_WT_NORES") else _lib.ptr(res)), None, 0, _lib.ptr(out), Cout, None, _lib.current_stream())
    for _ in range(100):
        run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(200):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 5.0
    items = nfr * (H // 16) ** 2 * (Cout // 64)
    rounds = (items + 255) // 256
    if hasattr(L, "vd_debug_r64_stamps") and os.environ.get("VD_WINO_KERNEL", "") != "s64":
        import ctypes
        st = (ctypes.c_ulonglong * 8)()
        L.vd_debug_r64_stamps.restype = ctypes.c_int
        L.vd_debug_r64_stamps.argtypes = [ctypes.c_void_p]
        assert L.vd_debug_r64_stamps(st) == 0
        t = list(st)
        print(f"   last item of block 7 (cycles): prologue {t[1]-t[0]}, loop {t[2]-t[1]} = {(t[2]-t[1]) / (Cin // 16):.0f}/chunk, output transform {t[3]-t[2]}; "
              f"{(t[5]-t[4]) / 100:.2f} us -> {(t[3]-t[0]) / max(t[5]-t[4], 1) * 0.1:.2f} GHz")
    print(f"Cin {Cin:4d} Cout {Cout:4d} H {H:2d}: {us:7.1f} us, {us / rounds / (Cin // 16) * 1e3:6.0f} ns per chunk (floor 1300), "
          f"{2 * 9 * Cin * Cout * nfr * H * H / us * 1e-6:6.1f} TFLOP/s direct-equivalent", flush=True)
