#!/usr/bin/env python3
"""Phase timing of conv3x3_wino_s64_kernel (block 0, wave 0) from shader-clock stamps; needs a library built with
-DVD_WINO_TIMING (VD_HIPCC_FLAGS=-DVD_WINO_TIMING, or VD_LIB=... python tools/winos_timing.py).
Cycles: prologue / main loop / output transform."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

SHAPES = [(128, 128, 128, 64), (128, 256, 128, 64), (128, 384, 128, 64), (128, 256, 256, 32), (128, 640, 256, 32), (128, 384, 384, 16), (128, 512, 512, 8), (128, 1024, 512, 8)]
L = _lib.lib()
S64 = True
stamps = L.vd_debug_s64_stamps
op = L.vd_op_conv_wino_s64
stamps.restype = ctypes.c_int
stamps.argtypes = [ctypes.c_void_p]
for nfr, Cin, Cout, H in SHAPES:
    x0 = torch.rand(nfr, H, H, Cin, device="cuda") - 0.5
    ws = torch.randint(-2000, 2000, (48 * Cout * Cin,), device="cuda", dtype=torch.int16)
    b = torch.rand(Cout, device="cuda")
    res = torch.rand(nfr, H, H, Cout, device="cuda")
    out = torch.empty(nfr, H, H, Cout, device="cuda")
    st = (ctypes.c_ulonglong * 10)()
    for _ in range(3):
        _lib.check(op(_lib.ptr(x0), Cin, nfr, H, H, 0, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(res), None, 0,
                                           _lib.ptr(out), Cout, None, _lib.current_stream()))
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        op(_lib.ptr(x0), Cin, nfr, H, H, 0, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(res), None, 0, _lib.ptr(out), Cout, None, _lib.current_stream())
    ev1.record()
    torch.cuda.synchronize()
    kus = ev0.elapsed_time(ev1) * 50.0
    assert stamps(st) == 0
    t = list(st)
    nch = Cin // 16
    print(f"Cin {Cin:4d} Cout {Cout:4d} H {H:2d}: prologue {t[1]-t[0]:6d}  loop {t[2]-t[1]:8d} ({(t[2]-t[1])/nch:6.0f}/chunk, {96 if S64 else 48} MFMA = {3072 if S64 else 1536})"
          f"  epilogue {t[3]-t[2]:6d}  total {t[3]-t[0]:8d} = {(t[9]-t[8])/100:.1f} us -> {(t[3]-t[0])/max(t[9]-t[8],1)*0.1:.2f} GHz |" + (f" pro: patch {t[4]-t[0]} transform {t[5]-t[4]} split {t[1]-t[5]}; epi: n0 {t[6]-t[2]} n1 {t[3]-t[6]} |" if S64 else "") + f" kernel {kus:7.1f} us = {2*9*Cin*Cout*nfr*H*H/kus*1e-6:6.1f} TFLOP/s direct-equivalent", flush=True)
