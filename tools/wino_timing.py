#!/usr/bin/env python3
"""Phase timing of conv3x3_wino_kernel (block 0, wave 0) from shader-clock stamps; needs a library built with
-DVD_WINO_TIMING (VD_LIB=... python tools/wino_timing.py).  Prints cycles: prologue / main loop / output transform."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

SHAPES = [(128, 128, 0, 128, 64, 1), (128, 256, 0, 256, 32, 1), (128, 384, 256, 256, 32, 1), (128, 384, 0, 384, 16, 1),
          (128, 512, 0, 512, 8, 1), (128, 512, 512, 512, 8, 1)]
L = _lib.lib()
L.vd_debug_wino_stamps.restype = ctypes.c_int
L.vd_debug_wino_stamps.argtypes = [ctypes.c_void_p]
for nfr, C0, C1, Cout, H, pro in SHAPES:
    Cin = C0 + C1
    x0 = torch.rand(nfr, H, H, Cin, device="cuda") - 0.5
    ww = torch.rand(16 * Cout * Cin, device="cuda") * 0.05
    b = torch.rand(Cout, device="cuda")
    res = torch.rand(nfr, H, H, Cout, device="cuda")
    out = torch.empty(nfr, H, H, Cout, device="cuda")
    st = (ctypes.c_ulonglong * 10)()
    for _ in range(3):
        _lib.check(L.vd_op_conv(_lib.ptr(x0), None, Cin, Cin, nfr, H, H, 0, 1, 1, 3, None, None, _lib.ptr(ww), _lib.ptr(b),
                                None, None, 0, _lib.ptr(res), None, 0, _lib.ptr(out), Cout, _lib.current_stream()))
        torch.cuda.synchronize()
    assert L.vd_debug_wino_stamps(st) == 0
    if hasattr(L, "vd_debug_wino_segments"):
        sg = (ctypes.c_ulonglong * 32)()
        L.vd_debug_wino_segments.restype = ctypes.c_int
        L.vd_debug_wino_segments.argtypes = [ctypes.c_void_p]
        assert L.vd_debug_wino_segments(sg) == 0
        nch = Cin // 16
        for w in range(4):
            print(f"   wave {w} per chunk: " + "  ".join(f"g{i}: issue+mfma {sg[8*w+2*i]/nch:6.0f} post {sg[8*w+2*i+1]/nch:5.0f}" for i in range(4)))
    t = list(st)
    nch = Cin // 16
    print(f"Cin {Cin:4d} Cout {Cout:4d} H {H:2d}: prologue {t[1]-t[0]:6d}  loop {t[2]-t[1]:8d} ({(t[2]-t[1])/nch:6.0f}/chunk, ideal 8192)"
          f"  epilogue {t[3]-t[2]:6d} = addr+res loads {t[4]-t[2]} + barrier {t[5]-t[4]} + Z write {t[6]-t[5]} + barrier {t[7]-t[6]}"
          f" + Z read/store {t[3]-t[7]}   total {t[3]-t[0]:8d} = {(t[9]-t[8])/100:.1f} us -> shader clock "
          f"{(t[3]-t[0])/max(t[9]-t[8],1)*0.1:.2f} GHz", flush=True)
