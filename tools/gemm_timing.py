#!/usr/bin/env python3
"""Phase timing of gemm_frag_kernel (one mid-grid block, wave 0) from shader-clock stamps; needs a library built with
-DVD_GEMM_TIMING (VD_LIB=... python tools/gemm_timing.py).  Cycles: prologue / K loop / epilogue."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

SHAPES = [(32768, 384, 1152, 0), (32768, 384, 384, 1), (8192, 512, 1536, 0), (131072, 640, 256, 0), (524288, 256, 128, 0)]
L = _lib.lib()
L.vd_debug_gemm_stamps.restype = ctypes.c_int
L.vd_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
if hasattr(L, "vd_debug_gemm_occupancy"):
    print("blocks per CU (runtime occupancy query, <128,128>):", L.vd_debug_gemm_occupancy())
for M, K, N, res in SHAPES:
    x = torch.rand(M, K, device="cuda") - 0.5
    wf = torch.rand(N * K, device="cuda") * 0.05
    b = torch.rand(N, device="cuda")
    r = torch.rand(M, N, device="cuda") if res else None
    out = torch.empty(M, N, device="cuda")
    st = (ctypes.c_ulonglong * 6)()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(3):
        if it == 2:
            ev[0].record()
        _lib.check(L.vd_op_conv(_lib.ptr(x), None, K, K, M, 1, 1, 0, 1, 0, 1, None, _lib.ptr(wf), None, _lib.ptr(b), None, None, 0,
                                _lib.ptr(r), None, 0, _lib.ptr(out), N, _lib.current_stream()))
    ev[1].record()
    torch.cuda.synchronize()
    assert L.vd_debug_gemm_stamps(st) == 0
    t = list(st)
    nch = K // 32
    us = ev[0].elapsed_time(ev[1]) * 1e3
    print(f"M {M:6d} K {K:4d} N {N:4d}: {us:7.1f} us {2.0*M*N*K/us/1e6:6.1f} TFLOP/s | prologue {t[1]-t[0]:6d}  loop {t[2]-t[1]:7d} "
          f"({(t[2]-t[1])/nch:6.0f}/chunk; 64 MFMA = 4096/wave)  epilogue {t[3]-t[2]:6d}  total {t[3]-t[0]:7d} | kernel "
          f"block lasted {(t[5]-t[4])/100:.1f} us -> shader clock {(t[3]-t[0])/max(t[5]-t[4],1)*0.1:.2f} GHz", flush=True)
