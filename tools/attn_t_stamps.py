#!/usr/bin/env python3
"""Phase stamps of one block of the matrix-pipe temporal attention kernel (library built with -DVD_ATT_TIMING):
VD_LIB=tools/_timing/att_timing.so python tools/attn_t_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

L = _lib.lib()
B, T, heads = 8, 16, 4
names = ["score terms (3 groups of requests + MFMAs)", "sum of the terms in LDS (3 barriers)", "softmax", "a.v: requests + MFMAs", "transpose + a.Rv", "stores"]
for HW, C in [(256, 384), (64, 512)]:
    qkv = torch.rand(B * T * HW, 3 * C, device="cuda") - 0.5
    R = [torch.rand(B, T, T, C, device="cuda") - 0.5 for _ in range(3)]
    out = torch.empty(B * T * HW, C, device="cuda")
    for rep in range(3):
        _lib.check(L.vd_op_attn_temporal(_lib.ptr(qkv), _lib.ptr(R[0]), _lib.ptr(R[1]), _lib.ptr(R[2]), None, B, T, HW, C, heads, 0,
                                         _lib.ptr(out), _lib.current_stream()))
        torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 16)()
    L.vd_debug_att_stamps.restype = ctypes.c_int
    L.vd_debug_att_stamps.argtypes = [ctypes.c_void_p]
    assert L.vd_debug_att_stamps(st) == 0
    v = list(st)
    print(f"HW={HW} C={C}: block (3,1,2) wave 0, 100 MHz ticks -> us:", " | ".join(f"{n}: {(v[i + 1] - v[i]) / 100:.2f}" for i, n in enumerate(names)),
          f"| total {(v[6] - v[0]) / 100:.2f}", flush=True)
