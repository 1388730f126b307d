#!/usr/bin/env python3
"""Phase stamps of ONE block of attn_temporal_mfma_kernel (timing build: tools/build_variant.sh att_t attn_temporal.hip -DVD_ATT_TIMING;
VD_LIB=tools/_timing/att_t.so python tools/attn_stamps.py): 100 MHz clock, 10 ns per tick."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

L = _lib.lib()
B, T, heads = 8, 16, 4
for HW, C in [(256, 384), (64, 512)]:
    qkv = torch.rand(B * T * HW, 3 * C, device="cuda") - 0.5
    R = [torch.rand(B, T, T, C, device="cuda") - 0.5 for _ in range(3)]
    mask = torch.ones(B, T, device="cuda")
    out = torch.empty(B * T * HW, C, device="cuda")
    for rep in range(3):
        _lib.check(L.vd_op_attn_temporal(_lib.ptr(qkv), _lib.ptr(R[0]), _lib.ptr(R[1]), _lib.ptr(R[2]), _lib.ptr(mask), B, T, HW, C, heads, 0, _lib.ptr(out),
                                         _lib.current_stream()))
        torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 16)()
    L.vd_debug_att_stamps.argtypes = [ctypes.c_void_p]
    assert L.vd_debug_att_stamps(st) == 0
    t = [st[i] for i in range(7)]
    names = ["score terms (first wave)", "scores to LDS + RPE adds (all waves)", "softmax", "a v", "transpose + a Rv", "stores"]
    print(f"HW={HW} C={C}: " + ", ".join(f"{n} {(t[i + 1] - t[i]) / 100:.2f} us" for i, n in enumerate(names)) + f"; block {(t[6] - t[0]) / 100:.2f} us")
