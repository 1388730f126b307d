#!/usr/bin/env python3
"""Kernel time of gemm_split.hip on the headline model's GEMM shapes (HIP events over 20 launches).  With VD_LIB pointing
at a -DVD_GS_SKIP=n build it is the ablation used in DESIGN.md (bit 0 no weight loads, 1 no A staging, 2 no split VALU).
  python tools/gemm_split_timing.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("video-diffusion_amd._lib")

SHAPES = [(32768, 384, 1152, 0), (8192, 512, 1536, 0), (32768, 384, 384, 1), (8192, 512, 512, 1), (524288, 256, 128, 0),
          (131072, 640, 256, 0), (2048, 512, 512, 0)]
L = _lib.lib()
for M, K, N, res in SHAPES:
    a = torch.rand(M, K, device="cuda") - 0.5
    ws = torch.randint(-2000, 2000, (3 * N * K,), device="cuda", dtype=torch.int16)
    b = torch.rand(N, device="cuda")
    r = torch.rand(M, N, device="cuda") if res else None
    out = torch.empty(M, N, device="cuda")
    call = lambda: _lib.check(L.vd_op_linear_split(_lib.ptr(a), M, K, _lib.ptr(ws), _lib.ptr(b), _lib.ptr(r), 0, _lib.ptr(out), N,  # noqa: E731
                                                   _lib.current_stream()))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    print(f"M {M:6d} K {K:4d} N {N:4d}{' +res' if res else '     '}: {us:7.1f} us = {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s fp32-equivalent "
          f"({12.0 * M * N * K / us * 1e-6 / 2516.6:.2f} of the bf16 pipe)", flush=True)
