#!/usr/bin/env python3
"""Sustained timing of gemm_split on the projection shapes of the headline window (VD_GS_NO192=1: 128x128 tiles only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402

L = _lib.lib()
for M, K, N, res in [(32768, 384, 1152, 0), (8192, 512, 1536, 0), (32768, 384, 384, 1), (8192, 512, 512, 1), (131072, 256, 256, 0), (524288, 256, 128, 0), (524288, 384, 128, 0), (131072, 512, 256, 0), (131072, 640, 256, 0)]:
    a = torch.rand(M, K, device="cuda") - 0.5
    w = torch.randint(-2000, 2000, (3 * N * K,), device="cuda", dtype=torch.int16)
    b = torch.rand(N, device="cuda")
    r = torch.rand(M, N, device="cuda") if res else None
    out = torch.empty(M, N, device="cuda")
    def run():
        _lib.check(L.vd_op_linear_split(_lib.ptr(a), M, K, _lib.ptr(w), _lib.ptr(b), _lib.ptr(r) if res else None, 0, _lib.ptr(out), N, _lib.current_stream()))
    for _ in range(20): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print(f"M {M:6d} K {K:4d} N {N:4d} res {res}: {us:7.1f} us  {2*M*K*N/us*1e-6:6.1f} TFLOP/s", flush=True)
