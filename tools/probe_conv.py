#!/usr/bin/env python3
"""Timing probes for the 3x3 fragment kernel on one layer shape: residual / prologue / tile-class variants."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib
L = _lib.lib()
def run(nfr, C, Cout, H, pro, res, reps=5):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(nfr, H, H, C, device="cuda", generator=g) - 0.5
    wf = torch.rand(9 * Cout * C, device="cuda", generator=g) * 0.05
    b = torch.rand(Cout, device="cuda", generator=g)
    A = torch.rand(nfr, C, device="cuda", generator=g) + 0.5 if pro else None
    B = torch.rand(nfr, C, device="cuda", generator=g) - 0.5 if pro else None
    r = torch.rand(nfr, H, H, Cout, device="cuda", generator=g) if res else None
    out = torch.empty(nfr, H, H, Cout, device="cuda")
    def f():
        _lib.check(L.vd_op_conv(_lib.ptr(x), None, C, C, nfr, H, H, 0, 1, 1, 3, None, _lib.ptr(wf), None, _lib.ptr(b), _lib.ptr(A),
                                _lib.ptr(B), 1 if pro else 0, _lib.ptr(r), None, 0, _lib.ptr(out), Cout, _lib.current_stream()))
    f(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps): f()
    e[1].record(); torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / reps
    return 2.0 * nfr * H * H * Cout * C * 9 / ms / 1e9
for (C, Cout, H) in [(128, 128, 64), (256, 256, 32), (512, 512, 8)]:
    for pro in (1, 0):
        for res in (1, 0):
            print(f"C={C}->{Cout} H={H} pro={pro} res={res}: {run(128, C, Cout, H, pro, res):6.1f} TFLOP/s", flush=True)
