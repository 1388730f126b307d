#!/usr/bin/env python3
"""Sustained-load probe: run one conv shape back-to-back for a few seconds while sampling rocm-smi clocks/power."""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib
L = _lib.lib()
def make(nfr, C, Cout, H, pro, res):
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
    return f, 2.0 * nfr * H * H * Cout * C * 9
samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            s = [l.strip() for l in o.splitlines() if "sclk" in l or "Power" in l or "fclk" in l or "mclk" in l]
            samples.append(" | ".join(x.split(":")[-2].strip()[-12:] + ":" + x.split(":")[-1].strip() for x in s[:4]))
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.4)
for (pro, res) in [(1, 1), (0, 0)]:
    f, fl = make(128, 128, 128, 64, pro, res)
    f(); torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        for _ in range(50): f()
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    stop = True; th.join()
    print(f"pro={pro} res={res}: {n} launches in {dt:.2f}s -> {fl * n / dt / 1e12:.1f} TFLOP/s sustained")
    for s in samples[-3:]: print("   ", s)
