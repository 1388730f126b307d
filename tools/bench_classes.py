#!/usr/bin/env python3
"""The headline bench without its side legs, printed as one line per kernel class (for tools/ab.sh A/B runs of library variants).
    python tools/bench_classes.py [bench.py flags]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-fp32-ref", "--no-dropin",
                    "--no-full-window", *sys.argv[1:]], capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-2000:])
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
print(f"steps/s {d['value']:.3f}  ms/step {d['ms_per_step']:.3f}")
for k, v in sorted(d["kernel_classes"].items(), key=lambda kv: -kv[1]["ms"]):
    print(f"   {k:34s} {v['launches']:3d} {v['ms']:7.3f} ms  {v['tflops']} TFLOP/s  {v['gbs']} GB/s")
