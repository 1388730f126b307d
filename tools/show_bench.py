#!/usr/bin/env python3
"""Print the headline and the per-class table of a bench.py JSON line.   python tools/show_bench.py out.json"""
import json
import sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["unit"], d["ms_per_step"], "ms/step")
print(d["roofline"])
for k, v in d["kernel_classes"].items():
    print(f"  {k:30s} n={v['launches']:3d} {v['ms']:8.3f} ms  {v['tflops'] or 0:7.1f} TFLOP/s {v['gbs'] or 0:8.1f} GB/s")
