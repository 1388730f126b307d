#!/usr/bin/env python3
"""Timing probe: the same two denoise steps launched eagerly vs replayed from a captured hipGraph (kernel-to-kernel
gaps and host launch cost).  The replay repeats the captured noise offsets, so it is a timing experiment only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import video_diffusion_amd as vda  # noqa: E402
from video_diffusion_amd import dist as vdist  # noqa: E402

vdist.init()
cfg = bench.headline_cfg()
model, diff = vda.create_video_model_and_diffusion(**cfg)
model.to("cuda").eval()
specs = model.param_specs()
vdist.share_weights(model, lambda: {k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in specs}, 0)
kw = bench.make_window(8, 16, cfg["image_size"], 4, seed=1234, device=torch.device("cuda"))
st = bench.Stepper(model, diff, kw, seed=5)
for i in range(4):
    st.step(249 - i)
torch.cuda.synchronize()

def eager(n):
    t0 = time.perf_counter()
    for i in range(n):
        st.step(200 - (i % 50))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("eager ms/step", round(eager(20), 3))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    st.stream = bench._lib.current_stream()
    st.step(100); st.step(99)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        st.stream = bench._lib.current_stream()
        st.step(100); st.step(99)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", round((time.perf_counter() - t0) / 20 * 1e3, 3))
