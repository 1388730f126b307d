"""Would the step run faster as TWO half-batches on two HIP streams (an HBM-bound pass of one half under an MFMA-bound
convolution of the other)?  Probe without touching the engine: two model instances (two engines, two workspaces), each
denoising B = 4 of the headline's 8 clips on its own stream, against one engine with B = 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_diffusion_amd as vda
from video_diffusion_amd import _lib
import bench

class A: pass
args = A(); args.image_size = 64; args.frames = 16; args.respacing = "ddim250"; args.num_res_blocks = 2
cfg = bench.bench_cfg(vda, args)
dev = torch.device("cuda", 0)
def make():
    model, diff = vda.create_video_model_and_diffusion(**cfg)
    model.load_state_dict({k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in model.param_specs()})
    model.to(dev).eval()
    return model, diff
def stepper(model, diff, B, seed, stream):
    kw = bench.make_window(B, 16, 64, 4, seed, dev)
    with torch.cuda.stream(stream):
        st = bench.Stepper(model, diff, kw, seed=seed)
    st.stream = stream.cuda_stream
    return st
nts = 250
def run(steppers, steps=20, warmup=5):
    for i in range(warmup):
        for s in steppers: s.step(nts - 1 - i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for s in steppers: s.step(nts - 1 - warmup - i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
m1, d1 = make()
s0 = torch.cuda.Stream(device=dev)
one = stepper(m1, d1, 8, 5, s0)
for rep in range(2):
    print("one engine, B=8, one stream: %.3f ms/step" % run([one]), flush=True)
m2, d2 = make()
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ha = stepper(m1, d1, 4, 6, sa)
hb = stepper(m2, d2, 4, 7, sb)
for rep in range(2):
    print("two engines, B=4 each, two streams: %.3f ms per step of both (= 8 clips)" % run([ha, hb]), flush=True)
    print("   the same two half-batches one after the other on one stream's worth of time: %.3f + %.3f ms" % (run([ha]), run([hb])), flush=True)
