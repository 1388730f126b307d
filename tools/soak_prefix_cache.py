"""Whole windows (2 x 250 steps, B = 8, Tw = 16, the default model) on the window executor with and without the prefix cache:
same seed, same Philox streams -- steps/s of both and the distance between the two videos after 500 chained stochastic steps."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import video_diffusion_amd as vda
from video_diffusion_amd.video_sample import infer_video
cfg = vda.video_model_and_diffusion_defaults()
cfg.update(T=16, image_size=64, rp_alpha=16, rp_beta=16, rp_gamma=16, timestep_respacing="ddim250")
model, diff = vda.create_video_model_and_diffusion(**cfg)
model.load_state_dict({k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in model.param_specs()})
model.to("cuda").eval()
batch = torch.rand(8, 28, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1
outs = {}
for pc in (False, True, False, True):
    torch.manual_seed(3)
    torch.cuda.synchronize()
    t0 = time.time()
    out, _ = infer_video("autoreg", model, diff, batch.cuda(), 16, 4, 12, executor="graph", prefix_cache=pc)   # 2 windows x 250 steps
    torch.cuda.synchronize()
    dt = time.time() - t0
    model.check_device_errors()
    outs[pc] = out
    print("prefix_cache=%s: 2 windows x 250 steps in %.2f s = %.2f steps/s; finite %s; latent std %.3f" % (
        pc, dt, 500 / dt, np.isfinite(out).all(), out[:, 4:].std()), flush=True)
d = np.abs(outs[True] - outs[False])
print("cached vs uncached video after 2 x 250 steps: max |d| = %.3e, mean |d| = %.3e" % (d.max(), d.mean()))
