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
torch.manual_seed(3)
t0 = time.time()
out, _ = infer_video("autoreg", model, diff, batch.cuda(), 16, 4, 12)      # 2 windows x 250 steps, B=8, Tw=16
torch.cuda.synchronize()
dt = time.time() - t0
model.check_device_errors()
print("windows 2 x 250 steps in %.1f s = %.1f steps/s; finite %s; range [%.3f, %.3f]; latent std %.3f" % (
    dt, 500 / dt, np.isfinite(out).all(), out.min(), out.max(), out[:, 4:].std()))
