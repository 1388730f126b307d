import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import video_diffusion_amd as vda
from test_gpu_engine import _oracle, _rand_window, kwargs_of
for over in [dict(), dict(use_rpe_net=False), dict(use_spatial_encoding=False), dict(attention_resolutions="1")]:
    cfg = {**vda.video_model_and_diffusion_defaults(), **dict(T=20, image_size=32, num_channels=32, num_res_blocks=1, rp_alpha=20, rp_beta=20, rp_gamma=20, timestep_respacing="ddim50"), **over}
    model, diff, ora = _oracle(cfg)
    for (B, T, n_obs) in [(1, 1, 0), (1, 1, 1), (2, 1, 0), (1, 2, 0), (1, 2, 1)]:
        c = _rand_window(B, T, 32, n_obs, seed=3)
        t = torch.tensor([31] * B)
        kw = {k: v for k, v in c.items() if k not in ("x", "observed_frames")}
        want = ora.eps(c["x"], t, kw)
        got, _ = diff._wrap_model(model)(c["x"].cuda(), t.cuda(), **kwargs_of(c))
        print(over, (B, T, n_obs), "max err %.3e" % (got.cpu() - want).abs().max().item(), flush=True)
