#!/usr/bin/env python3
"""Golden vectors for the SURVEY 8(f) rows, produced by IMPORTING the reference in this container:

  * scripts/video_sample_full.py infer_video (the vertical + horizontal sampler, :50-323) on the tiny model, CPU
  * improved_diffusion/test_util.py get_model_results_path / get_eval_run_identifier (:65-132) naming rules

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 /root/repo/tools/gen_golden_full.py

Third-party modules the reference imports but never reaches on these paths (lpips, imageio, blobfile/mpi4py via
dist_util, the dataset loaders) are replaced by empty stand-in modules so the import succeeds; nothing of the
reference is copied.  Writes tests/golden/full_sampler_tiny.npz and tests/golden/eval_paths.json.
"""
import importlib.util
import json
import logging
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"

spec = importlib.util.spec_from_file_location("weights_init", os.path.join(REPO, "video-diffusion_amd", "weights_init.py"))
weights_init = importlib.util.module_from_spec(spec)
spec.loader.exec_module(weights_init)

lp = types.ModuleType("lpips")
lp.LPIPS = type("LPIPS", (torch.nn.Module,), {})
lp.normalize_tensor = lambda x: x
sys.modules["lpips"] = lp
sys.modules["imageio"] = types.ModuleType("imageio")
du = types.ModuleType("improved_diffusion.dist_util")
du.load_state_dict = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not used"))
sys.modules["improved_diffusion.dist_util"] = du
ds = types.ModuleType("improved_diffusion.image_datasets")
for n in ("get_test_dataset", "get_train_dataset", "get_variable_length_dataset"):
    setattr(ds, n, None)
sys.modules["improved_diffusion.image_datasets"] = ds

import improved_diffusion  # noqa: E402
improved_diffusion.dist_util = du
improved_diffusion.image_datasets = ds
from improved_diffusion import script_util as su  # noqa: E402
from improved_diffusion import test_util as tu  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_video_sample_full", os.path.join(REF, "scripts", "video_sample_full.py"))
vsf = importlib.util.module_from_spec(spec)
spec.loader.exec_module(vsf)


def tiny_cfg(**over):
    d = su.video_model_and_diffusion_defaults()
    d.update(T=4, image_size=32, num_channels=32, num_res_blocks=1, rp_alpha=4, rp_beta=4, rp_gamma=4,
             timestep_respacing="ddim5")
    d.update(over)
    return d


def gen_full_sampler():
    cfg = tiny_cfg()
    model, diff = su.create_video_model_and_diffusion(**cfg)
    sd = {k: torch.from_numpy(weights_init.synth_param(k, tuple(v.shape))) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.eval()
    B, T, obs_len, max_frames, step = 2, 6, 2, 4, 1
    g = torch.Generator().manual_seed(21)
    batch = torch.rand(B, T, 3, 32, 32, generator=g) * 2 - 1
    out = {}
    for tag, vertical, obs_frames in (("v2_xtm1", 2, "x_t_minus_1"), ("v0_x0", 0, "x_0"), ("v5_x0", 5, "x_0")):
        vsf.args = Namespace(vertical_steps=vertical, observed_frames=obs_frames, save_all_timesteps=False)
        vsf.logger = logging.getLogger("ref_full")
        torch.manual_seed(1234)                        # p_sample draws th.randn_like from the global CPU generator
        samples, _ = vsf.infer_video("autoreg", model, diff, batch, max_frames, obs_len, step, None, use_gradient_method=False)
        out[f"samples_{tag}"] = samples.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "full_sampler_tiny.npz"), batch=batch.numpy(), noise_seed=np.array(1234),
                        cfg_json=np.array(json.dumps(cfg)), B=B, T=T, obs_length=obs_len, max_frames=max_frames, step_size=step,
                        **out)


def gen_eval_paths():
    cases = []
    base = dict(use_ddim=False, timestep_respacing="", eval_dir=None, checkpoint_path="/scratch/vd/saeids-checkpoints/abcdefg/ema_0.9999_550000.pt")
    for over in (dict(), dict(use_ddim=True), dict(timestep_respacing="ddim250"), dict(use_ddim=True, timestep_respacing="250"),
                 dict(checkpoint_path="/data/checkpoints/run7/sub/model_100.pt", timestep_respacing="ddim50"),
                 dict(eval_dir="/tmp/my_eval")):
        a = Namespace(**{**base, **over})
        for postfix in ("", "_x"):
            cases.append(dict(kind="model_results_path", args=vars(a), postfix=postfix,
                              expect=str(tu.get_model_results_path(a, postfix=postfix))))
    ident = dict(inference_mode="autoreg", max_frames=20, step_size=7, T=300, obs_length=36)
    for over in (dict(), dict(optimality="linspace-t"), dict(optimality=None), dict(dataset_partition="train"),
                 dict(dataset_partition="test"), dict(use_gradient_method=True), dict(use_gradient_method=False),
                 dict(override_dataset="carla"), dict(optimality="x", dataset_partition="train", use_gradient_method=True,
                                                       override_dataset="mazes"),
                 dict(inference_mode="hierarchy-2", max_frames=16, step_size=4, T=16, obs_length=4)):
        a = Namespace(**{**ident, **over})
        for postfix in ("", "_p"):
            cases.append(dict(kind="eval_run_identifier", args=vars(a), postfix=postfix,
                              expect=tu.get_eval_run_identifier(a, postfix=postfix)))
    json.dump(cases, open(os.path.join(OUT, "eval_paths.json"), "w"), indent=1)


def gen_more_schedulers():
    """the goal-directed / visualisation / frameskip schedules (inference_util.py:534-776), integer logic."""
    import contextlib
    import io
    from improved_diffusion import inference_util as iu
    cases = []
    for mode, args in [("goal-directed-autoreg", (30, 4, 10, 3)), ("goal-directed-autoreg", (64, 8, 20, 5)),
                       ("goal-directed-mixed", (40, 6, 12, 4)), ("goal-directed-mixed", (64, 8, 20, 5)),
                       ("goal-directed-hierarchy-2", (100, 10, 20, 5)), ("goal-directed-hierarchy-2", (64, 8, 16, 4)),
                       ("ho-et-al-for-vis", (64, 0, 16, 8)), ("ho-et-al-for-vis", (40, 0, 16, 8)),
                       ("baby-cond-ho-et-al-for-vis", (30, 4, 7, 3)),
                       ("google", (64, 8, 16, 8)), ("google", (100, 4, 16, 8)), ("google", (37, 5, 16, 8)),
                       ("like-google", (64, 8, 16, 8)), ("like-google", (50, 5, 12, 4)), ("like-google", (30, 1, 10, 3))]:
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                it = iter(iu.inference_strategies[mode](video_length=args[0], num_obs=args[1], max_frames=args[2],
                                                        step_size=args[3]))
                seq = []
                for o, l in it:
                    seq.append([[int(i) for i in o], [int(i) for i in l]])
                    if len(seq) > 400:
                        raise RuntimeError("does not terminate")
            cases.append(dict(mode=mode, args=list(args), seq=seq))
        except Exception as e:  # noqa: BLE001 -- record what the reference does, including failures
            cases.append(dict(mode=mode, args=list(args), error=type(e).__name__))
    json.dump(dict(modes=sorted(iu.inference_strategies.keys()), cases=cases), open(os.path.join(OUT, "schedulers_more.json"), "w"))


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen_more_schedulers()
    if "--schedulers-only" in sys.argv:
        sys.exit(0)
    gen_eval_paths()
    gen_full_sampler()
    for f in ("eval_paths.json", "full_sampler_tiny.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))
