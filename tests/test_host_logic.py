"""CPU: the product's host logic (integer schedulers, respacing, float64 tables, parameter table)
against the reference's golden vectors, and the C-ABI library's exported surface.
No GPU compute call is made here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import video_diffusion_amd as vda
from helpers import ROOT, load_json
from video_diffusion_amd import _lib
from video_diffusion_amd.inference_util import inference_strategies
from video_diffusion_amd.respace import space_timesteps
from video_diffusion_amd.script_util import create_gaussian_diffusion, video_model_and_diffusion_defaults


def test_space_timesteps_bit_exact():
    for c in load_json("space_timesteps.json"):
        if "error" in c:
            with pytest.raises(ValueError, match=re.escape(c["error"])):
                space_timesteps(c["n"], c["spec"])
        else:
            assert sorted(space_timesteps(c["n"], c["spec"])) == c["steps"], c["spec"]


@pytest.mark.parametrize("tag", ["linear1000_ddim250", "linear1000_full", "linear1000_ddim50",
                                 "cosine1000_ddim100", "linear1000_ddim5_small"])
def test_schedule_tables_bit_exact(tag):
    rec = load_json(f"schedule_{tag}.json")
    d = create_gaussian_diffusion(rescale_timesteps=True, rescale_learned_sigmas=True, **rec["kw"])
    assert list(d.timestep_map) == rec["timestep_map"]
    assert d.num_timesteps == rec["num_timesteps"]
    for name in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                 "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                 "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                 "posterior_mean_coef2"]:
        want = np.array([float.fromhex(h) for h in rec[name]])
        assert np.array_equal(getattr(d, name), want), name
    tab = d._device_tables()
    assert tab.shape == (12, d.num_timesteps) and tab.dtype == np.float32      # VD_NTAB rows (include/vd_amd.h)
    assert np.array_equal(tab[11], (1.0 - d.betas).astype(np.float32))          # alphas: the guidance weight


def test_schedulers_match_reference_sequences():
    rec = load_json("schedulers.json")
    seen = 0
    for c in rec["cases"]:
        if c["mode"] not in inference_strategies:
            continue
        vl, no, mf, ss = c["args"]
        it = iter(inference_strategies[c["mode"]](video_length=vl, num_obs=no, max_frames=mf, step_size=ss,
                                                  optimal_schedule_path=None))
        if "error" in c:
            with pytest.raises(AssertionError):
                list(it)
            continue
        got = [[[int(i) for i in o], [int(i) for i in l]] for o, l in it]
        assert got == c["seq"], (c["mode"], c["args"])
        seen += 1
    assert seen >= 10


def test_scheduler_known_answers_survey_appendix_d():
    def run(mode, *a):
        return list(inference_strategies[mode](video_length=a[0], num_obs=a[1], max_frames=a[2], step_size=a[3]))
    s = run("autoreg", 16, 4, 10, 1)
    assert len(s) == 12 and s[0] == ([0, 1, 2, 3], [4]) and s[-1] == (list(range(6, 15)), [15])
    assert run("independent", 16, 4, 16, 12) == [([0, 1, 2, 3], list(range(4, 16)))]
    s = run("autoreg", 300, 36, 20, 7)
    assert len(s) == 38 and s[-1] == (list(range(282, 295)), list(range(295, 300)))
    s = run("exp-past", 16, 4, 16, 4)
    assert [list(map(int, o)) for o, _ in s] == [[3, 2, 1, 0], [7, 6, 4, 5, 3, 2, 1, 0],
                                                  [11, 10, 8, 9, 7, 6, 5, 4, 3, 2, 1, 0]]
    s = run("autoreg", 16, 0, 10, 1)
    assert s[0] == ([], list(range(10))) and s[1] == (list(range(1, 10)), [10])


def test_scheduler_rejects_unfinished_conditioning():
    class Bad(vda.inference_util.InferenceStrategyBase):
        def next_indices(self):
            return [9], [4]
    with pytest.raises(AssertionError, match="not generated yet"):
        next(iter(Bad(16, 4, 10, 1)))


# ---------------------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vd_amd.h")).read()
    declared = set(re.findall(r"\b(vd_[a-z_0-9]+)\s*\(", header))
    L = _lib.lib()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/vd_amd.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == declared
    assert b"gfx950" in L.vd_version()
    assert L.vd_source_sha().decode() == _lib.source_sha()      # the loaded binary IS the sources beside it (content hash, not mtimes)


def _cfg(tag):
    d = video_model_and_diffusion_defaults()
    if tag.startswith("tiny"):
        d.update(T=4, image_size=32, num_channels=32, num_res_blocks=1, rp_alpha=4, rp_beta=4, rp_gamma=4)
        if tag == "tiny_table":
            d.update(use_rpe_net=False)
    else:
        d.update(T=16, image_size=int(tag[len("default"):]), rp_alpha=16, rp_beta=16, rp_gamma=16)
    return d


@pytest.mark.parametrize("tag", ["tiny", "tiny_table", "default64", "default128"])
def test_param_table_matches_reference_state_dict(tag):
    want = [(k, tuple(s)) for k, s in load_json("param_specs.json")[tag]]
    got = vda.param_specs(_cfg(tag))
    assert got == want        # names, shapes AND order of the reference's state_dict
    if tag == "default64":
        assert sum(int(np.prod(s)) for _, s in got) == 116052099


def test_factory_error_behaviour():
    d = _cfg("tiny")
    with pytest.raises(ValueError, match="unsupported image size"):
        vda.create_video_model_and_diffusion(**{**d, "image_size": 48})
    with pytest.raises(AssertionError):                       # bucket params missing (unet.py:423-427)
        vda.create_video_model_and_diffusion(**{**d, "rp_alpha": None, "rp_beta": None, "rp_gamma": None})
    with pytest.raises(ValueError, match="cannot create exactly"):
        vda.create_video_model_and_diffusion(**{**d, "timestep_respacing": "ddim300"})
    # do_cond_marg=False: the reference hands cond_emb_type to UNetVideoModel -> UNetModel.__init__, which does not take it
    # (script_util.py:275-300; probed on the imported reference by tools/gen_golden_r4.py): same exception, same text
    with pytest.raises(TypeError, match=r"UNetModel.__init__\(\) got an unexpected keyword argument 'cond_emb_type'"):
        vda.create_video_model_and_diffusion(**{**d, "do_cond_marg": False})
    from video_diffusion_amd import gaussian_diffusion as gd
    _, dx = vda.create_video_model_and_diffusion(**{**d, "predict_xstart": True})       # script_util.py:429-431
    assert dx.model_mean_type == gd.ModelMeanType.START_X
    model, diff = vda.create_video_model_and_diffusion(**{**d, "timestep_respacing": "ddim250"})
    assert diff.num_timesteps == 250 and diff.model_mean_type == gd.ModelMeanType.EPSILON
    with pytest.raises(RuntimeError, match="Missing key"):
        model.load_state_dict({})
    sd = {k: vda.weights_init.synth_param(k, s) for k, s in model.param_specs()}
    sd["time_embed.0.bias"] = sd["time_embed.0.bias"][:-1]
    with pytest.raises(RuntimeError, match="size mismatch for time_embed.0.bias"):
        model.load_state_dict(sd)
    import torch
    x = torch.zeros(1, 4, 3, 32, 32)
    with pytest.raises(RuntimeError, match="no CPU path"):    # the product path never falls back to the CPU
        model(x, torch.zeros(1), x0=x, obs_mask=x[:, :, :1, :1, :1], latent_mask=x[:, :, :1, :1, :1],
              kinda_marg_mask=x[:, :, :1, :1, :1], observed_frames="x_0")


def test_engine_reports_errors_not_crashes():
    L = _lib.lib()
    cfg = _lib.VdConfig()
    cfg.image_size, cfg.num_channels, cfg.num_res_blocks, cfg.num_heads = 48, 32, 1, 4
    h = ctypes.c_void_p()
    assert L.vd_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"unsupported image size" in L.vd_last_error()
    nbytes = ctypes.c_longlong()
    model, _ = vda.create_video_model_and_diffusion(**_cfg("tiny"))
    assert L.vd_workspace_bytes(model._handle, 2, 4, ctypes.byref(nbytes)) == 0 and nbytes.value > 0
    # compute entry points refuse to run without weights (no silent fallback)
    assert L.vd_unet_forward(model._handle, 2, 4, None, None, None, None, None, None, None, 0, None, None) == -1
    assert b"weights not set" in L.vd_last_error()


def test_eval_dir_naming_matches_reference(tmp_path):
    """improved_diffusion/test_util.py:65-132 naming rules against strings produced by the reference
    (tools/gen_golden_full.py): results/<subpath>/<stem>[_ddim][_respace<X>] and the run identifier."""
    from argparse import Namespace

    from helpers import load_json
    from video_diffusion_amd import test_util as tu
    cases = load_json("eval_paths.json")
    assert len(cases) >= 30
    for c in cases:
        a = Namespace(**c["args"])
        if c["kind"] == "model_results_path":
            assert str(tu.get_model_results_path(a, postfix=c["postfix"])) == c["expect"], c
        else:
            assert tu.get_eval_run_identifier(a, postfix=c["postfix"]) == c["expect"], c
    # '*latest' checkpoints get their training step appended (read from the file)
    ck = tmp_path / "my-checkpoints" / "run1" / "ema_latest.pt"
    ck.parent.mkdir(parents=True)
    import torch
    torch.save({"step": 4321, "state_dict": {}, "config": {}}, ck)
    a = Namespace(use_ddim=True, timestep_respacing="ddim50", eval_dir=None, checkpoint_path=str(ck))
    assert str(tu.get_model_results_path(a)) == "results/run1/ema_latest_4321_ddim_respaceddim50"
    # the lock sits next to the file and is released afterwards
    f = tmp_path / "model_config.json"
    with tu.Protect(f):
        assert (tmp_path / "model_config.json.lock").exists()
        f.write_text("{}")
    with tu.Protect(f, timeout=0.5):
        pass


def test_remaining_schedulers_match_reference():
    """goal-directed / visualisation / frameskip schedules (inference_util.py:534-776) against the
    sequences -- and the failures -- of the reference itself (tools/gen_golden_full.py)."""
    import contextlib
    import io

    from helpers import load_json
    from video_diffusion_amd import inference_util as iu
    rec = load_json("schedulers_more.json")
    assert len(rec["cases"]) >= 15
    for c in rec["cases"]:
        L, n_obs, max_frames, step = c["args"]
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                it = iter(iu.inference_strategies[c["mode"]](video_length=L, num_obs=n_obs, max_frames=max_frames, step_size=step))
                seq = []
                for o, l in it:
                    seq.append([[int(i) for i in o], [int(i) for i in l]])
                    assert len(seq) <= 400
            assert "seq" in c and seq == c["seq"], (c["mode"], c["args"])
        except AssertionError as e:
            if "error" not in c:
                raise
            assert c["error"] == "AssertionError", (c, e)
    # every non-adaptive mode of the reference registry exists here
    missing = [m for m in rec["modes"] if not m.startswith("adaptive") and m not in iu.inference_strategies]
    assert not missing, missing


def test_adaptive_schedulers_match_reference_sequences():
    """adaptive-autoreg / adaptive-hierarchy-N with distance='l2' (inference_util.py:137-229,421-531): per-item observed
    lists picked by farthest-point selection on the frames themselves, against what the imported reference produced on
    the same seeded videos (tools/gen_golden_adaptive.py).  Where the reference never terminates (hierarchy with no
    observed frames: its backwards search for a finished frame runs below index 0 forever) the mirror raises."""
    import torch
    rec = load_json("schedulers_adaptive.json")
    seen = 0
    for c in rec["cases"]:
        T, n_obs, max_frames, step = c["args"]
        g = torch.Generator().manual_seed(c["seed"])
        v = torch.rand(c["B"], T, 3, 4, 4, generator=g) * 2 - 1
        it = iter(inference_strategies[c["mode"]](distance="l2", video_length=T, num_obs=n_obs, max_frames=max_frames,
                                                  step_size=step, optimal_schedule_path=None))
        want = c.get("seq", c.get("seq_before_error"))
        got = []
        try:
            while len(got) < 200:
                it.set_videos(v)
                try:
                    obs, lat = next(it)
                except StopIteration:
                    break
                got.append([[[int(i) for i in o] for o in obs], [[int(i) for i in l] for l in lat]])
            assert "error" not in c, (c["mode"], c["args"])
        except RuntimeError:
            assert c.get("error") == "TimeoutError", (c["mode"], c["args"])
        assert got == want, (c["mode"], c["args"], got[:2], want[:2])
        seen += 1
    assert seen == 7
    with pytest.raises(NotImplementedError):
        it = iter(inference_strategies["adaptive-autoreg"](distance="lpips", video_length=8, num_obs=2, max_frames=4, step_size=2))
        it.set_videos(torch.zeros(1, 8, 3, 4, 4))
        next(it)


# ------------------------------------------------------------------------------------------------------------------
# The sampling job around infer_video (scripts/video_sample.py:192-239, 570-590) and the NLL window builder
class _CountingDiffusion:
    """Stand-in sampler for the job tests (no GPU here): counts its p_sample calls, leaves x unchanged."""
    num_timesteps = 3

    def __init__(self):
        self.calls = 0

    def p_sample(self, model, x, t, **kw):
        self.calls += 1
        return {"sample": x}


def _job_args(tmp_path, **over):
    from argparse import Namespace
    base = dict(checkpoint_path="", inference_mode="autoreg", T=6, max_frames=4, obs_length=2, step_size=2, batch_size=2,
                num_videos=5, timestep_respacing="ddim5", observed_frames="x_0", image_size=32, num_channels=32,
                num_res_blocks=1, seed=0, adaptive_distance="l2", executor="eager", eval_dir=str(tmp_path / "out"),
                out_dir=None, use_ddim=False, sample_idx=None, num_samples=1, indices=None, task_id=None,
                subset_size=None, optimality=None, use_gradient_method=False, save_all_timesteps=False, videos=None,
                synthetic=True)
    base.update(over)
    return Namespace(**base)


def _run_job(args):
    import torch
    from video_diffusion_amd import video_sample
    diff = _CountingDiffusion()

    def create(**kw):
        model, _ = vda.create_video_model_and_diffusion(**kw)
        return model, diff

    out = video_sample.run(args, create=create, device=torch.device("cpu"))
    return out, diff


def test_sampling_job_decides_todo_before_sampling_and_resumes_without_a_step(tmp_path):
    """video_sample.py:207-239: per batch and sample index the names are formed and the disk is looked at BEFORE
    infer_video; a finished (batch, sample) costs no denoise step; a partially finished batch is sampled again but
    only its missing files are written."""
    args = _job_args(tmp_path, num_samples=2)
    out, diff = _run_job(args)
    files = sorted(os.listdir(out / "samples"))
    assert files == [f"sample_{i:04d}-{s}.npy" for i in range(5) for s in range(2)]
    windows = 2                                                           # autoreg: frames 2-3, 4-5 with step_size 2
    assert diff.calls == 3 * 2 * windows * _CountingDiffusion.num_timesteps     # 3 batches x 2 samples
    a = np.load(out / "samples" / "sample_0003-1.npy")
    assert a.dtype == np.uint8 and a.shape == (6, 3, 32, 32)
    out2, diff2 = _run_job(_job_args(tmp_path, num_samples=2))
    assert out2 == out and diff2.calls == 0                              # the resumed job repeats nothing
    stamp = os.path.getmtime(out / "samples" / "sample_0002-0.npy")
    os.remove(out / "samples" / "sample_0003-0.npy")                     # batch (2, 3), sample 0 is now half done
    out3, diff3 = _run_job(_job_args(tmp_path, num_samples=2))
    assert diff3.calls == windows * _CountingDiffusion.num_timesteps
    assert os.path.exists(out / "samples" / "sample_0003-0.npy")
    assert os.path.getmtime(out / "samples" / "sample_0002-0.npy") == stamp        # its finished neighbour is left alone


def test_sampling_job_indices_task_id_subset_and_sample_idx(tmp_path):
    """video_sample.py:570-590: --indices as given; --task_id = one batch worth of consecutive items; --subset_size = the
    first N; --sample_idx overrides --num_samples; file names carry the DATASET index (dataset_idx_translate, :202-203)."""
    from video_diffusion_amd import video_sample as vs
    out_a, _ = _run_job(_job_args(tmp_path / "a", indices=[4, 1, 3], sample_idx=7, num_samples=3))
    assert sorted(os.listdir(out_a / "samples")) == ["sample_0001-7.npy", "sample_0003-7.npy", "sample_0004-7.npy"]
    out_b, _ = _run_job(_job_args(tmp_path / "b", task_id=1))
    assert sorted(os.listdir(out_b / "samples")) == ["sample_0002-0.npy", "sample_0003-0.npy"]
    out, _ = _run_job(_job_args(tmp_path / "c", subset_size=3))
    assert sorted(os.listdir(out / "samples")) == ["sample_0000-0.npy", "sample_0001-0.npy", "sample_0002-0.npy"]
    with pytest.raises(AssertionError):
        vs.resolve_indices(_job_args(tmp_path, task_id=0, subset_size=2), 5)
    # a video is a function of its dataset index alone: the same item through another batch composition gives the same file
    a = np.load(out_a / "samples" / "sample_0003-7.npy")
    b = np.load(out_b / "samples" / "sample_0003-0.npy")
    assert np.array_equal(a[:2], b[:2])                                  # the observed frames pass through untouched


def test_sampling_job_reads_videos_from_a_file_and_names_the_run_from_the_options_as_given(tmp_path, monkeypatch):
    """--videos file.npy (uint8, the format the tool writes) beside --synthetic; --T / --max_frames unset: T comes from the
    dataset, max_frames from the model config, and the run directory says 'None' for both, as the reference's does
    (its eval_dir is formed right after parse_args, video_sample.py:530-533)."""
    import torch
    vids = (np.random.RandomState(0).rand(3, 6, 3, 32, 32) * 255).astype(np.uint8)
    np.save(tmp_path / "v.npy", vids)
    cfg = vda.video_model_and_diffusion_defaults()
    cfg.update(T=4, image_size=32, num_channels=32, num_res_blocks=1, rp_alpha=4, rp_beta=4, rp_gamma=4)
    model, _ = vda.create_video_model_and_diffusion(**cfg)
    sd = {k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in model.param_specs()}
    cfg["max_frames"] = 4                                                 # a training option the checkpoint's config records (video_train.py:126)
    ck = tmp_path / "checkpoints" / "r" / "ema_100.pt"
    ck.parent.mkdir(parents=True)
    torch.save({"state_dict": sd, "config": cfg, "step": 100}, ck)
    monkeypatch.chdir(tmp_path)
    args = _job_args(tmp_path, checkpoint_path=str(ck), videos=str(tmp_path / "v.npy"), T=None, max_frames=None, eval_dir=None,
                     use_ddim=True)
    out, diff = _run_job(args)
    assert str(out) == "results/r/ema_100_ddim_respaceddim5/autoreg_None_2_None_2"
    assert args.T == 6 and args.max_frames == 4 and diff.calls > 0
    got = np.load(tmp_path / out / "samples" / "sample_0002-0.npy")
    assert np.abs(got[:2].astype(int) - vids[2, :2].astype(int)).max() <= 1        # uint8 -> [-1, 1] -> truncated uint8
    mc = json.load(open(tmp_path / out / "model_config.json"))
    assert mc["use_ddim"] is True and mc["timestep_respacing"] == "ddim5" and mc["max_frames"] == 4


def test_nll_window_table_equals_the_per_item_construction():
    """scripts/video_nll.py:150-165 builds x0 / masks / frame_indices item by item; the mirror builds them as one gather."""
    import torch
    from video_diffusion_amd.video_nll import _window_table, run_bpd_evaluation
    obs = [[0, 1], [5, 2, 0], []]
    lat = [[2, 3, 7], [3], [4, 6]]
    batch = torch.randn(3, 8, 3, 4, 4)
    seen = {}

    class Dev:
        device = torch.device("cpu")

    class Diff:
        def calc_bpd_loop_subsampled(self, model, x0, clip_denoised, model_kwargs, latent_mask, t_seq):
            seen.update(model_kwargs, x_start=x0, lm=latent_mask)
            return {"total_bpd": torch.ones(3), "vb": torch.ones(3, 5)}

    out = run_bpd_evaluation(Dev(), Diff(), batch, True, obs, lat)
    F = 5
    x0 = torch.zeros(3, F, 3, 4, 4)
    om, lm, fi = torch.zeros(3, F, 1, 1, 1), torch.zeros(3, F, 1, 1, 1), torch.zeros(3, F, dtype=torch.long)
    for i, (o, l) in enumerate(zip(obs, lat)):
        for s, f in enumerate(o + l):
            x0[i, s] = batch[i, f]
            fi[i, s] = f
            (om if s < len(o) else lm)[i, s] = 1
    assert torch.equal(seen["x0"], x0) and torch.equal(seen["x_start"], x0) and torch.equal(seen["x_t_minus_1"], x0)
    assert torch.equal(seen["obs_mask"], om) and torch.equal(seen["latent_mask"], lm) and torch.equal(seen["lm"], lm)
    assert torch.equal(seen["frame_indices"], fi) and seen["frame_indices"].dtype == torch.long
    assert not seen["kinda_marg_mask"].any() and seen["observed_frames"] == "x_0"
    assert out["total_bpd"].tolist() == [5.0] * 3 and out["vb"].tolist() == [25.0] * 3     # x window length; vb summed over t
    t, o_, l_ = _window_table([[1]], [[2, 3]], 2)                        # a batch longer than the index lists: empty rows
    assert t.tolist() == [[1, 2, 3], [0, 0, 0]] and o_.tolist() == [[True, False, False], [False] * 3]
    assert l_.tolist() == [[False, True, True], [False] * 3]


def test_integration_import_block_resolves_every_name_the_script_uses(tmp_path):
    """INTEGRATION.md A shows the import swap a maintainer makes in scripts/video_sample.py:12-19.  The 'after' half is executed
    as written, and every attribute the reference's scripts take from the swapped modules (tests/golden/script_imports.json,
    tools/gen_script_imports.py) must resolve in the mirror -- a block that leaves `dist_util` or `test_util` unbound is a NameError
    at the script's first use of them.  `dist_util.load_state_dict` must open a checkpoint file like `torch.load`."""
    import torch
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## A."):text.index("## B.")]
    block = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    after = block[block.index("# after"):]
    assert "improved_diffusion" not in after
    ns = {}
    exec(compile(after, "INTEGRATION.md#A", "exec"), ns)
    used = load_json("script_imports.json")
    for script, mods in used.items():
        for mod, names in mods.items():
            for name in names:
                holder = ns if mod == "script_util" else ns[mod]
                assert (name in holder) if isinstance(holder, dict) else hasattr(holder, name), (script, mod, name)
    ck = tmp_path / "c.pt"
    torch.save({"state_dict": {"w": torch.arange(3.0)}, "config": {"T": 4}, "step": 5}, ck)
    data = ns["dist_util"].load_state_dict(str(ck), map_location="cpu")
    assert data["step"] == 5 and torch.equal(data["state_dict"]["w"], torch.arange(3.0))
    assert ns["dist_util"].dev().type in ("cpu", "cuda") and ns["dist_util"].get_world_size() == 1
    ref = "/root/reference/scripts"
    if os.path.isdir(ref):                               # in the build container: the fixture still says what the scripts say
        for script, mods in used.items():
            src = open(os.path.join(ref, script)).read()
            for mod in ("dist_util", "inference_util", "test_util"):
                assert sorted(set(re.findall(rf"\b{mod}\.([A-Za-z_][A-Za-z0-9_]*)", src))) == mods.get(mod, []), (script, mod)


def test_timing_only_kernel_builds_still_compile(tmp_path):
    """The hot kernels carry timing-only ablation hooks (results WRONG, never in the product library; tools/build_variant.sh builds them
    for same-box A/B runs: -DVD_R64_ABL, -DVD_GS_SKIP, -DVD_ATTN_ABL, the cycle-stamp builds).  Nothing else compiles those paths, so
    this does: hipcc for gfx950, device code only, all variants in parallel."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    cs = os.path.join(ROOT, "video-diffusion_amd", "csrc")
    variants = [("conv_wino_r64.hip", ["-DVD_R64_ABL=34", "-DVD_WINO_TIMING"]),
                ("conv_wino_r64.hip", ["-DVD_R64_ABL=29"]),
                ("conv_wino_z128.hip", ["-DVD_WINO_TIMING", *_lib.SOURCE_FLAGS.get("conv_wino_z128.hip", [])]),
                ("gemm_split.hip", ["-DVD_GS_SKIP=15", "-DVD_GS_TIMING"]),
                ("attn_spatial.hip", ["-DVD_ATTN_ABL=7"]),
                ("attn_temporal.hip", ["-DVD_ATT_TIMING"])]

    def build(i):
        src, flags = variants[i]
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *flags,
                            "--cuda-device-only", "-c", os.path.join(cs, src), "-o", str(tmp_path / f"v{i}.o")], capture_output=True, text=True)
        return src, flags, r.returncode, r.stderr[-1500:]

    with ThreadPoolExecutor(max_workers=6) as pool:
        for src, flags, rc, err in pool.map(build, range(len(variants))):
            assert rc == 0, (src, flags, err)


def test_sampling_job_run_directory_follows_the_dataset_options(tmp_path):
    """--dataset_partition train -> 'trainset_' prefix, variable_length -> a subdirectory (video_sample.py:603-604); --override_dataset ->
    '<name>_' prefix and model_config.json's `dataset` (:555-556); --use_gradient_method -> 'gradientmethod_' (test_util.py:111-132)."""
    out, _ = _run_job(_job_args(tmp_path / "a", dataset_partition="train", override_dataset="carla", num_videos=1, indices=[0]))
    assert out.name == "carla_trainset_autoreg_4_2_6_2"
    assert json.load(open(out / "model_config.json"))["dataset"] == "carla"
    out, _ = _run_job(_job_args(tmp_path / "b", dataset_partition="variable_length", num_videos=1, inference_mode="independent"))
    assert out.name == "variable_length" and out.parent.name == "independent_4_2_6_2"
    # --optimality reads <eval_dir>/optimal_schedule.pt (video_sample.py:199-200), which the reference's video_optimal_schedule.py writes: without it, its error
    with pytest.raises(FileNotFoundError, match="optimal_schedule.pt"):
        _run_job(_job_args(tmp_path / "c", num_videos=1, optimality="linspace-t", inference_mode="independent"))
    assert (tmp_path / "c" / "out" / "independent_optimal-linspace-t_4_2_6_2").is_dir()


def test_nll_job_scores_every_window_once_and_skips_finished_videos(tmp_path):
    """video_nll.run (scripts/video_nll.py:87-140,262-352) with a stand-in sampler: every video gets one pickle {metric: array over the
    schedule's windows} under elbos/, the frame-index lists are saved and re-checked, a second run scores nothing, --task_id selects one
    batch-sized block, an explicit --indices is refused as in the reference."""
    import pickle
    import torch
    from video_diffusion_amd import video_nll
    calls = []

    class Diff:
        num_timesteps = 3

        def calc_bpd_loop_subsampled(self, model, x0, clip_denoised, model_kwargs, latent_mask, t_seq):
            calls.append((tuple(x0.shape), model_kwargs["frame_indices"][0].tolist(), int(model_kwargs["obs_mask"][0].sum())))
            B = x0.shape[0]
            return {"total_bpd": torch.full((B,), float(len(calls))), "vb": torch.ones(B, 3)}

    def create(**kw):
        model, _ = vda.create_video_model_and_diffusion(**kw)
        return model, Diff()

    def args(**over):
        a = _job_args(tmp_path, num_videos=3, batch_size=2, clip_denoised=True, indices_path=None, **over)
        return a

    out = video_nll.run(args(), create=create, device=torch.device("cpu"))
    assert out.name == "autoreg_4_2_6_2" and sorted(os.listdir(out / "elbos")) == ["elbo_0_respaceddim5.pkl", "elbo_1_respaceddim5.pkl", "elbo_2_respaceddim5.pkl"]
    # autoreg(T=6, obs 2, max_frames 4, step 2): windows ([0,1],[2,3]) and ([2,3],[4,5]); two batches (2 + 1 videos) x two windows
    assert [c[1] for c in calls] == [[0, 1, 2, 3], [2, 3, 4, 5]] * 2 and [c[0][0] for c in calls] == [2, 2, 1, 1] and all(c[2] == 2 for c in calls)
    rec = pickle.load(open(out / "elbos" / "elbo_2_respaceddim5.pkl", "rb"))
    assert rec["total_bpd"].shape == (2,) and rec["total_bpd"].tolist() == [3.0 * 4, 4.0 * 4]        # x window length (4 frames)
    assert rec["vb"].tolist() == [12.0, 12.0]                                                          # summed over t, x window length
    obs_saved, lat_saved = torch.load(out / "frame_indices.pt")
    assert len(obs_saved) == 3 and obs_saved[0] == [[0, 1], [2, 3]] and lat_saved[2] == [[2, 3], [4, 5]]
    n = len(calls)
    video_nll.run(args(), create=create, device=torch.device("cpu"))
    assert len(calls) == n                                                                             # everything on disk: no network call
    os.remove(out / "elbos" / "elbo_1_respaceddim5.pkl")
    video_nll.run(args(task_id=0), create=create, device=torch.device("cpu"))                          # block #0 = items 0, 1: one file missing -> the batch again
    assert len(calls) == n + 2 and os.path.exists(out / "elbos" / "elbo_1_respaceddim5.pkl")
    with pytest.raises(IndexError):                                                                    # block #1 = items 2, 3; item 3 does not exist: `Subset`
        os.remove(out / "elbos" / "elbo_2_respaceddim5.pkl")                                           # fails on it in the reference as well
        video_nll.run(args(task_id=1), create=create, device=torch.device("cpu"))
    with pytest.raises(NotImplementedError):
        video_nll.run(args(indices=[0]), create=create, device=torch.device("cpu"))


def test_full_sampler_host_loop_write_back_against_a_plain_restatement():
    """video_sample_full.infer_video's host logic on CPU with a stand-in step (sample = x + 1 on latent slots, observed slots untouched): the
    vertical + horizontal loop nest, the per-item gathers / scatters of the adaptive branch and the all-timestep record must equal a plain
    item-by-item restatement of scripts/video_sample_full.py:88-323.  (The arithmetic of the step is the GPU tests' business.)"""
    import torch
    from video_diffusion_amd import inference_util as iu
    from video_diffusion_amd.video_sample_full import infer_video

    class Model:
        device = torch.device("cpu")

        def check_device_errors(self):
            pass

    class Diff:
        num_timesteps = 4

        def __init__(self):
            self.calls = []

        def p_sample(self, model, x, t, clip_denoised=True, model_kwargs=None, **kw):
            self.calls.append((int(t[0]), model_kwargs["observed_frames"], model_kwargs["frame_indices"].tolist()))
            lat = model_kwargs["latent_mask"]
            return {"sample": x + lat * (1.0 + 0.01 * model_kwargs["frame_indices"][:, :, None, None, None].float())}

    B, T, obs, mf, step = 2, 9, 3, 5, 2
    batch = torch.rand(B, T, 3, 4, 4, generator=torch.Generator().manual_seed(1)) * 2 - 1
    for mode, vertical in (("autoreg", 0), ("autoreg", 2), ("adaptive-autoreg", 1), ("hierarchy-2", 4)):
        d = Diff()
        got, every = infer_video(mode, Model(), d, batch, mf, obs, step, vertical_steps=vertical, observed_frames="x_t_minus_1",
                                 save_all_timesteps=True, adaptive_distance="l2")
        # plain restatement
        samples = torch.zeros_like(batch)
        samples[:, :obs] = batch[:, :obs]
        rec = torch.zeros(B, 4, T, 3, 4, 4)
        rec[:, :, :obs] = samples[:, :obs].unsqueeze(1)
        adaptive = "adaptive" in mode

        def passes(timesteps, record_rows):
            it = iter(iu.inference_strategies[mode](video_length=T, num_obs=obs, max_frames=mf, step_size=step, optimal_schedule_path=None,
                                                    **(dict(distance="l2") if adaptive else {})))
            while True:
                if adaptive:
                    it.set_videos(samples)
                try:
                    o, l = next(it)
                except StopIteration:
                    return
                for i in range(B):
                    oi, li = (o[i], l[i]) if adaptive else (o, l)
                    x = samples[i, list(oi) + list(li)].clone()
                    for r, ts in enumerate(timesteps):
                        x[len(oi):] += 1.0 + 0.01 * torch.tensor(li).float()[:, None, None, None]
                        if record_rows is not None:
                            rec[i, record_rows[r], li] = x[len(oi):]
                    samples[i, li] = x[len(oi):]

        steps = list(range(4))[::-1]
        if vertical:
            passes(steps[:vertical], list(range(vertical)))
        for k, ts in enumerate(steps[vertical:]):
            passes([ts], None)
            rec[:, vertical + k] = samples
        assert np.array_equal(got, samples.numpy()), (mode, vertical)
        assert np.array_equal(every, rec.numpy()), (mode, vertical)
        # vertical windows are sampled with x_0, horizontal ones with the option as given; timesteps high to low
        assert {c[1] for c in d.calls if c[0] >= 4 - vertical} <= {"x_0", "x_t_minus_1"} and d.calls[-1][0] == 0
        if vertical:
            assert d.calls[0][1] == "x_0"


def test_sampling_job_covers_every_selected_item_exactly_once_across_ranks():
    """The job's partition (video_sample.py:570-590 + one process per GPU): for any dataset size, batch size, world size and selection
    option, the batches dealt to the ranks r, r + R, .. cover every selected dataset item exactly once, in dataset order within a batch."""
    from argparse import Namespace
    from video_diffusion_amd import dist as vdist
    from video_diffusion_amd import video_sample as vs
    rng = np.random.RandomState(0)
    for _ in range(200):
        n, bs, world = int(rng.randint(1, 40)), int(rng.randint(1, 9)), int(rng.randint(1, 9))
        kind = rng.randint(4)
        a = Namespace(batch_size=bs, indices=None, task_id=None, subset_size=None)
        if kind == 1:
            a.indices = sorted(rng.choice(n, size=rng.randint(1, n + 1), replace=False).tolist())
        elif kind == 2:
            a.task_id = int(rng.randint(0, max(1, n // bs)))
        elif kind == 3:
            a.subset_size = int(rng.randint(1, n + 1))
        sel = vs.resolve_indices(a, n)
        batches = [sel[k:k + bs] for k in range(0, len(sel), bs)]
        seen = []
        for r in range(world):
            for t in vdist.task_ids(len(batches), r, world):
                seen += batches[t]
        assert sorted(seen) == sorted(sel) and len(set(seen)) == len(seen)
        want = a.indices if kind == 1 else list(range(a.task_id * bs, (a.task_id + 1) * bs)) if kind == 2 else list(range(a.subset_size)) if kind == 3 else list(range(n))
        assert sel == want


def test_windowed_sampler_host_loop_and_all_timestep_record_against_a_plain_restatement():
    """video_sample.infer_video's host logic on CPU with a stand-in step: window assembly, the write-back of the last n_latent slots, the
    per-item branch of the adaptive modes and the (B, num_timesteps, T, ...) record of --save_all_timesteps (scripts/video_sample.py:84-186)."""
    import torch
    from video_diffusion_amd import inference_util as iu
    from video_diffusion_amd.video_sample import infer_video

    class Model:
        device = torch.device("cpu")

        def check_device_errors(self):
            pass

    class Diff:
        num_timesteps = 3

        def p_sample(self, model, x, t, clip_denoised=True, model_kwargs=None, **kw):
            assert model_kwargs["x_t_minus_1"] is model_kwargs["x0"] and kw["return_attn_weights"] is False
            return {"sample": x * 0.5 + model_kwargs["latent_mask"] * (float(t[0]) + 0.1 * model_kwargs["frame_indices"][:, :, None, None, None].float())}

    B, T, obs, mf, step = 2, 10, 2, 5, 3
    batch = torch.rand(B, T, 3, 4, 4, generator=torch.Generator().manual_seed(2)) * 2 - 1
    for mode in ("autoreg", "adaptive-autoreg", "hierarchy-2", "independent"):
        got, every = infer_video(mode, Model(), Diff(), batch, mf, obs, step, adaptive_distance="l2", save_all_timesteps=True)
        samples = torch.zeros_like(batch)
        samples[:, :obs] = batch[:, :obs]
        rec = torch.zeros(B, 3, T, 3, 4, 4)
        rec[:, :, :obs] = samples[:, :obs].unsqueeze(1)
        adaptive = "adaptive" in mode
        it = iter(iu.inference_strategies[mode](video_length=T, num_obs=obs, max_frames=mf, step_size=step, optimal_schedule_path=None,
                                                **(dict(distance="l2") if adaptive else {})))
        while True:
            if adaptive:
                it.set_videos(samples)
            try:
                o, l = next(it)
            except StopIteration:
                break
            for i in range(B):
                oi, li = (o[i], l[i]) if adaptive else (o, l)
                x = samples[i, list(oi) + list(li)].clone()
                for r, ts in enumerate((2, 1, 0)):
                    x = x * 0.5
                    x[len(oi):] += ts + 0.1 * torch.tensor(li).float()[:, None, None, None]
                    rec[i, r, li] = x[len(oi):]
                samples[i, li] = x[len(oi):]
        assert np.array_equal(got, samples.numpy()), mode
        assert np.array_equal(every, rec.numpy()), mode
        plain, placeholder = infer_video(mode, Model(), Diff(), batch, mf, obs, step, adaptive_distance="l2")
        assert np.array_equal(plain, got) and placeholder.shape == (1,)
