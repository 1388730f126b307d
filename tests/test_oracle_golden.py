"""CPU: pin the oracle (oracle/*.py) against golden vectors minted from the imported reference.

tools/gen_golden.py wrote tests/golden/ by running /root/reference on CPU; these
tests never read the reference.  An oracle that fails here may not be used as a checker.
"""
import json

import numpy as np
import pytest
import torch

from helpers import case_inputs, close, load_json, load_npz, n_cases, synth_sd
from oracle.sampler_ref import SamplerRef
from oracle.schedule_ref import ScheduleRef, space_steps
from oracle.unet_ref import UNetRef, sinus_embedding


def test_space_timesteps_exact():
    for c in load_json("space_timesteps.json"):
        if "error" in c:
            with pytest.raises(ValueError):
                space_steps(c["n"], c["spec"])
        else:
            assert sorted(space_steps(c["n"], c["spec"])) == c["steps"], c["spec"]


@pytest.mark.parametrize("tag", ["linear1000_ddim250", "linear1000_full", "linear1000_ddim50",
                                 "cosine1000_ddim100", "linear1000_ddim5_small"])
def test_schedule_tables_bit_exact(tag):
    rec = load_json(f"schedule_{tag}.json")
    s = ScheduleRef(**rec["kw"])
    assert s.timestep_map == rec["timestep_map"]
    assert s.num_timesteps == rec["num_timesteps"]
    for name in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                 "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                 "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                 "posterior_mean_coef2"]:
        want = np.array([float.fromhex(h) for h in rec[name]])
        assert np.array_equal(getattr(s, name), want), name      # float64, bit for bit


def test_known_answers_survey_appendix_c():
    s = ScheduleRef(timestep_respacing="ddim250")
    assert s.timestep_map[:3] == [0, 4, 8] and s.timestep_map[-1] == 996
    np.testing.assert_allclose(s.betas[:3], [1e-4, 5.99065564e-4, 9.17602993e-4], rtol=1e-8)
    np.testing.assert_allclose(s.alphas_cumprod[-1], 4.287736906800033e-05, rtol=1e-12)


def _oracle_for(rec):
    cfg = json.loads(str(rec["cfg_json"]))
    specs = load_json("param_specs.json")
    return cfg


def _build(cfg, spec_tag=None):
    from oracle.unet_ref import topology  # noqa: F401
    import video_diffusion_amd as vda
    specs = vda.param_specs(cfg)
    return UNetRef(cfg, synth_sd(specs))


@pytest.mark.parametrize("name", ["unet_tiny.npz", "unet_tiny_table.npz", "unet_tiny_frameenc.npz",
                                  "unet_tiny_noss.npz"])
def test_unet_eps_matches_reference(name):
    rec = load_npz(name)
    cfg = json.loads(str(rec["cfg_json"]))
    net = _build(cfg)
    sampler = SamplerRef(ScheduleRef(cfg["diffusion_steps"], cfg["noise_schedule"], cfg["timestep_respacing"],
                                     cfg["sigma_small"], cfg["rescale_timesteps"]), net)
    for ci in range(n_cases(rec)):
        c = case_inputs(rec, ci)
        kw = dict(x0=c["x0"], obs_mask=c["obs_mask"], latent_mask=c["latent_mask"],
                  kinda_marg_mask=c["kinda_marg_mask"], frame_indices=c["frame_indices"],
                  observed_frames=c["observed_frames"], x_t_minus_1=c["x0"])
        eps = sampler.eps(c["x"], c["t"], kw)
        close(eps, c["eps"], atol=2e-5, rtol=2e-5)


def test_block_activations_match_reference():
    rec = load_npz("blocks_tiny.npz")
    urec = load_npz("unet_tiny.npz")
    cfg = json.loads(str(urec["cfg_json"]))
    net = _build(cfg)
    c = case_inputs(urec, 0)
    sched = ScheduleRef(timestep_respacing=cfg["timestep_respacing"])
    tm = torch.tensor([float(sched.model_timestep(int(rec["t"])))] * c["x"].shape[0])
    # timestep embedding (nn.py:89-107 + unet.py:605-610): latent frames see tm, observed frames see 0
    B, T = c["x"].shape[:2]
    om = c["obs_mask"].view(B, T)
    tfr = (tm.view(B, 1) * (1 - om)).reshape(-1)
    emb = net.lin(torch.nn.functional.silu(net.lin(sinus_embedding(tfr, cfg["num_channels"]), "time_embed.0")),
                  "time_embed.2")
    close(emb, rec["emb"], atol=2e-5, rtol=2e-5)
    net.taps = {"attn_t": [], "out_blocks": []}
    net(c["x"], tm, x0=c["x0"], obs_mask=c["obs_mask"], latent_mask=c["latent_mask"],
        kinda_marg_mask=c["kinda_marg_mask"], frame_indices=c["frame_indices"])
    close(net.taps["attn_t"][0][:, ::7, ::4, :], rec["in3_tattn"], atol=2e-5, rtol=2e-5)
    close(net.taps["out_blocks"][0][:, ::4, ::3, ::3], rec["out0"], atol=5e-5, rtol=5e-5)
    close(net.taps["out_blocks"][-1][:, ::4, ::3, ::3], rec["out_last"], atol=5e-5, rtol=5e-5)


def test_psample_and_ddim_match_reference():
    rec = load_npz("psample_tiny.npz")
    cfg = json.loads(str(load_npz("unet_tiny.npz")["cfg_json"]))
    net = _build(cfg)
    s = SamplerRef(ScheduleRef(timestep_respacing=cfg["timestep_respacing"]), net)
    T = {k: torch.from_numpy(rec[k]) for k in ["x", "x0", "noise", "obs_mask", "latent_mask", "kinda_marg_mask",
                                                "frame_indices"]}
    kw = dict(x0=T["x0"], obs_mask=T["obs_mask"], latent_mask=T["latent_mask"],
              kinda_marg_mask=T["kinda_marg_mask"], frame_indices=T["frame_indices"])
    B = T["x"].shape[0]
    for t_val in [249, 248, 1, 0]:
        t = torch.tensor([t_val] * B)
        o = s.p_sample(T["x"], t, kw, T["noise"])
        # x0_hat = sqrt(1/acp)*x - sqrt(1/acp-1)*eps amplifies an eps difference by up to 153x at t=249
        # (gaussian_diffusion.py:374-382): the bound on pred_xstart is the eps bound times that gain.
        gain = 1.0 + float(s.s.sqrt_recipm1_alphas_cumprod[t_val])
        close(o["pred_xstart"], rec[f"t{t_val}_pred_xstart"], atol=1e-5 * gain, rtol=5e-5)
        close(o["mean"], rec[f"t{t_val}_mean"], atol=5e-5, rtol=5e-5)
        close(o["sample"], rec[f"t{t_val}_psample"], atol=5e-5, rtol=5e-5)
        assert np.array_equal(o["log_variance"][:, 0, 0, 0, 0].numpy(), rec[f"t{t_val}_log_variance"])
        assert np.array_equal(o["variance"][:, 0, 0, 0, 0].numpy(), rec[f"t{t_val}_variance"])
        for eta in (0, 1):
            d = s.ddim_sample(T["x"], t, kw, T["noise"], eta=float(eta), eps=o["eps"])
            close(d["sample"], rec[f"t{t_val}_ddim_eta{eta}"], atol=1e-4, rtol=1e-4)
    q = s.q_sample(T["x0"], torch.tensor([3] * B), T["noise"])
    assert np.array_equal(q.numpy(), rec["q_sample_t3"])


def test_window_loop_matches_reference():
    rec = load_npz("window_tiny.npz")
    cfg = json.loads(str(rec["cfg_json"]))
    net = _build(cfg)
    s = SamplerRef(ScheduleRef(timestep_respacing=cfg["timestep_respacing"]), net)
    T = {k: torch.from_numpy(rec[k]) for k in ["x0", "obs_mask", "latent_mask", "kinda_marg_mask", "frame_indices"]}
    kw = dict(x0=T["x0"], obs_mask=T["obs_mask"], latent_mask=T["latent_mask"],
              kinda_marg_mask=T["kinda_marg_mask"], frame_indices=T["frame_indices"])
    noises = [torch.from_numpy(n) for n in rec["noises"]]
    out = s.window_loop(T["x0"], kw, noises)
    close(out, rec["final"], atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("tag,vertical,obs_frames", [("v2_xtm1", 2, "x_t_minus_1"), ("v0_x0", 0, "x_0"), ("v5_x0", 5, "x_0")])
def test_full_sampler_matches_reference(tag, vertical, obs_frames):
    """scripts/video_sample_full.py infer_video (vertical + horizontal loop nest) run by the reference itself
    (tools/gen_golden_full.py) against the oracle's restatement, same global-generator noise sequence."""
    from video_diffusion_amd import inference_util as iu
    rec = load_npz("full_sampler_tiny.npz")
    cfg = json.loads(str(rec["cfg_json"]))
    net = _build(cfg)
    s = SamplerRef(ScheduleRef(timestep_respacing=cfg["timestep_respacing"]), net)
    T, obs_len = int(rec["T"]), int(rec["obs_length"])
    gen = torch.Generator().manual_seed(int(rec["noise_seed"]))

    def sched():
        return iter(iu.inference_strategies["autoreg"](video_length=T, num_obs=obs_len, max_frames=int(rec["max_frames"]),
                                                       step_size=int(rec["step_size"])))
    out = s.full_loop(torch.from_numpy(rec["batch"]), sched, obs_len, vertical, obs_frames,
                      lambda shape: torch.randn(shape, generator=gen))
    ref = rec[f"samples_{tag}"]
    err = np.abs(out.numpy() - ref)
    assert err.mean() < 5e-5, err.mean()                  # 20 chained ddim5 steps: drift, not a per-step bound
    close(out, ref, atol=2e-2, rtol=1e-2)
    assert np.array_equal(out.numpy()[:, :obs_len], rec["batch"][:, :obs_len])


def _loops_setup(name):
    rec = load_npz(name)
    cfg = json.loads(str(rec["cfg_json"]))
    s = SamplerRef(ScheduleRef(timestep_respacing=cfg["timestep_respacing"]), _build(cfg))
    kw = {k: torch.from_numpy(rec[k]) for k in ["x0", "obs_mask", "latent_mask", "kinda_marg_mask", "frame_indices"]}
    return rec, s, kw


@pytest.mark.parametrize("obsf", ["x_0", "x_t_minus_1", "x_t"])
def test_p_sample_loop_matches_reference(obsf):
    """gaussian_diffusion.py:450-595 run by the reference from a seeded global generator (tools/gen_golden_loops.py):
    the restatement must consume the generator in the same order (initial image; per step x_t_minus_1's noise, random_t's
    uniform, x_random's noise, p_sample's noise) to land on the same trajectory."""
    rec, s, kw = _loops_setup("loops_tiny.npz")
    torch.manual_seed(int(rec[f"p_{obsf}_seed"]))
    outs = list(s.p_sample_loop_progressive(tuple(rec["x0"].shape), kw, obsf))
    assert len(outs) == 5
    close(outs[0][0]["sample"], rec[f"p_{obsf}_step0"], atol=5e-5, rtol=5e-5)
    last_kw = outs[-1][1]
    assert np.array_equal(last_kw["random_t"].numpy(), rec[f"p_{obsf}_random_t"])
    close(last_kw["x_t_minus_1"], rec[f"p_{obsf}_x_t_minus_1"], atol=1e-6, rtol=1e-6)
    final = outs[-1][0]["sample"].numpy()
    assert np.abs(final - rec[f"p_{obsf}_final"]).mean() < 5e-5           # 5 chained ddim5 steps: drift bound
    close(final, rec[f"p_{obsf}_final"], atol=5e-3, rtol=5e-3)


@pytest.mark.parametrize("eta", [0, 1])
def test_ddim_sample_loop_matches_reference(eta):
    """gaussian_diffusion.py:670-748, eta = 0 and 1."""
    rec, s, kw = _loops_setup("loops_tiny.npz")
    torch.manual_seed(int(rec[f"ddim_eta{eta}_seed"]))
    outs = list(s.ddim_sample_loop_progressive(tuple(rec["x0"].shape), kw, eta=float(eta)))
    close(outs[0]["sample"], rec[f"ddim_eta{eta}_step0"], atol=1e-4, rtol=1e-4)
    final = outs[-1]["sample"].numpy()
    assert np.abs(final - rec[f"ddim_eta{eta}_final"]).mean() < 5e-5
    close(final, rec[f"ddim_eta{eta}_final"], atol=5e-3, rtol=5e-3)


def test_nll_terms_match_reference():
    """p_mean_variance / _vb_terms_bpd / _prior_bpd / calc_bpd_loop_subsampled (gaussian_diffusion.py:229-372,750-790,
    909-1002; losses.py) against what the imported reference computed."""
    rec, s, kw = _loops_setup("nll_tiny.npz")
    x0, lm = kw["x0"], kw["latent_mask"]
    B = x0.shape[0]
    for tv in (4, 2, 0):
        t = torch.tensor([tv] * B)
        x_t = torch.from_numpy(rec[f"t{tv}_x_t"])
        mv = s.mean_variance(x_t, t, kw, clip=True)
        gain = 1.0 + float(s.s.sqrt_recipm1_alphas_cumprod[tv])
        close(mv["pred_xstart"], rec[f"t{tv}_pred_xstart"], atol=1e-5 * gain, rtol=5e-5)
        close(mv["mean"], rec[f"t{tv}_mean"], atol=5e-5, rtol=5e-5)
        assert np.array_equal(mv["variance"].numpy(), rec[f"t{tv}_variance"])
        assert np.array_equal(mv["log_variance"].numpy(), rec[f"t{tv}_log_variance"])
        for clip in (1, 0):
            vb = s.vb_terms_bpd(x0, x_t, t, kw, clip=bool(clip), latent_mask=lm)
            close(vb["output"], rec[f"t{tv}_vb_clip{clip}"], atol=1e-5, rtol=2e-4)
        close(s.vb_terms_bpd(x0, x_t, t, kw)["output"], rec[f"t{tv}_vb_nomask"], atol=1e-5, rtol=2e-4)
    close(s.prior_bpd(x0, lm), rec["prior_bpd"], atol=1e-7, rtol=1e-5)
    close(s.prior_bpd(x0), rec["prior_bpd_nomask"], atol=1e-7, rtol=1e-5)
    torch.manual_seed(int(rec["bpd_seed"]))
    m = s.calc_bpd_loop_subsampled(x0, kw, clip=True, latent_mask=lm)
    for k in ("total_bpd", "prior_bpd", "vb", "xstart_mse", "mse"):
        close(m[k], rec[f"bpd_{k}"], atol=1e-5, rtol=5e-4)
    torch.manual_seed(int(rec["bpd2_seed"]))
    m2 = s.calc_bpd_loop_subsampled(x0, kw, clip=True, latent_mask=lm, t_seq=rec["bpd2_t_seq"])
    for k in ("total_bpd", "vb", "mse"):
        close(m2[k], rec[f"bpd2_{k}"], atol=1e-5, rtol=5e-4)


def test_full_size_oracle_matches_reference_golden():
    """The oracle at FULL size (default 64x64 model, 116 M parameters, one 16-frame clip) against eps of the imported
    reference (tools/gen_golden_r3.py full).  The same script timed both on this container's 8 cores -- reference 1.86 s,
    oracle 1.89 s per step, max |d eps| 1.9e-6 (tests/golden/full_size_reference_vs_oracle.json) -- which is what backs
    bench.py's `cpu_baseline.kind: "port"`: the restatement is the same workload as the reference."""
    rec = load_npz("unet_full64.npz")
    cfg = json.loads(str(rec["cfg_json"]))
    import video_diffusion_amd as vda
    model, _ = vda.create_video_model_and_diffusion(**{k: cfg[k] for k in vda.video_model_and_diffusion_defaults()})
    net = UNetRef(cfg, synth_sd(model.param_specs()))
    sched = ScheduleRef(cfg["diffusion_steps"], cfg["noise_schedule"], cfg["timestep_respacing"], cfg["sigma_small"],
                        cfg["rescale_timesteps"])
    T, n_obs = int(rec["T"][0]), int(rec["n_obs"][0])
    g = torch.Generator().manual_seed(int(rec["seed"][0]))
    x0 = torch.rand(1, T, 3, 64, 64, generator=g) * 2 - 1
    x0[:, n_obs:] = 0
    x = torch.randn(1, T, 3, 64, 64, generator=g)
    obs = torch.zeros(1, T, 1, 1, 1)
    obs[:, :n_obs] = 1
    kw = dict(x0=x0, obs_mask=obs, latent_mask=1 - obs, kinda_marg_mask=torch.zeros(1, T, 1, 1, 1),
              frame_indices=torch.arange(T).view(1, T))
    got = SamplerRef(sched, net).eps(x, torch.tensor([int(rec["t"][0])]), kw)
    close(got, rec["eps"], atol=2e-5, rtol=1e-4)
    tm = load_json("full_size_reference_vs_oracle.json")["unet_full64.npz"]
    assert tm["max_abs_eps_diff"] < 2e-5 and 0.5 < tm["oracle_s_per_step"] / tm["reference_s_per_step"] < 2.0


@pytest.mark.parametrize("case", ["c32", "c64", "c64tab"])
def test_guidance_gradient_matches_reference(case):
    """use_gradient_method (gaussian_diffusion.py:264-271,350-364): x.grad, the shifted mean and p_sample's draw of the
    imported reference (tools/gen_golden_r3.py grad) against autograd through the oracle's own network."""
    rec = load_npz("grad_tiny.npz")
    cfg = json.loads(str(rec[f"{case}_cfg_json"]))
    import video_diffusion_amd as vda
    model, _ = vda.create_video_model_and_diffusion(**{k: cfg[k] for k in vda.video_model_and_diffusion_defaults()})
    net = UNetRef(cfg, synth_sd(model.param_specs()))
    sched = ScheduleRef(cfg["diffusion_steps"], cfg["noise_schedule"], cfg["timestep_respacing"], cfg["sigma_small"],
                        cfg["rescale_timesteps"])
    ora = SamplerRef(sched, net)
    g = lambda k: torch.from_numpy(rec[f"{case}_{k}"])  # noqa: E731
    kw = dict(x0=g("x0"), obs_mask=g("obs_mask"), latent_mask=g("latent_mask"), kinda_marg_mask=g("kinda_marg_mask"),
              frame_indices=g("frame_indices"), x_t_minus_1=g("x_t_minus_1"))
    B = g("x").shape[0]
    for t_val in [249, 100, 1, 0]:
        o = ora.guided_p_sample(g("x"), torch.tensor([t_val] * B), kw, g("noise"), g("noise2"))
        want = rec[f"{case}_t{t_val}_grad"]
        scale = float(np.abs(want).max())
        close(o["grad"], want, atol=2e-5 * scale, rtol=1e-3)
        close(o["mean"], rec[f"{case}_t{t_val}_mean"], atol=1e-4 * scale, rtol=1e-3)
        close(o["sample"], rec[f"{case}_t{t_val}_psample"], atol=1e-4 * scale, rtol=1e-3)


@pytest.mark.parametrize("name", ["dup", "allz", "t0", "ls"])
def test_cond_emb_variants_and_learn_sigma_forward_match_reference(name):
    """cond_emb_type duplicate / all-initzero / t=0 (unet.py:932-947,1014-1019 -- 't=0' with the reference's write through an
    expanded tensor: a whole batch item gets timestep -1 once one of its frames is observed) and the 6-channel network of
    learn_sigma=True at Boundary A (tools/gen_golden_r3.py variants)."""
    rec = load_npz("variants_tiny.npz")
    cfg = json.loads(str(rec[f"{name}_cfg_json"]))
    import video_diffusion_amd as vda
    model, _ = vda.create_video_model_and_diffusion(**{k: cfg[k] for k in vda.video_model_and_diffusion_defaults()})
    net = UNetRef(cfg, synth_sd(model.param_specs()))
    sched = ScheduleRef(cfg["diffusion_steps"], cfg["noise_schedule"], cfg["timestep_respacing"], cfg["sigma_small"],
                        cfg["rescale_timesteps"])
    ora = SamplerRef(sched, net)
    g = lambda k: torch.from_numpy(rec[f"{name}_{k}"])  # noqa: E731
    kw = dict(x0=g("x0"), obs_mask=g("obs_mask"), latent_mask=g("latent_mask"), kinda_marg_mask=g("kinda_marg_mask"),
              frame_indices=g("frame_indices"))
    for t_val in [100, 0]:
        t = torch.tensor([t_val] * 2)
        close(ora.eps(g("x"), t, kw), rec[f"{name}_t{t_val}_out"], atol=2e-5, rtol=1e-4)
        if name == "ls":
            assert str(rec[f"{name}_t{t_val}_psample_error"]) == "AssertionError"      # the reference cannot sample with it
            continue
        o = ora.p_sample(g("x"), t, kw, g("noise"))
        close(o["sample"], rec[f"{name}_t{t_val}_psample"], atol=2e-5, rtol=1e-4)
