#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/sec of the HIP engine on BASELINE.json configs[1].

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
                  [--image-size 64|128 --batch B --frames T --respacing ddim250 --executor eager|graph]

N > 1 with no WORLD_SIZE in the environment: this process starts N children (one per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set) BEFORE it touches a GPU and exits with the worst child's code; under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one of the ranks.

Workload (BASELINE.json configs[1]): default 64x64 video model (116 M parameters, closed-form synthetic
weights: no checkpoints exist), window (B=8, T=16 = 4 observed + 12 latent, 3x64x64), timestep_respacing='ddim250',
sampler = the ancestral `p_sample` that scripts/video_sample.py actually calls.  One "step" = one
`diffusion.p_sample` on that window = UNet forward + posterior update, inputs resident in HBM, noise drawn in-kernel
(Philox) so the timed region contains nothing but the step.  N GPUs: the test-set batch shard of
video_sample.py:577-582, no collective inside the step; `--scaling weak` (default) gives every rank its own B-clip
window, `--scaling strong` splits ONE B-clip window into B/N clips per rank.  The weights reach ranks > 0 through ONE
RCCL broadcast of the packed buffer.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel class: algorithmic FLOPs of its launches / their HIP-event durations, measured live in a
                separate profiled step on the launch stream; `traffic` comes from the committed rocprofv3 PMC passes
                (PMC cannot be read from inside the process) and says so in `traffic_source`
  dropin        the same K steps timed through `diffusion.p_sample(model, x, t, model_kwargs=...)`, the call
                scripts/video_sample.py:151 makes (per-step torch.randn_like, fresh output tensors, kwargs marshalling)
  cpu_baseline  the CPU oracle (torch fp32, the node's host cores) on the same window: 1 warm-up + median of 3 steps
  full_window   ONE whole window -- all 250 chained steps of the headline batch through video_sample.infer_video, the loop of
                scripts/video_sample.py:149-168 with its H2D / D2H and host loop -- timed once: seconds per clip batch under
                sustained clocks, beside the 20-step extrapolation `sec_per_clip_batch`
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2516.6    # v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16, 1024 SIMDs, 2.4 GHz (dense, no sparsity)
MFMA_BF16_SUSTAINED_RANDOM_TFLOPS = 1850.0   # measured: tools/mfma_rate.hip, random operands (constant operands: 2460)
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--batch", type=int, default=8, help="clips per window (per GPU when weak, in total when strong)")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--obs", type=int, default=4, help="observed frames of the window")
    ap.add_argument("--image-size", type=int, default=64)
    ap.add_argument("--respacing", default="ddim250")
    ap.add_argument("--num-res-blocks", type=int, default=2)
    ap.add_argument("--sampler", choices=["p_sample", "ddim"], default="p_sample",
                    help="ddim: ddim_sample with eta = 0 (SURVEY 8d config 2 asks for it beside the p_sample headline; the default run reports it as `ddim_sample_eta0`)")
    ap.add_argument("--executor", choices=["eager", "graph"], default="eager",
                    help="graph: the window executor (one captured hipGraph per window shape, device-resident step counter)")
    ap.add_argument("--prefix-cache", action="store_true",
                    help="with --executor graph: the observed frames' activations before the first attention layer once per "
                         "window (vd_set_window_prefix_cache); never the headline -- the default run reports it as an extra object")
    ap.add_argument("--suffix-skip", action="store_true",
                    help="with --executor graph: everything behind the last attention layer without the purely observed frames "
                         "(vd_set_window_suffix_skip); never the headline -- the default run reports it as an extra object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32-ref", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--parity-margin", action="store_true",
                    help="also report max |eps - reference eps| of the headline window against tests/golden/unet_full64_b8.npz (the default run "
                         "does, for its own arithmetic and, through its VD_MATH children, for the other two)")
    ap.add_argument("--no-full-window", action="store_true",
                    help="skip the whole-window leg (one 250-step window through video_sample.infer_video, ~6 s)")
    return ap.parse_args()


def rank_env(base, rank, world, port):
    """Environment of rank `rank` of a `world`-rank job on ONE node (what torch.distributed.run would set); pure, so a CPU test
    can hold it to the contract: RANK = LOCAL_RANK in [0, world), rendezvous on 127.0.0.1, dmabuf IPC for RCCL."""
    assert 0 <= rank < world and 0 < port < 65536
    return {**base, "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
            "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}


def self_launch(args):
    """`python bench.py --gpus N` by itself: one child per GPU, started before this process has made any GPU call
    (a process that has initialised the GPU must not start other programs on these boxes).  No exec: children are
    ordinary subprocesses, rank 0's stdout is ours, and we leave with the worst return code.  Ranks > 0 keep their stderr
    in gpurun_out/rank<r>.err (a rank that dies on the 8-GPU node must leave its reason somewhere)."""
    from video_diffusion_amd import _lib
    _lib.build()                                  # hipcc only; the ranks then find the library fresh
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    errdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out")
    os.makedirs(errdir, exist_ok=True)
    procs, errs = [], []
    for r in range(args.gpus):
        err = open(os.path.join(errdir, f"rank{r}.err"), "w") if r else None
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=rank_env(dict(os.environ), r, args.gpus, port),
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=err))
    worst = 0
    for r, p in enumerate(procs):
        rc = p.wait()
        if errs[r] is not None:
            errs[r].close()
        if rc != 0:
            worst = rc if worst == 0 or abs(rc) > abs(worst) else worst
            if r:
                try:
                    sys.stderr.write(f"[bench] rank {r} exited {rc}: " + open(os.path.join(errdir, f"rank{r}.err")).read()[-2000:] + "\n")
                except OSError:
                    pass
    return worst


def bench_cfg(vda, args):
    cfg = vda.video_model_and_diffusion_defaults()
    cfg.update(T=args.frames, image_size=args.image_size, num_res_blocks=args.num_res_blocks, rp_alpha=args.frames, rp_beta=args.frames,
               rp_gamma=args.frames, timestep_respacing=args.respacing)
    return cfg


def make_window(B, T, S, n_obs, seed, device):
    import torch
    g = torch.Generator().manual_seed(seed)
    video = torch.rand(B, T, 3, S, S, generator=g) * 2 - 1
    x0 = video.clone()
    x0[:, n_obs:] = 0                                   # latent slots start as zeros (video_sample.py:70-71)
    obs = torch.zeros(B, T, 1, 1, 1)
    obs[:, :n_obs] = 1
    kw = dict(frame_indices=torch.arange(T).view(1, T).repeat(B, 1), x0=x0, obs_mask=obs, latent_mask=1 - obs,
              kinda_marg_mask=torch.zeros(B, T, 1, 1, 1))
    return {k: v.to(device) for k, v in kw.items()}


class Stepper:
    """The hot loop of video_sample.py:150-168 on the C ABI with every per-step host allocation hoisted out."""

    def __init__(self, model, diff, kw, seed, sampler="p_sample", eta=0.0):
        import torch
        from video_diffusion_amd import _lib
        self._lib = _lib
        self.sampler, self.eta = sampler, float(eta)
        self.model, self.diff, self.seed = diff._bind(model), diff, seed
        x = kw["x0"].clone().float().contiguous()
        self.B, self.T = x.shape[:2]
        self.k = model._pack_kwargs(x, {**kw, "x_t_minus_1": kw["x0"], "observed_frames": "x_0"})
        self.bufs = [x, torch.empty_like(x)]
        self.per = x[0].numel()
        self.ts = [torch.full((self.B,), i, dtype=torch.int64, device=x.device) for i in range(diff.num_timesteps)]
        self.stream = _lib.current_stream()
        self.count = 0

    def step(self, t_index):
        _lib = self._lib
        src, dst = self.bufs[self.count & 1], self.bufs[(self.count + 1) & 1]
        k = self.k
        args = (self.model._handle, self.B, self.T, _lib.ptr(src), _lib.ptr(k["obs_src"]), _lib.ptr(k["obs_mask"]), _lib.ptr(k["latent_mask"]),
                _lib.ptr(k["kinda_marg_mask"]), _lib.ptr(k["frame_indices"]), _lib.ptr(self.ts[t_index]), k["obs_mode"], 1)
        tail = (None, self.seed, self.count * self.B * self.per, _lib.ptr(dst), None, None, self.stream)
        if self.sampler == "p_sample":
            rc = _lib.lib().vd_p_sample(*args, *tail)
        else:                                                # ddim_sample (gaussian_diffusion.py:597-634)
            rc = _lib.lib().vd_ddim_sample(*args, self.eta, *tail)
        _lib.check(rc)
        self.count += 1
        return dst

    def result(self):
        return self.bufs[self.count & 1]


class DropInStepper:
    """The reference's own call: `diffusion.p_sample(model, x, t, clip_denoised=True, model_kwargs=kw)['sample']`
    (scripts/video_sample.py:151-168), nothing hoisted."""

    def __init__(self, model, diff, kw):
        import torch
        self.model, self.diff = model, diff
        self.kw = {**kw, "x_t_minus_1": kw["x0"], "observed_frames": "x_0"}
        self.x = kw["x0"].clone()
        self.B = self.x.shape[0]
        self.torch = torch

    def step(self, t_index):
        t = self.torch.tensor([t_index] * self.B, device=self.x.device)
        self.x = self.diff.p_sample(self.model, self.x, t, clip_denoised=True, model_kwargs=self.kw)["sample"]
        return self.x

    def result(self):
        return self.x


class GraphStepper:
    """The window executor (vd_window_*): one hipGraph per window shape, step index and Philox counter on the device."""

    def __init__(self, model, diff, kw, seed, prefix_cache=False, suffix_skip=False):
        from video_diffusion_amd.executor import WindowExecutor
        self.ex = WindowExecutor(model, diff, prefix_cache=prefix_cache, suffix_skip=suffix_skip)
        self.kw = {**kw, "x_t_minus_1": kw["x0"], "observed_frames": "x_0"}
        self.x = kw["x0"].clone().float().contiguous()
        self.seed = seed
        self.started = False
        self.nts = diff.num_timesteps

    def step(self, t_index):
        if not self.started or t_index != self.next_t:
            self.ex.begin(self.x, self.kw, t_start=t_index, seed=self.seed, sampler="p_sample")
            self.started = True
        self.ex.run(1)
        self.next_t = t_index - 1
        return None

    def result(self):
        return self.ex.x


def timed(stepper, order, warmup, steps, vdist, device):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    import torch
    nts = len(order)
    for i in range(warmup):
        stepper.step(order[i % nts])
    torch.cuda.synchronize()
    vdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        stepper.step(order[(warmup + i) % nts])
    torch.cuda.synchronize()
    vdist.barrier()
    torch.cuda.synchronize()
    return vdist.max_over_ranks(time.perf_counter() - t0, device=device)


def profile_step(stepper, t_index):
    import ctypes
    import torch
    from video_diffusion_amd import _lib
    L = _lib.lib()
    n = L.vd_profile_classes()
    out = (ctypes.c_double * (4 * n))()
    torch.cuda.synchronize()
    _lib.check(L.vd_profile_begin())
    stepper.step(t_index)
    _lib.check(L.vd_profile_end(out, 4 * n))
    rows = {}
    for i in range(n):
        cnt, ms, fl, by = out[4 * i:4 * i + 4]
        if cnt:
            rows[L.vd_profile_class_name(i).decode()] = dict(launches=int(cnt), ms=ms, gflop=fl / 1e9, mb=by / 1e6)
    return rows


def parity_margin(vda, model, diff, cfg):
    """max |eps - eps_reference| of the HEADLINE window (B = 8 x T = 16 x 64 x 64, default 116 M model) against the fixture the imported
    reference produced for it (tests/golden/unet_full64_b8.npz, tools/gen_golden_r4.py: every 4th pixel of every frame), in the process'
    arithmetic -- the end-to-end accuracy beside the speed.  `tol_used` = max |d| / (1e-4 + 1e-4 |ref|): the share of the tier's tolerance."""
    import numpy as np
    import torch
    path = os.path.join(ROOT, "tests", "golden", "unet_full64_b8.npz")
    if not os.path.exists(path):
        return None
    rec = dict(np.load(path, allow_pickle=False))
    gcfg = json.loads(str(rec["cfg_json"]))
    if any(gcfg.get(k) != cfg.get(k) for k in gcfg if k in cfg):
        return {"skipped": "the bench configuration is not the fixture's"}
    B, T, n_obs, seed, S = int(rec["B"][0]), int(rec["T"][0]), int(rec["n_obs"][0]), int(rec["seed"][0]), gcfg["image_size"]
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(B, T, 3, S, S, generator=g) * 2 - 1
    x0[:, n_obs:] = 0
    obs = torch.zeros(B, T, 1, 1, 1)
    obs[:, :n_obs] = 1
    x = torch.randn(B, T, 3, S, S, generator=torch.Generator().manual_seed(seed + 1))
    dev = model.device
    kw = dict(frame_indices=torch.arange(T).view(1, T).repeat(B, 1).to(dev), x0=x0.to(dev), obs_mask=obs.to(dev), latent_mask=(1 - obs).to(dev),
              kinda_marg_mask=torch.zeros(B, T, 1, 1, 1).to(dev), x_t_minus_1=x0.to(dev), observed_frames="x_0")
    t = torch.tensor([int(rec["t"][0])] * B, device=dev)
    got, _ = diff._wrap_model(model)(x.to(dev), t, **kw)
    got = got.cpu().double().numpy()[:, :, :, ::4, ::4]
    ref = rec["eps_sub"].astype(np.float64)
    d = np.abs(got - ref)
    return {"max_abs_deps": float(d.max()), "mean_abs_deps": float(d.mean()), "tol_used": float((d / (1e-4 + 1e-4 * np.abs(ref))).max()),
            "eps_absmax": float(rec["eps_absmax"][0]), "values": int(d.size)}


def host_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (a GPU box advertises every core of the host but grants a share of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return min(n, int(os.environ.get("VD_CPU_BASELINE_CORES", "16")))


def cpu_baseline(cfg, sd, B, T, n_obs, t_index, nts):
    """The oracle (a torch-CPU fp32 restatement of the reference; kind 'port') on the node's host cores:
    1 warm-up + the median of VD_CPU_BASELINE_STEPS (3) timed steps, ~50 s of CPU work; a step that would pass
    VD_CPU_BASELINE_BUDGET_S is not started, and `sample` says how many were timed."""
    import torch
    from oracle.sampler_ref import SamplerRef
    from oracle.schedule_ref import ScheduleRef
    from oracle.unet_ref import UNetRef
    cores = host_cores()
    torch.set_num_threads(cores)
    ora = SamplerRef(ScheduleRef(cfg["diffusion_steps"], cfg["noise_schedule"], cfg["timestep_respacing"],
                                 cfg["sigma_small"], cfg["rescale_timesteps"]), UNetRef(cfg, sd))
    kw = make_window(B, T, cfg["image_size"], n_obs, seed=1234, device="cpu")
    x = kw["x0"].clone()
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
    t = torch.tensor([t_index] * B)
    budget = float(os.environ.get("VD_CPU_BASELINE_BUDGET_S", "75"))
    t_all = time.perf_counter()
    times = []
    for i in range(1 + int(os.environ.get("VD_CPU_BASELINE_STEPS", "3"))):   # step 0 = warm-up (allocator, thread pool, oneDNN primitives)
        t0 = time.perf_counter()
        ora.p_sample(x, t, kw, noise)
        times.append(time.perf_counter() - t0)
        if i >= 1 and time.perf_counter() - t_all + times[-1] > budget:
            break
    timed_ = sorted(times[1:]) if len(times) > 1 else times
    dt = timed_[len(timed_) // 2]
    return dict(value=1.0 / dt, unit="denoise-steps/sec", cores=cores, kind="port",
                sample=f"p_sample steps of the same (B={B},T={T},{cfg['image_size']}x{cfg['image_size']}) window: 1 warm-up "
                       f"({times[0]:.1f} s) + median of {len(timed_)} timed ({', '.join(f'{v:.1f}' for v in times[1:])} s)",
                s_per_clip_batch=nts * dt)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import video_diffusion_amd as vda
    from video_diffusion_amd import dist as vdist

    # The matrix products run in the library's default arithmetic (VD_MATH=f16x3: fp32 operands carried as two fp16 pieces,
    # three piece products, fp32 accumulation -- as accurate against fp64 as the fp32 MFMA).  For a reader who wants
    # the number in the EXACT-split arithmetic (bf16x6, the default of earlier rounds) or with every matrix product on the fp32
    # MFMA, the same benchmark is run first in child processes with VD_MATH set -- started before this process touches the GPU.
    math = os.environ.get("VD_MATH") or "f16x3"
    fp32_ref = x6_ref = pc_ref = ss_ref = shapes_ref = None
    # (never under a profiler: its preloaded library has already initialised the GPU in this process, and starting
    # another program from such a process is not allowed on the GPU boxes)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(
        k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")) for k in os.environ)
    SIDE_OFF = ["--no-cpu-baseline", "--no-roofline", "--no-fp32-ref", "--no-dropin", "--no-full-window"]

    def child(extra, env=None):
        """The same benchmark in a child process (started before this process touches the GPU); its JSON line, or {'error': ...}."""
        c = subprocess.run([sys.executable, os.path.abspath(__file__), *sys.argv[1:], *SIDE_OFF, *extra], env={**os.environ, **(env or {})},
                           capture_output=True, text=True)
        try:
            return json.loads([l for l in c.stdout.splitlines() if l.startswith("{")][-1])
        except Exception:                                            # noqa: BLE001 - the headline run must not depend on a side leg
            return {"error": (c.stderr or c.stdout)[-300:]}

    def pick(ref, note, mode=None, **more):
        if "error" in ref:
            return ref
        pm = (ref.get("parity_margin") or {}).get(mode)
        return {"value": ref["value"], "ms_per_step": ref["ms_per_step"], **({"parity_margin": pm} if pm else {}), "note": note, **more}

    if args.gpus == 1 and not args.no_fp32_ref and not profiled and math == "f16x3":
        fp32_ref = pick(child(["--parity-margin"], {"VD_MATH": "fp32"}), "VD_MATH=fp32: every matrix product on v_mfma_f32_32x32x2_f32", mode="fp32")
        x6_ref = pick(child(["--parity-margin"], {"VD_MATH": "bf16x6"}),
                      "VD_MATH=bf16x6: fp32 operands split EXACTLY into three bf16 pieces, six piece products (the default arithmetic of rounds "
                      "1-3; same kernels, twice the MFMAs)", mode="bf16x6")
        if args.executor == "eager":
            # the window executor's opt-in modes: work that a window really does only once (prefix cache), or that its caller never reads
            # (suffix skip), is outside `value` -- which stays the step that recomputes every frame -- and is reported beside it
            ref = child(["--executor", "graph", "--prefix-cache"])
            pc_ref = pick(ref, "window executor with vd_set_window_prefix_cache: the observed frames' activations before the first attention "
                               "layer are computed once per window (250 steps), the timed steps run those blocks on the other frames only; same "
                               "arithmetic per frame (tests/test_gpu_engine.py: within 2e-6 of the uncached window); opt-in, NOT the headline",
                          **({} if "error" in ref else {"cached_frames_per_window": ref["config"].get("cached_frames")}))
            try:
                ref = child(["--executor", "graph", "--suffix-skip"])
                both = child(["--executor", "graph", "--suffix-skip", "--prefix-cache"])
                ss_ref = {"value": ref["value"], "ms_per_step": ref["ms_per_step"], "suffix_frames_per_window": ref["config"].get("suffix_frames"),
                          "with_prefix_cache": {"value": both.get("value"), "ms_per_step": both.get("ms_per_step"),
                                                "cached_frames_per_window": both.get("config", {}).get("cached_frames")},
                          "note": "window executor with vd_set_window_suffix_skip: behind the last attention layer (decoder blocks at 32x32 / "
                                  "64x64, both Upsample convs, the head) the captured step runs on the frames that are not pure observations; "
                                  "those frames' samples equal the full step's to the bit (tests/test_gpu_engine.py), the observed frames' "
                                  "entries of the window are meaningless and never read (scripts/video_sample.py:170-186); opt-in, NOT the headline; "
                                  "with_prefix_cache: both opt-ins together"}
                # the window shapes of BASELINE configs[2] (MineRL 64x64: 20 frames, 13 of them observed), configs[3] (UCF101 128x128: B = 4 x
                # T = 16, the un-respaced 1000-step schedule) and configs[4] (CARLA 128x128: 20 frames, 10 observed, DDIM-50): the eager step
                # (what `value` is at the headline shape), the graph executor, and the graph executor with the suffix skip
                shapes_ref = {}
                # windows per video: autoreg walks (T - obs_length) latent frames step_size at a time -- configs[2] (300 - 36) / 7 -> 38 windows,
                # configs[4] (500 - 36) / 10 -> 47; a whole video (batch of 8) is priced as that many windows of the benchmarked shape (the last
                # window of a video has fewer latent frames and is cheaper with the suffix skip, dearer without: not modelled)
                for name, shape, nsteps, graph, nwin in (
                        ("configs2_window_20f_13obs", ["--frames", "20", "--obs", "13", "--steps", "10", "--warmup", "3"], 250, True, 38),
                        ("configs3_window_128px_B4_16f_ddpm1000", ["--image-size", "128", "--batch", "4", "--respacing", "", "--steps", "8", "--warmup", "2"], 1000, False, 1),
                        ("configs4_window_128px_20f_10obs_ddim50", ["--image-size", "128", "--frames", "20", "--obs", "10", "--respacing", "ddim50",
                                                                      "--steps", "5", "--warmup", "2"], 50, True, 47)):
                    e = child(shape)
                    shapes_ref[name] = {"eager_ms_per_step": e.get("ms_per_step"), "eager_steps_per_sec": e.get("value"), "steps_per_window": nsteps,
                                        "sec_per_window_eager": round(e["ms_per_step"] * nsteps / 1e3, 3) if "ms_per_step" in e else None,
                                        **({"error": e["error"]} if "error" in e else {})}
                    if graph:
                        a, b2 = child(["--executor", "graph", *shape]), child(["--executor", "graph", "--suffix-skip", *shape])
                        ss_ref[name] = {"ms_per_step": a["ms_per_step"], "ms_per_step_suffix_skip": b2["ms_per_step"],
                                        "sec_per_window": round(a["ms_per_step"] * nsteps / 1e3, 3),
                                        "sec_per_window_suffix_skip": round(b2["ms_per_step"] * nsteps / 1e3, 3),
                                        "suffix_frames_per_window": b2["config"].get("suffix_frames"), "steps_per_window": nsteps}
                        # what a user of infer_video(executor='graph', suffix_skip=True) sees: seconds for one batch of whole videos (every frame the
                        # caller reads is bit-identical to the default path, tests/test_gpu_engine.py)
                        if "ms_per_step" in e:
                            shapes_ref[name].update(windows_per_video=nwin,
                                                    sec_per_video_batch_eager=round(nwin * e["ms_per_step"] * nsteps / 1e3, 1),
                                                    sec_per_video_batch_graph=round(nwin * a["ms_per_step"] * nsteps / 1e3, 1),
                                                    sec_per_video_batch_graph_suffix_skip=round(nwin * b2["ms_per_step"] * nsteps / 1e3, 1),
                                                    speedup_of_the_opt_in=round(e["ms_per_step"] / b2["ms_per_step"], 3))
            except Exception as e:                                   # noqa: BLE001
                ss_ref = {**(ss_ref or {}), "error": repr(e)[-300:]}

    # rehearsal knobs for a one-GPU box (the N > 1 path is the driver's to run on an 8-GPU node): VD_BENCH_BACKEND=gloo
    # and VD_BENCH_ALL_ON_DEVICE0=1 put every rank on device 0 (RCCL refuses two ranks on one GPU, gloo does not)
    assert torch.cuda.is_available(), "bench.py measures the HIP engine: it needs a GPU (no CPU fallback)"
    # the rank's GPU is chosen before the process group exists and handed to it (dist.init: set_device, then init_process_group(device_id=...))
    rank, local_rank, world = vdist.init(backend=os.environ.get("VD_BENCH_BACKEND"), device_index=0 if os.environ.get("VD_BENCH_ALL_ON_DEVICE0") else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("VD_BENCH_ALL_ON_DEVICE0"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    cfg = bench_cfg(vda, args)
    S, T, n_obs = args.image_size, args.frames, args.obs
    if args.scaling == "strong":
        assert args.batch % world == 0, "--scaling strong splits the B-clip window evenly: batch % gpus == 0"
        B = args.batch // world
    else:
        B = args.batch
    model, diff = vda.create_video_model_and_diffusion(**cfg)
    model.to(device).eval()
    specs = model.param_specs()
    sd_holder = {}

    def make_sd():
        sd_holder["sd"] = {k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in specs}
        return sd_holder["sd"]

    vdist.share_weights(model, make_sd, rank)           # one RCCL broadcast of the packed buffer
    import torch.distributed as tdist
    rccl_ranks = tdist.get_world_size() if tdist.is_initialized() else 1   # what the group says after the broadcast, not what the flags asked for
    assert rccl_ranks == args.gpus, f"process group has {rccl_ranks} ranks, --gpus {args.gpus}"

    kw = make_window(B, T, S, n_obs, seed=1234 + rank, device=device)
    assert not args.prefix_cache or args.executor == "graph", "--prefix-cache is a mode of the window executor (--executor graph)"
    assert not args.suffix_skip or args.executor == "graph", "--suffix-skip is a mode of the window executor (--executor graph)"
    stepper = GraphStepper(model, diff, kw, seed=5 + rank, prefix_cache=args.prefix_cache, suffix_skip=args.suffix_skip) if args.executor == "graph" else \
        Stepper(model, diff, kw, seed=5 + rank, sampler=args.sampler)
    assert args.sampler == "p_sample" or args.executor == "eager", "--sampler ddim: eager executor"
    nts = diff.num_timesteps
    order = list(range(nts))[::-1]

    elapsed = timed(stepper, order, args.warmup, args.steps, vdist, device)
    assert torch.isfinite(stepper.result()).all()
    model.check_device_errors()                                    # a non-finite network output inside the timed steps (vd_device_errors bit 1) fails the run

    # SURVEY 8d config 2 asks for the p_sample number AND, separately, ddim_sample with eta = 0 (gaussian_diffusion.py:597-634): the same
    # window, the same network forward, the deterministic update -- timed here in the same process (never `value`)
    ddim_leg = None
    if world == 1 and args.executor == "eager" and args.sampler == "p_sample" and not args.no_dropin:
        el = timed(Stepper(model, diff, kw, seed=5 + rank, sampler="ddim", eta=0.0), order, args.warmup, args.steps, vdist, device)
        ddim_leg = {"value": round(args.steps / el, 4), "ms_per_step": round(1e3 * el / args.steps, 3),
                    "what": "vd_ddim_sample (eta = 0) on the same window: UNet forward + the DDIM update of gaussian_diffusion.py:597-634"}
    margin = None
    if rank == 0 and (args.parity_margin or (world == 1 and not args.no_roofline and args.executor == "eager")):
        margin = parity_margin(vda, model, diff, cfg)

    full_window = None
    if world == 1 and not args.no_full_window and args.executor == "eager" and rank == 0:
        # the loop being replaced, whole: scripts/video_sample.py:149-168 (250 chained p_sample calls on one window of the
        # test-set batch + the window's H2D / D2H), through the drop-in caller video_sample.infer_video
        from video_diffusion_amd import video_sample
        gb = torch.Generator().manual_seed(99)
        batch = torch.rand(B, T, 3, S, S, generator=gb) * 2 - 1
        torch.manual_seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, _ = video_sample.infer_video("independent", model, diff, batch, max_frames=T, obs_length=n_obs, step_size=max(T - n_obs, 1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        import numpy as np
        assert np.isfinite(out).all() and out.shape == tuple(batch.shape)
        full_window = {"sec_per_clip_batch": round(dt, 3), "steps": nts, "steps_per_sec": round(nts / dt, 3),
                       "what": f"video_sample.infer_video('independent', ...): one window of {B} clips x {T} frames ({n_obs} observed), {nts} chained "
                               "p_sample steps from the host loop (executor='eager'), torch.randn_like noise, window H2D + D2H included"}

    dropin = None
    if not args.no_dropin:
        elapsed_d = timed(DropInStepper(model, diff, kw), order, args.warmup, args.steps, vdist, device)
        dropin = elapsed_d

    roofline, classes = None, None
    if not args.no_roofline:
        prof_stepper = stepper if args.executor == "eager" else Stepper(model, diff, kw, seed=5 + rank)
        classes = profile_step(prof_stepper, order[0])
        if rank == 0:
            name = max((k for k in classes if k.startswith(("gemm_", "igemm", "conv3x3"))), key=lambda k: classes[k]["ms"])
            c = classes[name]
            achieved = c["gflop"] / c["ms"]                                   # GFLOP/ms = TFLOP/s
            # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
            # runs, gfx950 x2 correction on FETCH_SIZE; tools/profile_bench.sh): PMC cannot be read inside this process
            traffic, traffic_source = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_dominant.json")
            headline = (S, args.batch, T, args.num_res_blocks) == (64, 8, 16, 2) and B == 8
            pmc_kernels = {}
            if os.path.exists(pmc) and headline:
                # the counters belong to the kernel source they were collected on: a kernel edited since then carries no
                # traffic figure until tools/profile_bench.sh + tools/make_pmc_dominant.py have been re-run
                import hashlib

                def fresh(r):
                    srcs = r.get("kernel_sources", {})
                    return bool(srcs) and all(os.path.exists(os.path.join(ROOT, f)) and hashlib.sha1(open(os.path.join(ROOT, f), "rb").read()).hexdigest() == h
                                              for f, h in srcs.items())
                pmc_kernels = {k: (r if fresh(r) else None) for k, r in json.load(open(pmc)).get("kernels", {}).items()}
                src = "profiles/pmc_dominant.json (rocprofv3 --pmc passes of this workload on this kernel source, not this run)"
                if pmc_kernels.get(name):
                    rec = pmc_kernels[name]
                    pmc_proof = {k: rec[k] for k in ("mfma_insts_per_launch", "mfma_insts_expected_per_launch", "mfma_insts_ratio") if k in rec}
                    traffic = round(rec["hbm_bytes_per_launch"])
                    traffic_source = src
                elif name in pmc_kernels:
                    traffic_source = "profiles/pmc_dominant.json is stale (kernel source changed since the PMC passes): traffic withheld"
            split_conv = name in ("conv3x3_wino_r64_kernel", "conv3x3_wino_z128_kernel")
            # multiplications a direct 3x3 conv spends per one the kernel executes: F(2x2,3x3) = 36 / 16; conv_wino_z128.hip folds the
            # column half of the output transform into the accumulation (six MFMA groups per four positions): 36 / 24
            wino_gain = 1.5 if name == "conv3x3_wino_z128_kernel" else 2.25
            pieces = 3 if math == "f16x3" else 6                              # piece products per element product
            # the dominant kernel runs on the 16-bit matrix pipe (fp32 operands carried as fp16 / bf16 pieces) unless
            # VD_MATH=fp32 keeps it on the fp32 MFMA: `peak` is the dense peak of the pipe it uses (f16 = bf16 rate)
            peak = PEAK_BF16_MFMA_TFLOPS if split_conv or name.startswith("gemm_split") else PEAK_FP32_MFMA_TFLOPS
            roofline = dict(bound="mfma", kernel=name, achieved=round(achieved, 2), peak=peak, **(locals().get("pmc_proof") or {}),
                            unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_source,
                            launches_per_step=c["launches"], avg_launch_us=round(1e3 * c["ms"] / c["launches"], 1),
                            alg_gflop_per_launch=round(c["gflop"] / c["launches"], 3),
                            alg_mb_per_launch=round(c["mb"] / c["launches"], 2),
                            frac_note="achieved = ALGORITHMIC direct-form fp32 FLOPs / time; the share of the matrix pipe "
                                      "kept busy is mfma_executed_frac")
            if name == "conv3x3_wino_z128_kernel" and math == "f16x3":
                roofline["also_does"] = ("since round 5 this kernel applies the GroupNorm(+FiLM) affine + SiLU of its input in its patch staging (16 of its 16 launches "
                                         "at the headline): its launches take ~7 % longer for the same algorithmic FLOPs, and 13 activation passes + 3 skip-conv "
                                         "images of the step are gone (DESIGN.md 4; kernel_classes: affine_act_kernel)")
            if name.startswith("conv3x3_wino"):
                # `achieved` counts the ALGORITHMIC flops of a direct fp32 3x3 convolution (2*M*Cout*Cin*9).  Winograd
                # F(2x2,3x3) executes 16/36 of the multiplications (24/36 with the folded column transform); the split kernel spends
                # `pieces` piece products on each, so the matrix pipe executes achieved * pieces / wino_gain flops (fp32 kernel: achieved / 2.25)
                ex = achieved * (pieces / wino_gain if split_conv else 1 / 2.25)
                roofline["direct_over_executed_multiplications"] = wino_gain
                roofline["piece_products"] = pieces if split_conv else 1
                roofline["mfma_executed_tflops"] = round(ex, 2)
                roofline["mfma_executed_frac"] = round(ex / peak, 4)
                if split_conv:
                    # the nominal peak assumes 2.4 GHz; a loop of nothing but this MFMA on random operands sustains 1.83-1.87
                    # PFLOP/s on this board (power; tools/mfma_rate.hip) -- reported beside the nominal fraction
                    roofline["mfma_sustained_random_operands_tflops"] = MFMA_BF16_SUSTAINED_RANDOM_TFLOPS
                    roofline["mfma_executed_over_sustained"] = round(ex / MFMA_BF16_SUSTAINED_RANDOM_TFLOPS, 4)
                roofline["fp32_mfma_peak"] = PEAK_FP32_MFMA_TFLOPS          # what a plain fp32 kernel is bounded by
                roofline["achieved_over_fp32_mfma_peak"] = round(achieved / PEAK_FP32_MFMA_TFLOPS, 4)
            # the next two classes by time, same figures (verdict r4 #6: `traffic` for ~75 % of the step, not for one kernel)
            others = []
            for k2 in sorted((k for k in classes if k != name and k.startswith(("gemm_", "igemm", "conv3x3"))), key=lambda k: -classes[k]["ms"])[:3]:
                c2 = classes[k2]
                pk = PEAK_BF16_MFMA_TFLOPS if (k2.startswith(("conv3x3_wino_r64", "conv3x3_wino_z128", "gemm_split")) and math != "fp32") else PEAK_FP32_MFMA_TFLOPS
                r2 = pmc_kernels.get(k2)
                others.append(dict(kernel=k2, bound="mfma", achieved=round(c2["gflop"] / c2["ms"], 2), peak=pk, unit="TFLOP/s",
                                   frac=round(c2["gflop"] / c2["ms"] / pk, 4), launches_per_step=c2["launches"], ms_per_step=round(c2["ms"], 3),
                                   alg_mb_per_launch=round(c2["mb"] / c2["launches"], 2),
                                   traffic=round(r2["hbm_bytes_per_launch"]) if r2 else None,
                                   traffic_over_algorithmic=round(r2["hbm_bytes_per_launch"] / (c2["mb"] / c2["launches"] * 1e6), 3) if r2 and c2["mb"] else None,
                                   mfma_busy_frac_pmc=round(r2["mfma_busy_frac"], 4) if r2 and r2.get("mfma_busy_frac") else None,
                                   mfma_insts_ratio=r2.get("mfma_insts_ratio") if r2 else None))
            roofline["other_kernels"] = others
            if pmc_kernels.get(name) and pmc_kernels[name].get("mfma_busy_frac"):
                roofline["mfma_busy_frac_pmc"] = round(pmc_kernels[name]["mfma_busy_frac"], 4)
            if traffic and c["mb"]:
                roofline["traffic_over_algorithmic"] = round(traffic / (c["mb"] / c["launches"] * 1e6), 3)
    vdist.barrier()
    # the last collective of the run: every rank leaves the job together (a rank that exits with its RCCL communicator alive can stall the others' teardown)
    if tdist.is_initialized():
        tdist.destroy_process_group()

    if rank != 0:
        return
    # weak: every rank denoises its own window -> windows/s add up; strong: the ranks share ONE window
    windows_per_step = world if args.scaling == "weak" else 1
    value = windows_per_step * args.steps / elapsed
    headline = (S, args.batch, T, args.respacing, args.num_res_blocks, n_obs) == (64, 8, 16, "ddim250", 2, 4)
    nparam = sum(int(torch.tensor(s).prod()) for _, s in specs)
    workload = ("BASELINE configs[1]: BAIR-shaped 64x64, T=16 (4 obs + 12 latent), batch 8 per GPU, ddim250 respacing, "
                "p_sample, independent mode, default 116M-param video UNet") if headline and args.scaling == "weak" and args.sampler == "p_sample" else \
        (f"{S}x{S}, T={T} ({n_obs} obs + {T - n_obs} latent), batch {args.batch} {'per GPU' if args.scaling == 'weak' else 'in total'}, "
         f"{args.respacing or 'no'} respacing, {args.sampler}{' (eta 0)' if args.sampler == 'ddim' else ''}, default video UNet (num_res_blocks={args.num_res_blocks}, {nparam / 1e6:.0f}M params)")
    line = {
        "metric": "denoise-steps/sec", "value": round(value, 4), "unit": "denoise-steps/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "arithmetic": "VD_MATH=fp32: every matrix product on the fp32 MFMA" if math == "fp32" else
                      "VD_MATH=bf16x6: fp32 operands and fp32 accumulation; matrix products as six bf16 piece products of EXACTLY split "
                      "fp32 operands" if math == "bf16x6" else
                      "VD_MATH=f16x3 (default): fp32 tensors and fp32 accumulation throughout; in the matrix products (3x3 convs as "
                      "Winograd F(2x2,3x3), linear layers, 1x1 and stride-2 convs) every fp32 operand is carried as two fp16 pieces "
                      "(22 significand bits, exact power-of-two scaling) and a product is three piece products on the f16 MFMA; error "
                      "against fp64 held to the fp32-MFMA kernel's (max <= 1.5x, mean <= 1.25x asserted in tests/test_gpu_ops.py and, on hostile operands, "
                      "tests/test_gpu_hostile.py; measured <= 1.06x / 1.02x, profiles/r04_split_accuracy.json, r05_hostile_accuracy.json); usable operand range "
                      "|x| < 2^15, beyond it NaN + vd_device_errors bit 1 (FloatingPointError); the exact-split and fp32-MFMA numbers of the same run: "
                      "bf16x6_exact_split, fp32_mfma_only; end-to-end margins: parity_margin",
        "sec_per_clip_batch": round(nts * elapsed / args.steps, 2),
        "lib_source_sha": __import__("video_diffusion_amd")._lib.lib().vd_source_sha().decode(),
        "config": {"workload": workload, "batch_per_gpu": B, "frames": T, "image_size": S, "respaced_steps": nts,
                   "parallelism": f"batch-shard x{world} (no collective in the step)", "rccl_ranks": rccl_ranks,
                   "executor": args.executor + ("+prefix_cache" if args.prefix_cache else "") + ("+suffix_skip" if args.suffix_skip else ""),
                   **({"cached_frames": stepper.ex.cached_frames} if args.prefix_cache else {}),
                   **({"suffix_frames": stepper.ex.suffix_frames} if args.suffix_skip else {})},
        "roofline": roofline,
    }
    if dropin is not None:
        v = windows_per_step * args.steps / dropin
        line["dropin"] = {"value": round(v, 4), "ms_per_step": round(1e3 * dropin / args.steps, 3),
                          "ratio_to_value": round(v / value, 4),
                          "what": "the same steps through diffusion.p_sample(model, x, t, clip_denoised=True, model_kwargs=kw) "
                                  "(scripts/video_sample.py:151), torch.randn_like noise, fresh tensors per step"}
    if full_window is not None:
        line["full_window"] = full_window
    if ddim_leg is not None:
        line["ddim_sample_eta0"] = ddim_leg
    if margin is not None:
        # end-to-end accuracy of every arithmetic mode of this run beside its speed: max |eps - reference eps| on the headline window
        line["parity_margin"] = {"fixture": "tests/golden/unet_full64_b8.npz: eps of the imported reference for the headline window (every 4th pixel "
                                            "of all 128 frames); tolerance of the tier 1e-4 + 1e-4 |ref|", math: margin,
                                 **{k: r["parity_margin"] for k, r in (("fp32", fp32_ref), ("bf16x6", x6_ref)) if r and r.get("parity_margin")}}
    if shapes_ref is not None:
        line["window_shapes"] = shapes_ref
    if fp32_ref is not None:
        line["fp32_mfma_only"] = fp32_ref
    if x6_ref is not None:
        line["bf16x6_exact_split"] = x6_ref
    if pc_ref is not None:
        line["window_prefix_cache_opt_in"] = pc_ref
    if ss_ref is not None:
        line["window_suffix_skip_opt_in"] = ss_ref
    if classes is not None:
        line["kernel_classes"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                      "tflops": round(v["gflop"] / v["ms"], 2) if v["gflop"] else None,
                                      "gbs": round(v["mb"] / v["ms"], 1)} for k, v in classes.items()}
    if world == 1 and not args.no_cpu_baseline:
        sd = sd_holder.get("sd") or make_sd()
        line["cpu_baseline"] = cpu_baseline(cfg, sd, B, T, n_obs, order[0], nts)
        line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
