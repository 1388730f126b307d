#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/sec of the HIP engine on BASELINE.json config 2.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload (BASELINE.json configs[1]): default 64x64 video model (116 M parameters,
closed-form synthetic weights: no checkpoints exist), window (B=8, T=16 = 4 observed + 12 latent,
3x64x64), timestep_respacing='ddim250', sampler = the ancestral `p_sample` that
scripts/video_sample.py actually calls.  One "step" = one `diffusion.p_sample` on that
window = UNet forward + posterior update, inputs resident in HBM, noise drawn in-kernel (Philox) so
the timed region contains nothing but the step.  N GPUs: every rank runs its own B=8 window (the
test-set batch shard of video_sample.py:577-582; no collective inside the step) -> weak scaling;
the weights reach ranks > 0 through ONE RCCL broadcast of the packed buffer.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel class (implicit-GEMM conv on fp32 MFMA): algorithmic FLOPs of its
                launches / their HIP-event durations measured in a separate profiled step
  cpu_baseline  the CPU oracle (torch fp32, all host cores) timed on the same window, 1 step
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import video_diffusion_amd as vda  # noqa: E402
from video_diffusion_amd import _lib, dist as vdist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2516.6    # v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16, 1024 SIMDs, 2.4 GHz (dense, no sparsity)
PEAK_HBM_GBS = 8000.0


def headline_cfg():
    cfg = vda.video_model_and_diffusion_defaults()
    cfg.update(T=16, image_size=64, rp_alpha=16, rp_beta=16, rp_gamma=16, timestep_respacing="ddim250")
    return cfg


def make_window(B, T, S, n_obs, seed, device):
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
    """The hot loop of video_sample.py:150-168 with every per-step host allocation hoisted out."""

    def __init__(self, model, diff, kw, seed):
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
        src, dst = self.bufs[self.count & 1], self.bufs[(self.count + 1) & 1]
        k = self.k
        rc = _lib.lib().vd_p_sample(self.model._handle, self.B, self.T, _lib.ptr(src), _lib.ptr(k["obs_src"]),
                                    _lib.ptr(k["obs_mask"]), _lib.ptr(k["latent_mask"]), _lib.ptr(k["kinda_marg_mask"]),
                                    _lib.ptr(k["frame_indices"]), _lib.ptr(self.ts[t_index]), k["obs_mode"], 1, None,
                                    self.seed, self.count * self.B * self.per, _lib.ptr(dst), None, None, self.stream)
        _lib.check(rc)
        self.count += 1
        return dst


def profile_step(stepper, t_index):
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


def cpu_baseline(cfg, sd, B, T, n_obs, t_index):
    """The oracle (a torch-CPU fp32 restatement of the reference; kind 'port') on the node's host cores."""
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
    t0 = time.perf_counter()
    ora.p_sample(x, t, kw, noise)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="denoise-steps/sec", cores=cores, kind="port",
                sample=f"1 p_sample step of the same (B={B},T={T},64x64) window, no warm-up, {dt:.1f} s",
                s_per_clip_batch=250 * dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32-ref", action="store_true")
    args = ap.parse_args()

    # The linear layers run as six bf16 piece products of exactly split fp32 operands (csrc/gemm_split.hip; as accurate
    # as the fp32 MFMA).  For a reader who wants the number with EVERY matrix product on the fp32 MFMA, the
    # same benchmark is run first in a child process with VD_MATH=fp32 -- started before this process touches the GPU.
    fp32_ref = None
    # (never under a profiler: its preloaded library has already initialised the GPU in this process, and starting
    # another program from such a process is not allowed on the GPU boxes)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(
        k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")) for k in os.environ)
    if args.gpus == 1 and not args.no_fp32_ref and not profiled and os.environ.get("VD_MATH") != "fp32":
        import subprocess
        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup",
                                str(args.warmup), "--batch", str(args.batch), "--frames", str(args.frames), "--no-cpu-baseline",
                                "--no-roofline", "--no-fp32-ref"], env={**os.environ, "VD_MATH": "fp32"}, capture_output=True, text=True)
        try:
            ref = json.loads([l for l in child.stdout.splitlines() if l.startswith("{")][-1])
            fp32_ref = {"value": ref["value"], "ms_per_step": ref["ms_per_step"], "note": "VD_MATH=fp32: every matrix product on v_mfma_f32_32x32x2_f32"}
        except Exception:                                            # noqa: BLE001 - the headline run must not depend on it
            fp32_ref = {"error": (child.stderr or child.stdout)[-300:]}

    rank, local_rank, world = vdist.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py measures the HIP engine: it needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    cfg = headline_cfg()
    B, T, n_obs = args.batch, args.frames, 4
    model, diff = vda.create_video_model_and_diffusion(**cfg)
    model.to(device).eval()
    specs = model.param_specs()
    sd_holder = {}

    def make_sd():
        sd_holder["sd"] = {k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in specs}
        return sd_holder["sd"]

    vdist.share_weights(model, make_sd, rank)           # one RCCL broadcast of the packed buffer

    kw = make_window(B, T, cfg["image_size"], n_obs, seed=1234 + rank, device=device)
    stepper = Stepper(model, diff, kw, seed=5 + rank)
    nts = diff.num_timesteps
    order = list(range(nts))[::-1]

    for i in range(args.warmup):
        stepper.step(order[i % nts])
    torch.cuda.synchronize()
    vdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        stepper.step(order[(args.warmup + i) % nts])
    torch.cuda.synchronize()
    vdist.barrier()
    torch.cuda.synchronize()
    elapsed = vdist.max_over_ranks(time.perf_counter() - t0, device=device)
    assert torch.isfinite(stepper.bufs[stepper.count & 1]).all()

    roofline, classes = None, None
    if not args.no_roofline:
        classes = profile_step(stepper, order[0])
        if rank == 0:
            name = max((k for k in classes if k.startswith(("igemm", "conv3x3"))), key=lambda k: classes[k]["ms"])
            c = classes[name]
            achieved = c["gflop"] / c["ms"]                                   # GFLOP/ms = TFLOP/s
            # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
            # runs, gfx950 x2 correction on FETCH_SIZE; tools/profile_bench.sh): PMC cannot be read inside this process
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_dominant.json")
            if os.path.exists(pmc):
                rec = json.load(open(pmc))
                if rec.get("kernel") == name:
                    traffic = round(rec["hbm_bytes_per_launch"])
            split_conv = name == "conv3x3_wino_s64_kernel"
            # the dominant kernel runs on the bf16 matrix pipe (fp32 operands split exactly into three bf16 pieces) unless
            # VD_CONV_SPLIT=0 / VD_MATH=fp32 keep it on the fp32 MFMA: `peak` is the dense peak of the pipe it uses
            peak = PEAK_BF16_MFMA_TFLOPS if split_conv else PEAK_FP32_MFMA_TFLOPS
            roofline = dict(bound="mfma", kernel=name, achieved=round(achieved, 2), peak=peak,
                            unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=traffic,
                            launches_per_step=c["launches"], avg_launch_us=round(1e3 * c["ms"] / c["launches"], 1),
                            alg_gflop_per_launch=round(c["gflop"] / c["launches"], 3),
                            alg_mb_per_launch=round(c["mb"] / c["launches"], 2))
            if name.startswith("conv3x3_wino"):
                # `achieved` counts the ALGORITHMIC flops of a direct fp32 3x3 convolution (2*M*Cout*Cin*9).  Winograd
                # F(2x2,3x3) executes 16/36 of the multiplications; the split kernel spends six bf16 piece products
                # on each, so the matrix pipe executes achieved * 6 / 2.25 bf16 flops (fp32 kernel: achieved / 2.25)
                ex = achieved * (6 / 2.25 if split_conv else 1 / 2.25)
                roofline["mfma_executed_tflops"] = round(ex, 2)
                roofline["mfma_executed_frac"] = round(ex / peak, 4)
                roofline["fp32_mfma_peak"] = PEAK_FP32_MFMA_TFLOPS          # what a plain fp32 kernel is bounded by
                roofline["achieved_over_fp32_mfma_peak"] = round(achieved / PEAK_FP32_MFMA_TFLOPS, 4)
    vdist.barrier()

    if rank != 0:
        return
    value = world * args.steps / elapsed
    line = {
        "metric": "denoise-steps/sec", "value": round(value, 4), "unit": "denoise-steps/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "arithmetic": "every matrix product on the fp32 MFMA" if os.environ.get("VD_MATH") == "fp32" else
                      "fp32 operands and fp32 accumulation throughout; matrix products (3x3 convs as Winograd F(2x2,3x3), "
                      "linear layers, 1x1 and stride-2 convs): fp32 operands split EXACTLY into three bf16 pieces, six piece "
                      "products on the bf16 MFMA with fp32 accumulation (error vs fp64 <= that of the fp32 MFMA; "
                      "tests/test_gpu_ops.py)" + ("; VD_CONV_SPLIT=0: 3x3 convs on the fp32 MFMA"
                                                  if os.environ.get("VD_CONV_SPLIT") == "0" else ""),
        "sec_per_clip_batch": round(250 * elapsed / args.steps, 2),
        "config": {"workload": "BASELINE configs[1]: BAIR-shaped 64x64, T=16 (4 obs + 12 latent), batch 8 per GPU, "
                               "ddim250 respacing, p_sample, independent mode, default 116M-param video UNet",
                   "batch_per_gpu": B, "frames": T, "image_size": 64, "respaced_steps": nts,
                   "parallelism": f"batch-shard x{world} (no collective in the step)"},
        "roofline": roofline,
    }
    if fp32_ref is not None:
        line["fp32_mfma_only"] = fp32_ref
    if classes is not None:
        line["kernel_classes"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                      "tflops": round(v["gflop"] / v["ms"], 2) if v["gflop"] else None,
                                      "gbs": round(v["mb"] / v["ms"], 1)} for k, v in classes.items()}
    if world == 1 and not args.no_cpu_baseline:
        sd = sd_holder.get("sd") or make_sd()
        line["cpu_baseline"] = cpu_baseline(cfg, sd, B, T, n_obs, order[0])
        line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
