"""CPU, world_size 2 over gloo: the N>1 path of bench.py / the sampling driver -- task sharding, the single
packed-weight broadcast, barrier and max-over-ranks timing (video-diffusion_amd/dist.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from video_diffusion_amd import dist as vdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """Stands in for the engine handle: what share_weights touches (no GPU in this test)."""

    def __init__(self):
        self.buf = torch.zeros(1000)
        self.received = False

    def load_state_dict(self, sd):
        self.buf.copy_(sd["w"])

    def packed_weights(self):
        return self.buf

    def mark_weights_received(self):
        self.received = True

    def weights_layout_id(self):
        return 42


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = vdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    model = _FakeModel()
    built = []

    def make_sd():
        built.append(1)
        return {"w": torch.arange(1000, dtype=torch.float32) * 0.5}

    vdist.share_weights(model, make_sd, rank)
    assert torch.equal(model.buf, torch.arange(1000, dtype=torch.float32) * 0.5)
    assert len(built) == (1 if rank == 0 else 0)            # only rank 0 touches the checkpoint
    assert model.received == (rank != 0)
    vdist.barrier()
    slow = vdist.max_over_ranks(1.0 + rank)                 # the slowest rank defines the job time
    total = vdist.sum_over_ranks(10.0)
    tasks = vdist.task_ids(7, rank, world)
    out.put((rank, slow, total, tasks))
    dist.destroy_process_group()


def test_two_rank_sharding_and_weight_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [2.0, 2.0] and [r[2] for r in res] == [20.0, 20.0]
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]
    assert sorted(res[0][3] + res[1][3]) == list(range(7))   # a partition: nothing dropped, nothing doubled


def _tiny_model():
    import video_diffusion_amd as vda
    cfg = vda.video_model_and_diffusion_defaults()
    cfg.update(T=4, image_size=32, num_channels=32, num_res_blocks=1, rp_alpha=4, rp_beta=4, rp_gamma=4, timestep_respacing="ddim5")
    model, _ = vda.create_video_model_and_diffusion(**cfg)
    return vda, model


def _engine_worker(rank, world, port, out, env):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), **env.get(rank, {}))
    vdist.init(backend="gloo")
    vda, model = _tiny_model()
    specs = model.param_specs()

    def make_sd():
        return {k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in specs}

    try:
        vdist.share_weights(model, make_sd, rank)          # the REAL packer -> one broadcast -> mark_weights_received
    except RuntimeError as e:
        out.put((rank, "error", str(e)[:80]))
        dist.destroy_process_group()
        return
    got = model.packed_weights().clone()
    # what this rank would have packed by itself from the same checkpoint: the broadcast payload (bf16 bit patterns
    # carried in an fp32 tensor) must arrive byte for byte
    _, own = _tiny_model()
    own.load_state_dict(make_sd())
    want = own.pack_on_host()
    from video_diffusion_amd import _lib
    missing = _lib.lib().vd_weights_missing(model._handle)
    out.put((rank, "ok", bool(torch.equal(got.view(torch.int32), want.view(torch.int32))), int(got.numel()), missing,
             int(got.view(torch.int32).ne(0).sum())))
    dist.destroy_process_group()


def _run_engine_ranks(env):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q, env)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_real_packed_weights_broadcast_is_byte_exact():
    """The engine's own packed image (kernel-ready layouts: bf16 piece planes inside an fp32 buffer) built on rank 0 by
    the C ABI's packer in host memory, broadcast once over gloo, marked received on rank 1: bit-identical to what rank 1
    packs by itself, nothing missing afterwards."""
    res = _run_engine_ranks({})
    for rank, status, same, numel, missing, nonzero in res:
        assert status == "ok" and same and missing == 0 and numel > 2_000_000 and nonzero > numel // 2, res


def test_layout_mismatch_between_ranks_is_refused():
    """a rank whose environment selects another arithmetic mode lays the buffer out differently; it must not
    accept rank 0's bytes."""
    res = _run_engine_ranks({1: {"VD_MATH": "fp32"}})
    assert [r[1] for r in res] == ["error", "error"] and all("layout" in r[2] for r in res), res


class _PassThroughDiffusion:
    """Stands in for the sampler in the CLI test: the step leaves x unchanged (no GPU here)."""
    num_timesteps = 2

    def p_sample(self, model, x, t, **kw):
        return {"sample": x}


def _cli_worker(rank, world, port, out, ckpt, workdir):
    import argparse
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.chdir(workdir)
    import video_diffusion_amd as vda
    from video_diffusion_amd import video_sample
    opened = []
    real_load = torch.load

    def spy_load(path, *a, **k):
        opened.append(str(path))
        return real_load(path, *a, **k)

    torch.load = spy_load

    def create(**kw):                    # the REAL engine handle (packer, layout id) with a pass-through sampler
        model, _ = vda.create_video_model_and_diffusion(**kw)
        return model, _PassThroughDiffusion()

    args = argparse.Namespace(checkpoint_path=ckpt, inference_mode="autoreg", T=6, max_frames=4, obs_length=2, step_size=2,
                              batch_size=1, num_videos=3, timestep_respacing="ddim5", observed_frames="x_0", image_size=32,
                              num_channels=32, num_res_blocks=1, seed=0, adaptive_distance="l2", executor="eager",
                              eval_dir=None, out_dir=None, use_ddim=False, sample_idx=0)
    out_dir = video_sample.run(args, create=create, device=torch.device("cpu"))
    files = sorted(os.listdir(os.path.join(str(out_dir), "samples")))
    out.put((rank, opened, str(out_dir), files))
    dist.destroy_process_group()


def test_two_rank_sampling_cli_shards_tasks_and_broadcasts_weights(tmp_path):
    """video_sample.run (the body of the CLI's main) with two ranks over gloo: rank 0 alone opens the checkpoint; its
    config reaches rank 1 as an object, its weights as ONE broadcast of the engine's packed image (the real packer);
    the tasks are dealt r, r+R; the files land under the reference's results/<ckpt subpath>/<stem>_<step>_respace<X>/
    <mode>_<max_frames>_<step_size>_<T>_<obs_length>/samples naming (test_util.py:65-132)."""
    vda, model = _tiny_model()
    cfg = vda.video_model_and_diffusion_defaults()
    cfg.update(T=4, image_size=32, num_channels=32, num_res_blocks=1, rp_alpha=4, rp_beta=4, rp_gamma=4)
    sd = {k: torch.from_numpy(vda.weights_init.synth_param(k, s)) for k, s in model.param_specs()}
    ckdir = tmp_path / "checkpoints" / "runs" / "abc123"
    ckdir.mkdir(parents=True)
    ckpt = str(ckdir / "ema_0.9999_latest.pt")
    torch.save({"state_dict": sd, "config": cfg, "step": 7000}, ckpt)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, q, ckpt, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, opened0, dir0, files0), (r1, opened1, dir1, files1) = res
    assert opened1 == [] and len(opened0) >= 1 and all(o == ckpt for o in opened0), (opened0, opened1)
    assert dir0 == dir1 == "results/runs/abc123/ema_0.9999_latest_7000_respaceddim5/autoreg_4_2_6_2"
    assert files1 == ["sample_0000-0.npy", "sample_0001-0.npy", "sample_0002-0.npy"]      # rank 0: videos 0, 2; rank 1: video 1
    import json
    import numpy as np
    mc = json.load(open(tmp_path / dir0 / "model_config.json"))
    assert mc["num_channels"] == 32 and mc["timestep_respacing"] == "ddim5"
    a = np.load(tmp_path / dir0 / "samples" / "sample_0001-0.npy")
    assert a.dtype == np.uint8 and a.shape == (6, 3, 32, 32)


def test_task_to_indices_mapping():
    """video_sample.py:577-582: indices = range(task_id*bs, (task_id+1)*bs)."""
    assert vdist.indices_for_task(3, 8) == list(range(24, 32))
    assert vdist.indices_for_task(3, 8, dataset_len=27) == [24, 25, 26]
    assert vdist.task_ids(5, 0, 1) == [0, 1, 2, 3, 4]
    assert vdist.max_over_ranks(3.5) == 3.5                  # single-process fallthrough


def test_bench_rank_environment_is_well_formed():
    """The N-rank child environment bench.py's self_launch builds (what the driver's torch.distributed.run would set)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (1, 2, 4, 8):
        envs = [bench.rank_env({"PATH": "/x", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, r, world, 29511) for r in range(world)]
        assert sorted(int(e["RANK"]) for e in envs) == list(range(world))
        for r, e in enumerate(envs):
            assert e["LOCAL_RANK"] == e["RANK"] == str(r) and e["WORLD_SIZE"] == e["LOCAL_WORLD_SIZE"] == str(world)
            assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"     # the container hostname may not resolve
            assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/x"       # dmabuf IPC: RCCL fails without it on this pool
    assert bench.rank_env({}, 0, 1, 1)["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    import pytest
    with pytest.raises(AssertionError):
        bench.rank_env({}, 2, 2, 29500)


def test_dist_init_picks_the_gpu_before_the_group_exists(monkeypatch):
    """RCCL: torch.cuda.set_device(local_rank) FIRST, then init_process_group(device_id=that device) -- the order of the
    reference's dist_util.py:100-110; a group without device_id guesses its GPU at the first collective."""
    calls = []
    monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("LOCAL_RANK", "3"); monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append(("set_device", d)))
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    monkeypatch.setattr(dist, "init_process_group", lambda **kw: calls.append(("init", kw)))
    monkeypatch.setattr(dist, "get_world_size", lambda: 8)
    monkeypatch.setattr(dist, "get_rank", lambda: 3)
    assert vdist.init() == (3, 3, 8)
    assert calls[0] == ("set_device", 3) and calls[1][0] == "init"
    kw = calls[1][1]
    assert kw["backend"] == "nccl" and kw["rank"] == 3 and kw["world_size"] == 8 and kw["device_id"] == torch.device("cuda", 3)
    # a rank whose GPU does not exist fails before it joins (not inside the first collective)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    import pytest
    with pytest.raises(RuntimeError, match="LOCAL_RANK 3"):
        vdist.init()
    # the one-GPU rehearsal: gloo, every rank on device 0, no device_id
    calls.clear()
    assert vdist.init(backend="gloo", device_index=0) == (3, 3, 8)
    assert calls[0] == ("set_device", 0) and "device_id" not in calls[1][1] and calls[1][1]["backend"] == "gloo"
