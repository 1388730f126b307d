"""One process per GPU, test-set sharding, one weight broadcast.

The reference shards SAMPLING by launching one process per GPU with a `--task_id`
(command_launchers.py:32-62 -> scripts/video_sample.py:577-582: indices =
range(task_id*bs, (task_id+1)*bs)); there is no collective in the denoise loop.
Its only parameter exchange is `sync_params`, one `dist.broadcast` per tensor
(dist_util.py:139-143).  Here: `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests), rank r takes tasks r, r+R, ..., and the
EMA weights travel as ONE packed buffer in a single broadcast.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device_index=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).

    With RCCL the process' GPU is chosen BEFORE the group exists and handed to it (`device_id`): a process group created
    without it guesses the device at its first collective from the current one, so a barrier / object broadcast issued before
    `torch.cuda.set_device` would run on GPU 0 from every rank (a hang or an invalid-argument on the 8-GPU node; the
    reference does the same in dist_util.py:100-110: set_device, then init_process_group).  `device_index` overrides
    LOCAL_RANK (the one-GPU rehearsal puts every rank on device 0 over gloo)."""
    rank, local_rank, world = env_world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dev = local_rank if device_index is None else device_index
    if backend == "nccl":
        if not torch.cuda.is_available():
            raise RuntimeError("dist.init: backend nccl (RCCL) needs a GPU")
        if dev >= torch.cuda.device_count():
            raise RuntimeError(f"dist.init: LOCAL_RANK {local_rank} -> device {dev}, but this process sees {torch.cuda.device_count()} GPUs")
    if torch.cuda.is_available() and dev < torch.cuda.device_count():
        torch.cuda.set_device(dev)             # (a gloo job on a box with fewer GPUs than ranks keeps the current device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        assert dist.get_world_size() == world and dist.get_rank() == rank
    return rank, local_rank, world


def task_ids(num_tasks, rank, world):
    """Tasks (= dataloader batches, video_sample.py:577-582) owned by `rank`: r, r+R, r+2R, ..."""
    return list(range(rank, num_tasks, world))


def indices_for_task(task_id, batch_size, dataset_len=None):
    """video_sample.py:577-582: the dataset indices of one task."""
    idx = list(range(task_id * batch_size, (task_id + 1) * batch_size))
    if dataset_len is not None:
        idx = [i for i in idx if i < dataset_len]
    return idx


def broadcast_packed(buf, src=0):
    """One collective for the whole parameter set (464 MB fp32 for the default 64x64 model)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(buf, src=src)
    return buf


def broadcast_object(obj, src=0):
    """A small picklable object (the checkpoint's `config` dict) from `src` to every rank; identity in a 1-rank job."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return obj


def share_weights(model, state_dict_fn, rank):
    """Rank 0 materialises the state_dict (disk / generator) and uploads it; everyone else receives the
    engine's packed device buffer over RCCL and marks it loaded."""
    if rank == 0:
        model.load_state_dict(state_dict_fn())
    buf = model.packed_weights()
    if dist.is_initialized() and dist.get_world_size() > 1:
        # the packed layout depends on per-process environment (VD_MATH): every rank must hold
        # rank 0's layout before it accepts rank 0's bytes
        # (checked collectively, so that EVERY rank raises instead of one rank leaving the others in the broadcast)
        mine = torch.tensor([model.weights_layout_id() & 0x7FFFFFFFFFFFFFFF, buf.numel()], dtype=torch.int64, device=buf.device)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo.cpu(), hi.cpu()):
            raise RuntimeError(f"rank {rank}: packed-weight layout differs between ranks (VD_MATH "
                               f"must agree on every rank): mine {mine.tolist()}, job min {lo.tolist()} max {hi.tolist()}")
    broadcast_packed(buf, src=0)
    if rank != 0:
        model.mark_weights_received()     # state_dict() stays rank 0's: other ranks hold only the packed device image
    # use_gradient_method: the backward-data image travels the same way (only when the option is on: enable_guidance())
    # Whether it is on is RANK 0's decision and travels first: ranks that disagreed would issue different collective sequences
    # (one more broadcast on some of them) and hang or corrupt the next collective instead of raising
    bwd = model.guidance_weights() if hasattr(model, "guidance_weights") else None
    want = bwd is not None
    if dist.is_initialized() and dist.get_world_size() > 1:
        flag = [want]
        dist.broadcast_object_list(flag, src=0)
        if flag[0] and not want:
            model.enable_guidance()       # rank 0 has it: allocate the buffer the image is about to land in
            bwd = model.guidance_weights()
        want = bool(flag[0])
    if want:
        broadcast_packed(bwd, src=0)
        if rank != 0:
            model.mark_guidance_received()
    return model


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device=None):
    """Timing reduction of the bench contract: the slowest rank defines the job's time."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
