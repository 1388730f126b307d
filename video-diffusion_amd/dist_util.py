"""The `dist_util` names the reference's sampling / NLL scripts use (improved_diffusion/dist_util.py:53-64,82-143), on
`torch.distributed` alone (no MPI, no blobfile): `load_state_dict` for a checkpoint file, `dev()` for this process'
device, `setup_dist` for joining the one-process-per-GPU job.  The engine's own start-up path does not go through
`sync_params` -- its weights travel as ONE packed buffer (`dist.share_weights`) -- but the function is here for scripts
that call it on ordinary tensors."""
import io

import torch
import torch.distributed as tdist

from . import dist as _vdist


def is_dist_avail_and_initialized():
    return tdist.is_available() and tdist.is_initialized()


def get_world_size():
    return tdist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return tdist.get_rank() if is_dist_avail_and_initialized() else 0


def setup_dist(backend=None):
    """Join the job RANK / LOCAL_RANK / WORLD_SIZE describe (dist_util.py:82-111): the GPU is picked before the group exists."""
    return _vdist.init(backend=backend)


def dev():
    """This process' device (dist_util.py:114-119): the current GPU -- `setup_dist` / `dist.init` made LOCAL_RANK's current."""
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def load_state_dict(path, **kwargs):
    """`torch.load` of a checkpoint file whose bytes only rank 0 fetches (dist_util.py:122-136 does this over MPI; here the
    bytes go out with `broadcast_object_list` when a process group exists, and it is a plain read otherwise)."""
    data = None
    if get_rank() == 0:
        with open(path, "rb") as f:
            data = f.read()
    if get_world_size() > 1:
        data = _vdist.broadcast_object(data, src=0)
    return torch.load(io.BytesIO(data), **kwargs)


def sync_params(params):
    """dist_util.py:139-143: one broadcast per tensor from rank 0."""
    if get_world_size() > 1:
        with torch.no_grad():
            for p in params:
                tdist.broadcast(p, 0)
