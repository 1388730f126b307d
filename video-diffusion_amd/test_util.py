"""Where an evaluation run keeps its files -- the on-disk contract that `video_eval.py` of the reference reads
(improved_diffusion/test_util.py:18-132): the results directory derived from the checkpoint path and the sampling
options, the run identifier derived from the inference schedule, and the lock that guards files shared between the
one-process-per-GPU workers.  Host-side string/file logic only."""
import os
from pathlib import Path

import torch

try:                                   # the reference subclasses filelock.FileLock (test_util.py:18-29)
    from filelock import FileLock as _FileLock
except ImportError:                    # pragma: no cover - filelock ships with torch's dependencies
    _FileLock = None


if _FileLock is not None:
    class Protect(_FileLock):
        """Lock `<path>.lock` next to `path` while a shared file is written (test_util.py:18-29)."""

        def __init__(self, path, timeout=2, **kwargs):
            path = Path(path)
            super().__init__(path.parent / f"{path.name}.lock", timeout=timeout, **kwargs)
else:                                  # pragma: no cover
    class Protect:
        def __init__(self, path, timeout=2, **kwargs):
            path = Path(path)
            self.lock_path, self.timeout, self._fd = path.parent / f"{path.name}.lock", timeout, None

        def __enter__(self):
            import time
            t0 = time.time()
            while True:
                try:
                    self._fd = os.open(self.lock_path, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
                    return self
                except FileExistsError:
                    if time.time() - t0 > self.timeout:
                        raise TimeoutError(str(self.lock_path))
                    time.sleep(0.05)

        def __exit__(self, *exc):
            os.close(self._fd)
            os.unlink(self.lock_path)


def get_model_results_path(args, postfix=""):
    """results/<checkpoint path after the first '*checkpoint*' component, minus the file>/<stem>[_<step>][_ddim][_respace<X>]
    -- or args.eval_dir untouched when it is given (test_util.py:65-108).  `args`: use_ddim, timestep_respacing,
    eval_dir, checkpoint_path.  A checkpoint named '*latest' gets its training step appended (read from the file)."""
    if args.use_ddim:
        postfix += "_ddim"
    if args.timestep_respacing != "":
        postfix += "_" + f"respace{args.timestep_respacing}"
    if args.eval_dir is not None:
        return Path(args.eval_dir)
    checkpoint_path = Path(args.checkpoint_path)
    name = f"{checkpoint_path.stem}"
    if name.endswith("latest"):
        name += f"_{torch.load(args.checkpoint_path, map_location='cpu')['step']}"
    if postfix != "":
        name += postfix
    path = None
    for idx, x in enumerate(checkpoint_path.parts):
        if "checkpoint" in x:
            path = Path(*(checkpoint_path.parts[idx + 1:]))
            break
    assert path is not None
    return Path("results") / path.parent / name


def get_eval_run_identifier(args, postfix=""):
    """<mode>[_optimal-<o>]_<max_frames>_<step_size>_<T>_<obs_length> with the trainset_/gradientmethod_/<dataset>_
    prefixes in the reference's order (test_util.py:111-132)."""
    res = args.inference_mode
    if hasattr(args, "optimality") and args.optimality is not None:
        res += f"_optimal-{args.optimality}"
    res += f"_{args.max_frames}_{args.step_size}_{args.T}_{args.obs_length}"
    if hasattr(args, "dataset_partition") and args.dataset_partition == "train":
        res = "trainset_" + res
    if hasattr(args, "use_gradient_method") and args.use_gradient_method:
        res = "gradientmethod_" + res
    if hasattr(args, "override_dataset") and args.override_dataset is not None:
        res = f"{args.override_dataset}_" + res
    if postfix != "":
        res += postfix
    return res
