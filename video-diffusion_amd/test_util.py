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


def _checkpoint_step(checkpoint_path):
    """The training step a checkpoint file records ('step' of the saved dict)."""
    return torch.load(checkpoint_path, map_location="cpu")["step"]


def get_model_results_path(args, postfix=""):
    """Where the samples of a checkpoint go (contract of test_util.py:65-108, pinned by tests/golden/eval_paths.json):
    `args.eval_dir` untouched when given; otherwise results/<dirs>/<label> with <dirs> = the directories of
    `args.checkpoint_path` strictly between its first component containing 'checkpoint' and the file, and
    <label> = <file stem>[_<training step> for a '*latest' file]<postfix>[_ddim][_respace<timestep_respacing>]."""
    if args.eval_dir is not None:
        return Path(args.eval_dir)
    ckpt = Path(args.checkpoint_path)
    marker = next((k for k, part in enumerate(ckpt.parts) if "checkpoint" in part), None)
    if marker is None:
        raise AssertionError(f"no '*checkpoint*' component in {ckpt}")
    label = [ckpt.stem]
    if ckpt.stem.endswith("latest"):
        label.append(f"_{_checkpoint_step(args.checkpoint_path)}")
    label.append(postfix)
    label.append("_ddim" if args.use_ddim else "")
    label.append(f"_respace{args.timestep_respacing}" if args.timestep_respacing != "" else "")
    return Path("results", *ckpt.parts[marker + 1:-1], "".join(label))


def get_eval_run_identifier(args, postfix=""):
    """[<override_dataset>_][gradientmethod_][trainset_]<mode>[_optimal-<o>]_<max_frames>_<step_size>_<T>_<obs_length><postfix>
    (contract of test_util.py:111-132; attributes that an `args` does not carry count as unset)."""
    dataset = getattr(args, "override_dataset", None)
    optimality = getattr(args, "optimality", None)
    head = "".join([
        f"{dataset}_" if dataset is not None else "",
        "gradientmethod_" if getattr(args, "use_gradient_method", False) else "",
        "trainset_" if getattr(args, "dataset_partition", None) == "train" else "",
    ])
    mode = args.inference_mode + (f"_optimal-{optimality}" if optimality is not None else "")
    window = "_".join(str(v) for v in (args.max_frames, args.step_size, args.T, args.obs_length))
    return f"{head}{mode}_{window}{postfix}"
