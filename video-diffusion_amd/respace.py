"""Timestep respacing: host-side mirror of `improved_diffusion/respace.py`.

`space_timesteps` is integer arithmetic and must be bit-exact (pinned by
tests/golden/space_timesteps.json); `SpacedDiffusion` rebuilds the betas of the
retained steps in float64 (respace.py:68-82) and hands the timestep map to the
engine, which applies `_WrappedModel`'s gather + 1000/N rescale on device
(respace.py:111-119) instead of re-uploading the map every step.
"""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """respace.py:7-58.  'ddimN' -> the first integer stride giving exactly N steps;
    'a,b,c' / [a,b,c] -> per-section fractional strides, rounded with round()."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired_count = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                steps = range(0, num_timesteps, stride)
                if len(steps) == desired_count:
                    return set(steps)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    taken, start = set(), 0
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        taken.update(start + round(k_stride) for k_stride in _walk(stride, count))
        start += size
    return taken


def _walk(stride, count):
    pos = 0.0                      # accumulated, not k*stride: the reference's float drift is part of the contract
    for _ in range(count):
        yield pos
        pos += stride


class SpacedDiffusion(GaussianDiffusion):
    """respace.py:61-100."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base_acp = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64), axis=0)
        last, new_betas = 1.0, []
        for i, acp in enumerate(base_acp):
            if i in self.use_timesteps:
                new_betas.append(1 - acp / last)
                last = acp
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _timestep_map_and_scale(self):
        scale = 1000.0 / self.original_num_steps if self.rescale_timesteps else 1.0
        return self.timestep_map, scale

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t                   # scaling is done by the wrapped model (respace.py:98-100)


class _WrappedModel:
    """respace.py:103-119: index -> original step (-> 0..1000 scale), then the network."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps

    def __call__(self, x, timesteps, **kwargs):
        map_tensor = th.tensor(self.timestep_map, device=timesteps.device, dtype=timesteps.dtype)
        new_ts = map_tensor[timesteps]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, timesteps=new_ts, **kwargs)
