"""Closed-form synthetic weights for a checkpoint-less model.

No checkpoints ship with the reference and an untrained
reference model emits eps == 0 because of its zero-initialised layers (F10:
`zero_module`, unet.py:155,419,747; RPENet.out, unet.py:278-279).  Every
parameter is therefore filled from a generator keyed only by the parameter's
checkpoint name and shape, so the reference (in the fixture generator), the
oracle and the HIP engine can be driven by bit-identical weights on any box.
"""
import zlib

import numpy as np


def synth_param(name, shape):
    """float32 ndarray for checkpoint entry `name` of `shape` (deterministic)."""
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    shape = tuple(int(s) for s in shape)
    u = rng.random(size=shape, dtype=np.float32) * np.float32(2) - np.float32(1)
    if name == "spatial_encoding":
        return u * np.float32(0.5)
    if name.endswith("lookup_table_weight"):
        return u * np.float32(0.5)
    if name.endswith(".bias"):
        return u * np.float32(0.1)
    if len(shape) == 1:                     # GroupNorm gain
        return np.float32(1) + u * np.float32(0.2)
    fan_in = int(np.prod(shape[1:]))
    return u * np.float32(np.sqrt(3.0 / fan_in))


def synth_state_dict(specs):
    """specs: iterable of (name, shape) -> {name: float32 ndarray}."""
    return {n: synth_param(n, s) for n, s in specs}
