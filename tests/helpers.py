"""Shared helpers for the parity tests: fixture loading, oracle construction."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Tolerance of the whole tier: single-step / per-op fp32 parity.
ATOL = 1e-4
RTOL = 1e-4


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def case_inputs(rec, ci):
    pre = f"c{ci}_"
    d = {k[len(pre):]: v for k, v in rec.items() if k.startswith(pre)}
    out = {k: torch.from_numpy(v) for k, v in d.items() if k not in ("observed_frames",)}
    out["observed_frames"] = str(d["observed_frames"])
    return out


def n_cases(rec):
    return len({k.split("_")[0] for k in rec if k.startswith("c") and k.split("_")[0][1:].isdigit()})


def synth_sd(specs):
    import video_diffusion_amd as vda
    return {n: torch.from_numpy(vda.weights_init.synth_param(n, s)) for n, s in specs}


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    lim = atol + rtol * np.abs(b)
    bad = err > lim
    assert not bad.any(), f"max|d|={err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)} " \
                          f"(ref {b.flat[err.argmax()]:.5f}); {bad.sum()} of {bad.size} outside atol={atol} rtol={rtol}"
    return float(err.max())
