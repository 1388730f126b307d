#!/usr/bin/env python3
"""Golden sequences of the ADAPTIVE frame schedulers from the imported reference (build container only).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 /root/repo/tools/gen_golden_adaptive.py

`adaptive-autoreg` / `adaptive-hierarchy-N` (inference_util.py:137-229,421-531) pick the observed frames of a window
per batch item by farthest-point selection on frame embeddings.  The reference supports two embeddings: the raw frames
(distance='l2') and LPIPS features (distance='lpips', what scripts/video_sample.py passes; needs the pretrained AlexNet
of the `lpips` package, which is not available offline).  The l2 variant is pure arithmetic on the sample tensor and is
what is pinned here: seeded synthetic videos, the full (obs per item, latents per item) sequence or the exception type.
"""
import json
import os
import signal
import sys
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")

lp = types.ModuleType("lpips")
lp.LPIPS = type("LPIPS", (torch.nn.Module,), {})
lp.normalize_tensor = lambda x: x
sys.modules["lpips"] = lp

from improved_diffusion import inference_util as iu  # noqa: E402


def videos(seed, B, T):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, T, 3, 4, 4, generator=g) * 2 - 1


def main():
    cases = []
    for mode, (T, n_obs, max_frames, step), B, seed in [
            ("adaptive-autoreg", (20, 4, 8, 3), 2, 1), ("adaptive-autoreg", (16, 0, 6, 2), 3, 2), ("adaptive-autoreg", (30, 6, 10, 5), 1, 3),
            ("adaptive-hierarchy-2", (30, 4, 8, 4), 2, 4), ("adaptive-hierarchy-2", (40, 6, 12, 5), 2, 5),
            ("adaptive-hierarchy-3", (60, 6, 12, 4), 1, 6), ("adaptive-hierarchy-2", (20, 0, 8, 4), 2, 7)]:
        v = videos(seed, B, T)
        rec = dict(mode=mode, args=[T, n_obs, max_frames, step], B=B, seed=seed)
        seq = []

        def on_alarm(signum, frame):
            raise TimeoutError("the reference does not terminate on this case")
        signal.signal(signal.SIGALRM, on_alarm)
        signal.alarm(20)
        try:
            it = iter(iu.inference_strategies[mode](distance="l2", video_length=T, num_obs=n_obs, max_frames=max_frames,
                                                    step_size=step, optimal_schedule_path=None))
            while len(seq) < 200:
                it.set_videos(v)
                try:
                    obs, lat = next(it)
                except StopIteration:
                    break
                seq.append([[[int(i) for i in o] for o in obs], [[int(i) for i in l] for l in lat]])
            rec["seq"] = seq
        except Exception as e:  # noqa: BLE001 -- record what the reference does, including failures
            rec["error"] = type(e).__name__
            rec["seq_before_error"] = seq
        finally:
            signal.alarm(0)
        cases.append(rec)
        print(mode, rec["args"], "steps", len(rec.get("seq", rec.get("seq_before_error", []))), rec.get("error"))
    json.dump(dict(cases=cases), open(os.path.join(OUT, "schedulers_adaptive.json"), "w"))


if __name__ == "__main__":
    main()
