"""Frame schedulers: which frames a window conditions on and which it generates.

Integer-only host logic mirroring `improved_diffusion/inference_util.py`
(base protocol :34-134; strategies :232-418; registry :779-799) and pinned
bit-exactly by tests/golden/schedulers.json.  A strategy is an iterator of
`(obs_frame_indices, latent_frame_indices)` pairs; `scripts/video_sample.py:75-97`
drives it with `iter()` / `next()`.

The adaptive (LPIPS-driven), goal-directed and visualisation strategies of the
reference need a perceptual network and are outside the hot path.
"""
import numpy as np


class InferenceStrategyBase:
    """inference_util.py:34-134: bookkeeping of finished frames + sanity checks."""

    def __init__(self, video_length, num_obs, max_frames, step_size, optimal_schedule_path=None):
        if optimal_schedule_path is not None:
            import torch
            self.optimal_schedule = torch.load(optimal_schedule_path)
        else:
            self.optimal_schedule = None
        self._video_length = video_length
        self._max_frames = max_frames
        self._num_obs = num_obs
        self._step_size = step_size
        self._done_frames = set(range(num_obs))
        self._obs_frames = list(range(num_obs))
        self._current_step = 0

    # -- protocol ------------------------------------------------------------------------------------
    def __iter__(self):
        self.step = 0
        return self

    def is_done(self):
        return len(self._done_frames) >= self._video_length

    def get_unconditional_indices(self):
        return list(range(self._max_frames))

    def next_indices(self):
        raise NotImplementedError

    @property
    def typename(self):
        return type(self).__name__

    def __next__(self):
        if self.is_done():
            raise StopIteration
        first_unconditional = self._num_obs == 0 and self._current_step == 0
        if first_unconditional:
            obs, latent = [], self.get_unconditional_indices()     # a whole window from pure noise (:85-90)
        else:
            obs, latent = self.next_indices()
            if self.optimal_schedule is not None:                  # --optimality override (:94-103)
                obs = self.optimal_schedule.get(self._current_step, [])
        assert isinstance(obs, list) and isinstance(latent, list)
        for idx in obs:
            assert idx in self._done_frames, (
                f"Attempting to condition on frame {idx} while it is not generated yet.\n"
                f"Generated frames: {self._done_frames}\nObserving: {obs}\nGenerating: {latent}")
        assert np.all(np.array(latent) < self._video_length)
        self._done_frames.update(latent)
        if first_unconditional:
            self._obs_frames = latent
        self._current_step += 1
        return obs, latent

    # -- shared helper ---------------------------------------------------------------------------------
    def _next_chunk(self, first, count):
        return list(range(first, min(first + count, self._video_length)))


class Autoregressive(InferenceStrategyBase):
    """:232-245 -- condition on the newest (max_frames - step_size) finished frames."""

    def next_indices(self):
        if not self._done_frames:
            return [], list(range(self._max_frames))
        obs = sorted(self._done_frames)[-(self._max_frames - self._step_size):]
        return obs, self._next_chunk(obs[-1] + 1, self._step_size)


class Independent(InferenceStrategyBase):
    """:248-259 -- always condition on the originally observed frames only."""

    def next_indices(self):
        obs = sorted(self._obs_frames)[-(self._max_frames - self._step_size):]
        return obs, self._next_chunk(max(self._done_frames) + 1, self._step_size)


class ReallyIndependent(InferenceStrategyBase):
    """:262-272 -- no conditioning at all, max_frames new frames per window."""

    def next_indices(self):
        return [], self._next_chunk(max(self._done_frames) + 1, self._max_frames)


class ExpPast(InferenceStrategyBase):
    """:275-293 -- frames at distances 1, 2, 4, ... back, then nearest-first fill up to max_frames.
    The observed list is NOT sorted; the network sees it in this order."""

    def next_indices(self):
        cur = max(self._done_frames) + 1
        obs = list(cur - 2 ** np.arange(int(np.log2(cur))))
        latent = list(range(cur, cur + min(self._step_size, self._video_length)))
        for back in range(1, cur + 1):
            if len(obs) + len(latent) >= self._max_frames:
                break
            if cur - back not in obs:
                obs.append(cur - back)
        return obs, latent


class MixedAutoregressiveIndependent(InferenceStrategyBase):
    """:296-312 -- half of the conditioning budget from the newest frames, the rest from the observed ones."""

    def next_indices(self):
        budget = self._max_frames - self._step_size
        chosen = set(sorted(self._done_frames)[-(budget // 2):])
        for i in sorted(self._obs_frames)[::-1]:
            chosen.add(i)
            if len(chosen) == budget:
                break
        return sorted(chosen), self._next_chunk(max(self._done_frames) + 1, self._step_size)


class HierarchyNLevel(InferenceStrategyBase):
    """:315-418 -- coarse-to-fine: level 1 spreads `step_size` latents over the whole video, deeper
    levels fill the gaps at geometrically shrinking strides, always conditioning on both sides."""
    N = None

    def _start_level_one(self, last):
        self.current_level = 1
        self.last_sampled_idx = last

    def get_unconditional_indices(self):
        self._start_level_one(self._video_length - 1)
        return [int(i) for i in np.linspace(0, self._video_length - 1, self._max_frames)]

    @property
    def sample_every(self):
        level1 = (self._video_length - len(self._obs_frames)) / (self._step_size - 1)
        return int(level1 ** ((self.N - self.current_level) / (self.N - 1)))

    def next_indices(self):
        L, done = self._video_length, self._done_frames
        if not done:
            self._start_level_one(L - 1)
            return [], [int(i) for i in np.linspace(0, L - 1, self._max_frames)]
        if len(done) == len(self._obs_frames):
            self._start_level_one(max(self._obs_frames))
        n_cond, n_new = self._max_frames - self._step_size, self._step_size

        idx = self.last_sampled_idx + self.sample_every
        if all(i in done for i in range(idx, L)):
            self.current_level += 1                                 # nothing left after idx: next, finer level
            self.last_sampled_idx = 0
            idx = min(i for i in range(L) if i not in done) - 1 + self.sample_every
        if self.current_level == 1:
            latent = [int(i) for i in np.linspace(max(self._obs_frames) + 1, L - 0.001, n_new)]
        else:
            latent = []
            while len(latent) < n_new and idx < L:
                if idx in done:
                    idx += 1
                else:
                    latent.append(idx)
                    idx += self.sample_every

        obs = [i for i in range(min(latent), max(latent)) if i in done]      # anything finished in between
        room = n_cond - len(obs)
        if room < 2:                                                # need one frame before AND after: shrink the step
            if self._step_size == 1:
                raise Exception("Cannot condition before and after even with step size of 1")
            self._step_size -= 1
            try:
                return self.next_indices()
            finally:
                self._step_size += 1
        obs.extend([i for i in range(max(latent) + 1, L) if i in done][:room // 2])
        n_before = n_cond - len(obs)
        if self.current_level == 1:
            obs.extend(list(np.linspace(0, max(self._obs_frames) + 0.999, n_before).astype(np.int32)))
        else:
            obs.extend([i for i in range(min(latent) - 1, -1, -1) if i in done][:n_before])
        self.last_sampled_idx = max(latent)
        return obs, latent

    @property
    def typename(self):
        return f"{super().typename}-{self.N}"


def get_hierarchy_n_level(n):
    return type(f"Hierarchy{n}Level", (HierarchyNLevel,), {"N": n})


# inference_util.py:779-799 (the strategies that need no perceptual network)
inference_strategies = {
    "autoreg": Autoregressive,
    "independent": Independent,
    "really-independent": ReallyIndependent,
    "exp-past": ExpPast,
    "mixed-autoreg-independent": MixedAutoregressiveIndependent,
    "hierarchy-2": get_hierarchy_n_level(2),
    "hierarchy-3": get_hierarchy_n_level(3),
    "hierarchy-4": get_hierarchy_n_level(4),
    "hierarchy-5": get_hierarchy_n_level(5),
}
