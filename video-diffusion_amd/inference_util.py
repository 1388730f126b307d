"""Frame schedulers: which frames a window conditions on and which it generates.

Integer-only host logic mirroring `improved_diffusion/inference_util.py`
(base protocol :34-134; strategies :232-418; registry :779-799) and pinned
bit-exactly by tests/golden/schedulers.json.  A strategy is an iterator of
`(obs_frame_indices, latent_frame_indices)` pairs; `scripts/video_sample.py:75-97`
drives it with `iter()` / `next()`.

The goal-directed, visualisation and frameskip ("google") strategies (:534-776) are
pinned by tests/golden/schedulers_more.json.  The adaptive strategies (:137-229, :421-531) choose the
observed frames per batch item by farthest-point selection on frame embeddings; with distance='l2' (raw
frames) they are pinned by tests/golden/schedulers_adaptive.json, distance='lpips' needs the pretrained
LPIPS network, which does not ship: register one with `set_lpips_embedder`.
"""
import numpy as np

_lpips_embedder = None


def set_lpips_embedder(fn):
    """`fn(frames (B,C,H,W)) -> embedding tensor`: what `LpipsEmbedder(net='alex', spatial=False)` is to the reference
    (inference_util.py:14-31,146-148).  The pretrained network is not available offline, so none is built in."""
    global _lpips_embedder
    _lpips_embedder = fn


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

    def _latents_and_between(self):
        """First half of a hierarchy step (:340-388): the latent grid of the current level and the finished frames that
        lie between its ends.  Returns (first_window_latents, None, None) for the unconditional start."""
        L, done = self._video_length, self._done_frames
        if not done:
            self._start_level_one(L - 1)
            return [int(i) for i in np.linspace(0, L - 1, self._max_frames)], None, None
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
        return latent, obs, n_cond - len(obs)

    def _with_smaller_step(self):
        """Not enough room to condition before AND after the latents: the same step with one latent fewer (:390-400)."""
        if self._step_size == 1:
            raise Exception("Cannot condition before and after even with step size of 1")
        self._step_size -= 1
        try:
            return self.next_indices()
        finally:
            self._step_size += 1

    def next_indices(self):
        L, done = self._video_length, self._done_frames
        latent, obs, room = self._latents_and_between()
        if obs is None:
            return [], latent
        if room < 2:
            return self._with_smaller_step()
        n_cond = self._max_frames - self._step_size
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


class AdaptiveInferenceStrategyBase(InferenceStrategyBase):
    """:137-211 -- per batch item, the observed frames of a window are chosen from the finished frames by
    farthest-point selection: start from the `always_selected` ones, then repeatedly take the candidate whose smallest
    squared distance to the frames chosen so far is largest.  `set_videos(samples)` hands over the current state of the
    videos before every step (scripts/video_sample.py:94-95); indices come back as one list per batch item."""

    def __init__(self, distance, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.distance = distance

    def set_videos(self, videos):
        self.videos = videos

    def embed(self, indices):
        import torch
        if self.distance == "l2":
            embs = [self.videos[:, i] for i in indices]
        elif self.distance == "lpips":
            if _lpips_embedder is None:
                raise NotImplementedError("distance='lpips' needs the pretrained LPIPS network (not available offline): "
                                          "inference_util.set_lpips_embedder(fn), or use distance='l2'")
            embs = [_lpips_embedder(self.videos[:, i]) for i in indices]
        else:
            raise NotImplementedError
        return torch.stack(embs, dim=1)

    def select_obs_indices(self, possible_next_indices, n, always_selected=(0,)):
        embs = self.embed(possible_next_indices)
        picked_per_item = []
        for b in range(len(self.videos)):
            nearest = [np.inf] * len(possible_next_indices)      # squared distance to the closest frame picked so far
            newest = always_selected[0]
            picked = [possible_next_indices[newest]]
            for i in range(1, n):
                for f in range(len(nearest)):
                    d = ((embs[b, newest] - embs[b][f]) ** 2).sum().cpu().item()
                    nearest[f] = min(nearest[f], d)
                newest = always_selected[i] if i < len(always_selected) else int(np.argmax(nearest))
                picked.append(possible_next_indices[newest])
            picked_per_item.append(picked)
        return picked_per_item

    def __next__(self):
        B = len(self.videos)
        if self._num_obs == 0 and self._current_step == 0:
            obs, latent = InferenceStrategyBase.__next__(self)           # one unconditional window for everybody
            return [obs for _ in range(B)], [latent for _ in range(B)]
        if self.is_done():
            raise StopIteration
        obs, latent = self.next_indices()
        assert isinstance(obs, list) and isinstance(latent, list)
        for idx in np.array(obs).flatten():
            assert idx in self._done_frames, (
                f"Attempting to condition on frame {idx} while it is not generated yet.\n"
                f"Generated frames: {self._done_frames}\nObserving: {obs}\nGenerating: {latent}")
        assert np.all(np.array(latent) < self._video_length)
        self._done_frames.update([idx for idx in latent if idx not in self._done_frames])
        self._current_step += 1
        return obs, [latent] * len(obs)


class AdaptiveAutoregressive(AdaptiveInferenceStrategyBase):
    """:214-229 -- the next `step_size` frames, conditioned on the newest finished frame plus the farthest-point picks."""

    def next_indices(self):
        if not self._done_frames:
            return [[]] * len(self.videos), list(range(self._max_frames))
        latent = self._next_chunk(max(self._done_frames) + 1, self._step_size)
        return self.select_obs_indices(sorted(self._done_frames)[::-1], self._max_frames - self._step_size), latent


class AdaptiveHierarchyNLevel(AdaptiveInferenceStrategyBase, HierarchyNLevel):
    """:421-519 -- the hierarchy's latent grid; always observed: the finished frames between the latents, the two
    nearest finished frames before them and the nearest one after; the rest of the budget by farthest-point selection."""

    def next_indices(self):
        L, done = self._video_length, self._done_frames
        latent, obs, room = self._latents_and_between()
        if obs is None:
            return [], latent
        if room < 2:
            return self._with_smaller_step()

        def back_from(i):
            while i not in done:
                i -= 1
                if i < 0:      # the reference loops forever here (no finished frame on that side, e.g. num_obs = 0)
                    raise RuntimeError("adaptive hierarchy: no finished frame before the latents to condition on")
            return i
        first = back_from(min(latent))
        obs.append(first)
        obs.append(back_from(first - 1))
        i = max(latent)
        while i not in done and i < L:
            i += 1
        if i < L:
            obs.append(i)
        candidates = list(done)
        obs = self.select_obs_indices(candidates, self._max_frames - self._step_size,
                                      always_selected=[candidates.index(i) for i in obs])
        self.last_sampled_idx = max(latent)
        return obs, latent


def get_adaptive_hierarchy_n_level(n):
    return type(f"AdaptiveHierarchy{n}Level", (AdaptiveHierarchyNLevel,), {"N": n})


GOAL_FRAMES = 5      # the goal-directed strategies treat the last five frames of the video as given (:536-539, :566-569)


class _GoalFramesMixin:
    """Marks the last `n` frames as observed / finished at construction (:536-539, :566-569, :616-619)."""

    def _give_goal_frames(self, n):
        for f in range(self._video_length - n, self._video_length):
            self._obs_frames.append(f)
            self._done_frames.add(f)

    def _take_goal_frames(self, n):
        for f in range(self._video_length - n, self._video_length):
            self._obs_frames.remove(f)
            self._done_frames.remove(f)


class GoalDirectedHierarchyNLevel(_GoalFramesMixin, HierarchyNLevel):
    """:534-555 -- the hierarchy schedule of a video that ends five frames early, every window also conditioning on the
    five goal frames (the window budget shrinks by five for the inner schedule)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._give_goal_frames(GOAL_FRAMES)

    def next_indices(self):
        self._take_goal_frames(GOAL_FRAMES)
        self._video_length -= GOAL_FRAMES
        self._max_frames -= GOAL_FRAMES
        try:
            obs, latent = super().next_indices()
            obs = obs + list(range(self._video_length, self._video_length + GOAL_FRAMES))
        finally:
            self._video_length += GOAL_FRAMES
            self._max_frames += GOAL_FRAMES
            self._give_goal_frames(GOAL_FRAMES)
        return obs, latent


def get_goal_directed_hierarchy_n_level(n):
    return type(f"GoalDirectedHierarchy{n}Level", (GoalDirectedHierarchyNLevel,), {"N": n})


class GoalDirectedAutoreg(_GoalFramesMixin, InferenceStrategyBase):
    """:565-582 -- autoregressive from the left with the five goal frames among the newest finished frames; the next
    latents start at the first unfinished frame and stop short of the last frame."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._give_goal_frames(GOAL_FRAMES)

    def next_indices(self):
        obs = sorted(self._done_frames)[-(self._max_frames - self._step_size):]
        first = next(i for i in range(self._video_length + 1) if i not in self._done_frames)
        return obs, list(range(first, min(first + self._step_size, self._video_length - 1)))


class GoalDirectedMixed(_GoalFramesMixin, InferenceStrategyBase):
    """:615-636 -- one goal frame (the last); half of the conditioning budget on the newest finished frames, the rest
    filled from the observed frames, latest first."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._give_goal_frames(1)

    def next_indices(self):
        budget = self._max_frames - self._step_size
        chosen = set(sorted(self._done_frames)[-(budget // 2):])
        for f in sorted(self._obs_frames, reverse=True):
            chosen.add(f)
            if len(chosen) == budget:
                break
        first = sorted(self._done_frames)[-2] + 1                   # just after the newest non-goal frame
        return sorted(chosen), self._next_chunk(first, self._step_size)


class BabyCondHoEtAlForVis(InferenceStrategyBase):
    """:585-593 -- a fixed 7-window illustration schedule (bypasses the bookkeeping, as the reference does)."""

    _WINDOWS = (([3, 5, 7, 9], [11, 13, 15]), ([9, 11, 13, 15], [17, 19, 21]), ([15, 17, 19, 21], [23, 25, 27]),
                ([9, 11, 13, 15], [10, 12, 14]), ([15, 17, 19, 21], [16, 18, 20]), ([21, 23, 25, 27], [22, 24, 26]),
                ([23, 24, 25, 26, 27], [28, 29]))

    def __iter__(self):
        return iter([(list(o), list(l)) for o, l in self._WINDOWS])


class HoEtAlForVis(InferenceStrategyBase):
    """:596-612 -- 16 frames spread over 0..60 first, then 9-frame windows [start-1, start+8) around the first
    unfinished frame, conditioning on whatever in them is finished."""

    def next_indices(self):
        if not self._done_frames:
            return [], [int(i) for i in np.linspace(0, 60, 16) if i < self._video_length]
        start = min(i for i in range(64) if i not in self._done_frames)
        window = range(start - 1, start + 8)
        obs = [i for i in window if i in self._done_frames]
        latent = [i for i in window if i not in self._done_frames]
        if 64 in latent:
            latent.remove(64)
            obs.append(55)
        return obs, latent


class GoogleFS4(InferenceStrategyBase):
    """:639-665 -- the frameskip-4 model of the two-model schedule: 16-frame windows on every 4th frame."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self._max_frames == 16, f"For GoogleFS4 strategy, max_frames must be 16, but got {self._max_frames}"

    def next_indices(self):
        newest = max(self._done_frames)
        obs = sorted(int(newest - 4 * i) for i in range(self._max_frames - self._step_size))
        first = max(obs) + 4
        latent = list(range(first, min(first + 4 * self._step_size, self._video_length), 4))
        while len(obs) + len(latent) < self._max_frames or min(obs) // 4 == 0:
            obs = [min(obs) - 4] + obs                              # pad to a full window with earlier every-4th frames
        return obs, latent

    def is_done(self):
        return self._video_length - max(self._done_frames) <= 4


class GoogleFS1(InferenceStrategyBase):
    """:668-706 -- the frameskip-1 model: fills the gaps the frameskip-4 pass left, 9-frame windows."""

    def __init__(self, done_frames, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self._max_frames == 9, "For GoogleFS1, max_frames should be 9."
        assert self._step_size == 6, ("For GoogleFS1, step_size should be 6, meaning that 6 frames will be generated in "
                                      "each inference step.")
        done = sorted(done_frames)
        fs4 = np.array(done[done.index(self._num_obs - 1):])
        assert np.all(fs4 % 4 == fs4[0] % 4), (
            "done_frames should come from a GoogleFS4 model and should only include frames that, starting from the last "
            f"observed frame, are 4 frames apart. Received {done}")
        assert max(done) + 4 >= self._video_length, (
            "done_frames should come from a GoogleFS4 model and should cover the entire video. But the last done_frame "
            f"is {max(done)}")
        self._done_frames = set(done)
        self._obs_frames = list(self._done_frames)

    def next_indices(self):
        first = self._num_obs - 1 + 8 * self._current_step
        obs = list(range(first, min(first + 9, self._video_length), 4))
        latent = list(range(obs[0] + 1, min(obs[0] + 8, self._video_length)))
        if len(obs) >= 2:
            latent.remove(obs[1])
        assert not set(obs) & set(latent)
        while len(obs) + len(latent) < 9:
            obs += [min(min(latent), min(obs)) - 1]
        return obs, latent


class Google(InferenceStrategyBase):
    """:709-736 -- frameskip-4 pass over the whole video, then the frameskip-1 pass in between (both forced to the
    window sizes of the two published models: 16/8 and 9/6)."""

    def __init__(self, video_length, num_obs, **ignored):
        super().__init__(video_length=video_length, num_obs=num_obs, max_frames=16, step_size=8)
        self.base_schedule = GoogleFS4(video_length=self._video_length, num_obs=self._num_obs,
                                       max_frames=self._max_frames, step_size=self._step_size)
        self._stage = "fs4"

    def next_indices(self):
        if self._stage == "fs4" and self.base_schedule.is_done():
            self.base_schedule = GoogleFS1(video_length=self._video_length, num_obs=self._num_obs, max_frames=9,
                                           step_size=6, done_frames=self.base_schedule._done_frames)
            self._stage = "fs1"
        return next(self.base_schedule)


class LikeGoogle(InferenceStrategyBase):
    """:739-776 -- one model playing both roles: first every 4th frame (in phase with the last observed frame),
    conditioning on the newest such frames; then runs of three latents between consecutive finished frames."""

    def next_indices(self):
        every4 = list(range((len(self._obs_frames) - 1) % 4, self._video_length, 4))
        todo = [i for i in every4 if i not in self._done_frames]
        if todo:
            latent = sorted(todo)[:self._step_size]
            n_cond = self._max_frames - len(latent)
            return sorted(i for i in every4 if i in self._done_frames)[-n_cond:], latent
        first = next(i for i in range(self._video_length) if i not in self._done_frames)
        obs, latent = [first - 1], []
        while len(obs) + len(latent) + 4 < self._max_frames and max(obs + latent) < self._video_length - 1:
            nxt = max(obs) + 1
            latent.extend(i for i in range(nxt, nxt + 3) if i < self._video_length)
            after = max(latent) + 1
            if after < self._video_length:
                obs.append(after)
        return obs, latent


# inference_util.py:779-799 (every strategy that needs no perceptual network; the adaptive-* ones need LPIPS)
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
    "goal-directed-autoreg": GoalDirectedAutoreg,
    "goal-directed-mixed": GoalDirectedMixed,
    "goal-directed-hierarchy-2": get_goal_directed_hierarchy_n_level(2),
    "ho-et-al-for-vis": HoEtAlForVis,
    "baby-cond-ho-et-al-for-vis": BabyCondHoEtAlForVis,
    "google": Google,
    "like-google": LikeGoogle,
    "adaptive-autoreg": AdaptiveAutoregressive,
    "adaptive-hierarchy-2": get_adaptive_hierarchy_n_level(2),
    "adaptive-hierarchy-3": get_adaptive_hierarchy_n_level(3),
}
