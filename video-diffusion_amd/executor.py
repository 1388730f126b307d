"""Window executor: the step loop of scripts/video_sample.py:149-168 as graph replays.

    for timestep in reversed(range(diffusion.num_timesteps)):
        local = diffusion.p_sample(model, local, t, clip_denoised=True, model_kwargs=kw)['sample']

The reference (and the eager path here) drives that loop from the host: ~330 kernel launches per step.  The executor
keeps the loop state on the device (respaced index, Philox counter), holds ONE captured hipGraph per window signature
and replays it `num_timesteps` times (C ABI: vd_window_begin / vd_window_run, include/vd_amd.h; BASELINE configs[4]).
The window's tensors live in buffers owned by this object, one set per (B, T), so every window of that shape reuses the
same graph: CARLA's 47 windows need two graphs (Tw = 20 and Tw = 14).

Noise: the engine's counter-based generator (Philox4x32-10 + Box-Muller inside the posterior kernel) with a seed drawn
from torch's global generator, so `torch.manual_seed` still makes a run reproducible; the draws are N(0, 1) like the
reference's `th.randn_like`, not the same stream.  `observed_frames` in {'x_0', 'x_t', 'x_t_minus_1'}.  'x_t_minus_1' has two
callers in the reference and they differ: `p_sample_loop` re-noises the clean observed frames to t - 1 before each step
(gaussian_diffusion.py:565-568) -- `renoise=True`, the default here: q_sample inside the graph from the same Philox stream --
while scripts/video_sample.py:149-166 calls `p_sample` directly with `x_t_minus_1 = x0`, a clean placeholder that is read as it
is at every step -- `renoise=False`, what `infer_video` passes, so that its graph and eager executors sample the same
conditional distribution.
"""
import torch as th

from . import _lib

_OBS_MODES = {"x_0": 0, "x_t": 1, "x_t_minus_1": 2}


class WindowExecutor:
    def __init__(self, model, diffusion, prefix_cache=False, suffix_skip=False):
        """prefix_cache: compute the observed frames' activations before the first attention layer once per window instead
        of once per step ('x_0' mode, cond_emb_type='channel'; include/vd_amd.h: vd_set_window_prefix_cache).
        suffix_skip: run everything behind the last attention layer without the purely observed frames ('x_0' and
        'x_t_minus_1'; vd_set_window_suffix_skip): the other frames' samples are those of the full step, the observed frames'
        entries of the returned window are meaningless (infer_video keeps only the latent frames, video_sample.py:170-186)."""
        self.model = diffusion._bind(model)
        self.diffusion = diffusion
        self.prefix_cache = bool(prefix_cache)
        self.suffix_skip = bool(suffix_skip)
        self.stream = th.cuda.Stream(device=self.model.device)        # a capture needs a non-default stream
        self._bufs = {}
        self.x = None

    def _buffers(self, B, T):
        key = (B, T)
        if key not in self._bufs:
            dev, S = self.model.device, self.model.image_size
            f = lambda *s: th.zeros(*s, dtype=th.float32, device=dev)  # noqa: E731
            self._bufs[key] = dict(x=f(B, T, 3, S, S), obs_src=f(B, T, 3, S, S), obs_mask=f(B * T), latent_mask=f(B * T),
                                   kinda_marg_mask=f(B * T), frame_indices=th.zeros(B, T, dtype=th.int64, device=dev))
        return self._bufs[key]

    def begin(self, x_init, model_kwargs, t_start=None, seed=None, sampler="p_sample", eta=0.0, clip_denoised=True, renoise=True):
        """Arm a window: copy its tensors into the executor's buffers, set the device counters, capture if new.
        renoise ('x_t_minus_1' only): True = p_sample_loop's form, the clean model_kwargs['x0'] frames re-noised to t - 1 inside
        every step; False = model_kwargs['x_t_minus_1'] read as it is at every step (a direct p_sample caller)."""
        mode = model_kwargs.get("observed_frames", "x_0")
        if mode not in _OBS_MODES:
            raise NotImplementedError(f"observed_frames={mode!r}: the window executor handles 'x_0', 'x_t' and 'x_t_minus_1'")
        self.diffusion._refuse_learned(x_init)                        # a learned variance cannot sample (gaussian_diffusion.py:283)
        B, T = x_init.shape[:2]
        bufs = self._buffers(B, T)
        # the engine holds ONE schedule: another diffusion may have been bound to this model since the last window
        # (vd_set_schedule drops the captured graphs, so a stale replay is impossible)
        self.diffusion._bind(self.model)
        cur = th.cuda.current_stream(self.model.device)
        self.stream.wait_stream(cur)
        with th.cuda.stream(self.stream):
            kw = self.model._pack_kwargs(x_init, model_kwargs)
            # the copies below read the caller's tensors on self.stream: keep the caching allocator from handing their
            # memory out again on the caller's stream before these reads have run
            for v in [x_init] + [v for v in kw.values() if isinstance(v, th.Tensor)]:
                if v.is_cuda:
                    v.record_stream(self.stream)
            bufs["x"].copy_(x_init)
            for k in ("obs_mask", "latent_mask", "kinda_marg_mask", "frame_indices"):
                bufs[k].copy_(kw[k].view(bufs[k].shape))
            if mode == "x_0":
                bufs["obs_src"].copy_(kw["obs_src"])
            elif mode == "x_t_minus_1" and renoise:                    # the CLEAN frames: the graph draws q_sample(x0, t - 1) from them
                bufs["obs_src"].copy_(model_kwargs["x0"].to(device=bufs["obs_src"].device, dtype=th.float32))
            elif mode == "x_t_minus_1":                                # the caller's tensor, read as it is by every step
                bufs["obs_src"].copy_(kw["obs_src"])
            if seed is None:
                seed = int(th.randint(0, 2 ** 62, (1,)).item())        # torch.manual_seed governs the run
            if t_start is None:
                t_start = self.diffusion.num_timesteps - 1
            obs_src = bufs["x"] if mode == "x_t" else bufs["obs_src"]
            _lib.check(_lib.lib().vd_set_window_prefix_cache(self.model._handle, 1 if self.prefix_cache else 0))
            _lib.check(_lib.lib().vd_set_window_suffix_skip(self.model._handle, 1 if self.suffix_skip else 0))
            _lib.check(_lib.lib().vd_window_begin(
                self.model._handle, B, T, _lib.ptr(bufs["x"]), _lib.ptr(obs_src), _lib.ptr(bufs["obs_mask"]),
                _lib.ptr(bufs["latent_mask"]), _lib.ptr(bufs["kinda_marg_mask"]), _lib.ptr(bufs["frame_indices"]),
                3 if (mode == "x_t_minus_1" and not renoise) else _OBS_MODES[mode], 0 if sampler == "p_sample" else 1, 1 if clip_denoised else 0, float(eta), seed, 0,
                int(t_start), self.stream.cuda_stream))
        self.x = bufs["x"]
        self.seed = seed
        self._left = int(t_start) + 1
        self._gen = int(_lib.lib().vd_window_generation(self.model._handle))
        return self

    def run(self, n_steps=None):
        """Replay the step graph n_steps times (default: down to t = 0).  Returns the window tensor (updated in place)."""
        if n_steps is None:
            n_steps = self._left
        # the step counters are one set per engine: refuse to continue a window another executor has re-armed since
        if getattr(self, "_gen", None) != int(_lib.lib().vd_window_generation(self.model._handle)):
            raise RuntimeError("WindowExecutor.run: another window was begun on this model since this executor's begin(); "
                               "begin() again")
        _lib.check(_lib.lib().vd_window_run(self.model._handle, int(n_steps), self.stream.cuda_stream))
        self._left -= int(n_steps)
        th.cuda.current_stream(self.model.device).wait_stream(self.stream)
        return self.x

    def sample_window(self, x_init, model_kwargs, sampler="p_sample", eta=0.0, seed=None, renoise=True):
        """All num_timesteps steps of one window; returns a fresh tensor."""
        self.begin(x_init, model_kwargs, seed=seed, sampler=sampler, eta=eta, renoise=renoise)
        return self.run(self.diffusion.num_timesteps).clone()

    @property
    def cached_frames(self):
        """Frames of the armed window whose prefix activations come from the cache (0: cache off or nothing observed)."""
        return int(_lib.lib().vd_window_prefix_frames(self.model._handle))

    @property
    def suffix_frames(self):
        """Frames the armed window's suffix (everything behind the last attention layer) runs on (0: all of them)."""
        return int(_lib.lib().vd_window_suffix_frames(self.model._handle))

    @property
    def graphs(self):
        return int(_lib.lib().vd_window_graphs(self.model._handle))
