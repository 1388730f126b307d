"""Host-side mirror of the reference's diffusion process for the SAMPLING path.

Same names, arguments, return values and error behaviour as
`improved_diffusion/gaussian_diffusion.py` (reference file:line cited per
method); the float64 schedule tables are built here with numpy exactly as the
reference does and uploaded once (as their float32 casts, which is what
`_extract_into_tensor`, gaussian_diffusion.py:1019-1031, feeds the arithmetic),
and every tensor operation of a step runs in the HIP engine behind the C ABI.
The NLL path (`p_mean_variance`, `_vb_terms_bpd`, `_prior_bpd`, `calc_bpd_loop_subsampled`) runs on
the engine as well: one UNet forward per timestep plus one fused likelihood kernel (csrc/misc.hip: vb_terms_kernel
restates losses.py's normal_kl / discretized_gaussian_log_likelihood).  Gradient guidance (`use_gradient_method`,
gaussian_diffusion.py:264-271,350-364) runs on the engine too: vd_guided_step = a taped forward, the loss gradient and
a backward-data pass of the whole UNet w.r.t. its input (csrc/backward.hip); `denoised_fn` and `return_attn_weights`
are served as well.  Training losses are out of scope.
"""
import enum
import math

import numpy as np
import torch as th

from . import _lib


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """gaussian_diffusion.py:20-52."""
    if schedule_name in ("linear", "noisier_linear"):
        scale = 1000 / num_diffusion_timesteps
        end = 0.02 if schedule_name == "linear" else 0.025
        return np.linspace(scale * 0.0001, scale * end, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """gaussian_diffusion.py:55-74."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


_OBS_MODES = {"x_0": 0, "x_t": 1, "x_t_minus_1": 2}


def _f32(t, device):
    return t.to(device=device, dtype=th.float32).contiguous()


class GaussianDiffusion:
    """gaussian_diffusion.py:107-172 (tables) + the sampling methods."""

    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas = alphas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        if model_mean_type not in (ModelMeanType.EPSILON, ModelMeanType.START_X):
            # (create_gaussian_diffusion only ever builds EPSILON or START_X, script_util.py:429-431)
            raise NotImplementedError("ModelMeanType.PREVIOUS_X: not reachable from the reference's factory, not built")
        # LEARNED / LEARNED_RANGE (learn_sigma=True) construct fine, as in the reference, and fail at the first step the way
        # the reference does: see _refuse_learned

    # -- schedule upload -------------------------------------------------------------------------
    def _model_log_variance(self):
        """gaussian_diffusion.py:299-317."""
        if self.model_var_type == ModelVarType.FIXED_LARGE:
            return np.log(np.append(self.posterior_variance[1], self.betas[1:]))
        return self.posterior_log_variance_clipped          # FIXED_SMALL; a learned variance never reads this row

    def _refuse_learned(self, x):
        """gaussian_diffusion.py:277-285 on a video tensor: `B, C = x.shape[:2]` takes C = T, and the reference asserts
        `model_output.shape == (B, C * 2, *x.shape[2:])` against the network's (B, T, 6, H, W): learn_sigma cannot sample in
        the reference either.  Same exception here."""
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            B, C = x.shape[:2]
            raise AssertionError(f"model_output.shape == {(B, C * 2, *x.shape[2:])} (gaussian_diffusion.py:283): a learned variance "
                                 f"is split along dim 1, which is T for video tensors; the reference fails here as well")

    def _device_tables(self):
        rows = [self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1,
                self.posterior_mean_coef2, self._model_log_variance(), self.alphas_cumprod,
                self.alphas_cumprod_prev, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod,
                self.posterior_log_variance_clipped, self.log_one_minus_alphas_cumprod, 1.0 - self.betas]
        return np.ascontiguousarray(np.stack(rows).astype(np.float32))

    def _timestep_map_and_scale(self):
        scale = 1000.0 / self.num_timesteps if self.rescale_timesteps else 1.0
        return list(range(self.num_timesteps)), scale

    def _bind(self, model):
        """Upload this process' tables into the model's engine (once per (diffusion, model) pair)."""
        model = getattr(model, "model", model)            # accept a _WrappedModel
        if getattr(model, "_bound_schedule", None) is not self:
            tab = self._device_tables()
            tmap, scale = self._timestep_map_and_scale()
            tm = np.ascontiguousarray(np.array(tmap, dtype=np.int32))
            _lib.check(_lib.lib().vd_set_schedule(model._handle, self.num_timesteps, _lib.ptr(tab), _lib.ptr(tm),
                                                  float(np.float32(scale))))
            _lib.check(_lib.lib().vd_set_model_mean_type(model._handle, 1 if self.model_mean_type == ModelMeanType.START_X else 0))
            model._bound_schedule = self
        return model

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    # -- per-step entry points ---------------------------------------------------------------------
    def _step(self, mode, model, x, t, clip_denoised, denoised_fn, model_kwargs, eta, noise,
              return_attn_weights=False, use_gradient_method=False):
        if model_kwargs is None:
            model_kwargs = {}
        self._refuse_learned(x)
        if return_attn_weights and use_gradient_method:
            raise NotImplementedError("return_attn_weights together with use_gradient_method")
        if use_gradient_method:
            if mode != 0 or denoised_fn is not None:
                raise NotImplementedError("use_gradient_method: p_sample without denoised_fn (the reference's ddim_sample "
                                          "has no such option, gaussian_diffusion.py:597-634)")
            out = self._guided(model, x, t, clip_denoised, model_kwargs, noise2=noise, want_sample=True)
            return out["sample"], out["pred_xstart"]
        if denoised_fn is not None:
            return self._step_denoised_fn(mode, model, x, t, clip_denoised, denoised_fn, model_kwargs, eta, noise, return_attn_weights)
        model = self._bind(model)
        B = x.shape[0]
        assert t.shape == (B,)                                   # gaussian_diffusion.py:273
        if t.device.type == "cpu" and B and (int(t.min()) < 0 or int(t.max()) >= self.num_timesteps):
            raise IndexError(f"index {int(t.max())} is out of bounds for dimension 0 with size {self.num_timesteps}")
        # (a device-resident t is range-checked by the kernels: NaN output + model.check_device_errors())
        dev = model.device
        xs = _f32(x, dev)
        kw = model._pack_kwargs(xs, model_kwargs)
        if noise is None:
            noise = th.randn_like(xs)                            # gaussian_diffusion.py:438 / :628 (drawn even for eta=0)
        else:
            noise = _f32(noise, dev)
        assert noise.shape == xs.shape
        tt = t.to(device=dev, dtype=th.int64).contiguous()
        sample = th.empty_like(xs)
        xstart = th.empty_like(xs)
        T = xs.shape[1]
        L = _lib.lib()
        common = (model._handle, B, T, _lib.ptr(xs), _lib.ptr(kw["obs_src"]), _lib.ptr(kw["obs_mask"]),
                  _lib.ptr(kw["latent_mask"]), _lib.ptr(kw["kinda_marg_mask"]), _lib.ptr(kw["frame_indices"]),
                  _lib.ptr(tt), kw["obs_mode"], 1 if clip_denoised else 0)
        self._last_attn = model._attn_capture(B, T) if return_attn_weights else None     # unet.py:457-466 per block
        try:
            if mode == 0:
                rc = L.vd_p_sample(*common, _lib.ptr(noise), 0, 0, _lib.ptr(sample), _lib.ptr(xstart), None,
                                   _lib.current_stream())
            else:
                rc = L.vd_ddim_sample(*common, float(eta), _lib.ptr(noise), 0, 0, _lib.ptr(sample), _lib.ptr(xstart), None,
                                      _lib.current_stream())
        finally:
            if return_attn_weights:
                model._attn_release()
        _lib.check(rc)
        return sample, xstart

    def _guided(self, model, x, t, clip_denoised, model_kwargs, noise2=None, want_sample=False, _noise=None):
        """p_mean_variance(..., use_gradient_method=True) (+ p_sample's noise add) on the engine: one taped forward, the
        loss gradient, one backward-data pass (gaussian_diffusion.py:264-271,350-364).  Draw order as in the reference:
        the noise of the x_{t-1} sample inside p_mean_variance first, p_sample's own noise second."""
        if self.model_mean_type != ModelMeanType.EPSILON:
            raise NotImplementedError("use_gradient_method with predict_xstart=True")
        base = self._bind(model)
        base._require_guidance()
        dev = base.device
        xs = _f32(x, dev)
        B, T = xs.shape[:2]
        assert t.shape == (B,)
        if t.device.type == "cpu" and B and (int(t.min()) < 0 or int(t.max()) >= self.num_timesteps):
            raise IndexError(f"index {int(t.max())} is out of bounds for dimension 0 with size {self.num_timesteps}")
        kw = base._pack_kwargs(xs, dict(model_kwargs, observed_frames="x_t"))      # obs_src is unused: every frame is latent
        xtm1 = _f32(model_kwargs["x_t_minus_1"], dev)
        noise = th.randn_like(xs) if _noise is None else _f32(_noise, dev)         # gaussian_diffusion.py:351
        if want_sample:
            noise2 = th.randn_like(xs) if noise2 is None else _f32(noise2, dev)     # gaussian_diffusion.py:438
        tt = t.to(device=dev, dtype=th.int64).contiguous()
        mean, xstart, grad = th.empty_like(xs), th.empty_like(xs), th.empty_like(xs)
        sample = th.empty_like(xs) if want_sample else None
        _lib.check(_lib.lib().vd_guided_step(
            base._handle, B, T, _lib.ptr(xs), _lib.ptr(kw["obs_mask"]), _lib.ptr(kw["latent_mask"]), _lib.ptr(kw["kinda_marg_mask"]),
            _lib.ptr(kw["frame_indices"]), _lib.ptr(tt), 1 if clip_denoised else 0, _lib.ptr(xtm1), _lib.ptr(noise),
            _lib.ptr(noise2) if want_sample else None, _lib.ptr(mean), _lib.ptr(xstart), _lib.ptr(grad),
            _lib.ptr(sample) if want_sample else None, _lib.current_stream()))
        return {"mean": mean, "pred_xstart": xstart, "grad": grad, "sample": sample}

    def _step_denoised_fn(self, mode, model, x, t, clip_denoised, denoised_fn, model_kwargs, eta, noise, return_attn_weights=False):
        """process_xstart with a caller's function (gaussian_diffusion.py:319-324): `denoised_fn` sees the UNCLIPPED x_0
        prediction, the clamp and the posterior run on what it returns.  Two launches around a host callback instead of
        the fused step: forward + x_0 (vd_p_mean_variance, clip off), then vd_posterior_from_xstart."""
        out = self.p_mean_variance(model, x, t, clip_denoised=False, model_kwargs=model_kwargs, return_attn_weights=return_attn_weights)
        self._last_attn = out["attn"]                              # the maps of the one forward this step makes (gaussian_diffusion.py:274-324)
        base = self._bind(model)
        dev = base.device
        xs = _f32(x, dev)
        B = xs.shape[0]
        x0 = _f32(denoised_fn(out["pred_xstart"]), dev)
        assert x0.shape == xs.shape
        noise = th.randn_like(xs) if noise is None else _f32(noise, dev)
        tt = t.to(device=dev, dtype=th.int64).contiguous()
        sample, xstart = th.empty_like(xs), th.empty_like(xs)
        _lib.check(_lib.lib().vd_posterior_from_xstart(
            base._handle, mode, B, xs[0].numel(), _lib.ptr(xs), _lib.ptr(x0), _lib.ptr(tt), 1 if clip_denoised else 0,
            float(eta), _lib.ptr(noise), 0, 0, _lib.ptr(sample), _lib.ptr(xstart), None, _lib.current_stream()))
        return sample, xstart

    def _extract(self, arr, t, shape):
        """_extract_into_tensor (gaussian_diffusion.py:1019-1031): float64 table gathered at t, cast to float32,
        broadcast to `shape` (tensor plumbing: a gather and a view)."""
        res = th.from_numpy(np.asarray(arr)).to(device=t.device)[t].float()
        while len(res.shape) < len(shape):
            res = res[..., None]
        return res.expand(shape)

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None,
                        return_attn_weights=False, use_gradient_method=False):
        """gaussian_diffusion.py:229-372 -> {'mean', 'variance', 'log_variance', 'pred_xstart', 'attn'} (+ 'eps', the raw
        model output, which the NLL loop reuses)."""
        self._refuse_learned(x)
        if return_attn_weights and use_gradient_method:
            raise NotImplementedError("return_attn_weights together with use_gradient_method")
        if use_gradient_method:
            if denoised_fn is not None:
                raise NotImplementedError("use_gradient_method with denoised_fn")
            g = self._guided(model, x, t, clip_denoised, model_kwargs or {}, _noise=getattr(self, "_guidance_noise", None))
            tt = t.to(device=g["mean"].device, dtype=th.int64)
            variance = self.posterior_variance if self.model_var_type == ModelVarType.FIXED_SMALL \
                else np.append(self.posterior_variance[1], self.betas[1:])
            return {"mean": g["mean"], "variance": self._extract(variance, tt, g["mean"].shape),
                    "log_variance": self._extract(self._model_log_variance(), tt, g["mean"].shape),
                    "pred_xstart": g["pred_xstart"], "attn": None, "grad": g["grad"]}
        if denoised_fn is not None:
            out = self.p_mean_variance(model, x, t, clip_denoised=False, model_kwargs=model_kwargs, return_attn_weights=return_attn_weights)
            base = self._bind(model)
            xs = _f32(x, base.device)
            x0 = _f32(denoised_fn(out["pred_xstart"]), base.device)
            tt = t.to(device=base.device, dtype=th.int64).contiguous()
            mean, xstart = th.empty_like(xs), th.empty_like(xs)
            _lib.check(_lib.lib().vd_posterior_from_xstart(
                base._handle, 0, xs.shape[0], xs[0].numel(), _lib.ptr(xs), _lib.ptr(x0), _lib.ptr(tt),
                1 if clip_denoised else 0, 0.0, None, 0, 0, None, _lib.ptr(xstart), _lib.ptr(mean), _lib.current_stream()))
            out.update(mean=mean, pred_xstart=xstart)
            return out
        model = self._bind(model)
        B = x.shape[0]
        assert t.shape == (B,)
        if t.device.type == "cpu" and B and (int(t.min()) < 0 or int(t.max()) >= self.num_timesteps):
            raise IndexError(f"index {int(t.max())} is out of bounds for dimension 0 with size {self.num_timesteps}")
        dev = model.device
        xs = _f32(x, dev)
        kw = model._pack_kwargs(xs, model_kwargs or {})
        tt = t.to(device=dev, dtype=th.int64).contiguous()
        mean, xstart, eps = th.empty_like(xs), th.empty_like(xs), th.empty_like(xs)
        attn = model._attn_capture(B, xs.shape[1]) if return_attn_weights else None
        try:
            rc = _lib.lib().vd_p_mean_variance(
                model._handle, B, xs.shape[1], _lib.ptr(xs), _lib.ptr(kw["obs_src"]), _lib.ptr(kw["obs_mask"]),
                _lib.ptr(kw["latent_mask"]), _lib.ptr(kw["kinda_marg_mask"]), _lib.ptr(kw["frame_indices"]), _lib.ptr(tt),
                kw["obs_mode"], 1 if clip_denoised else 0, _lib.ptr(mean), _lib.ptr(xstart), _lib.ptr(eps), _lib.current_stream())
        finally:
            if attn is not None:
                model._attn_release()
        _lib.check(rc)
        logvar = self._model_log_variance()
        variance = self.posterior_variance if self.model_var_type == ModelVarType.FIXED_SMALL \
            else np.append(self.posterior_variance[1], self.betas[1:])             # gaussian_diffusion.py:299-317
        return {"mean": mean, "variance": self._extract(variance, tt, xs.shape),
                "log_variance": self._extract(logvar, tt, xs.shape), "pred_xstart": xstart, "attn": attn, "eps": eps}

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """gaussian_diffusion.py:208-227 (host composition of schedule rows; the samplers fuse it in posterior_kernel)."""
        assert x_start.shape == x_t.shape
        mean = self._extract(self.posterior_mean_coef1, t, x_t.shape) * x_start + self._extract(self.posterior_mean_coef2, t, x_t.shape) * x_t
        return mean, self._extract(self.posterior_variance, t, x_t.shape), self._extract(self.posterior_log_variance_clipped, t, x_t.shape)

    # -- NLL path (scripts/video_nll.py) -------------------------------------------------------------
    def _nll_mask(self, latent_mask, B, T, dev):
        if latent_mask is None:
            return None
        m = _f32(latent_mask, dev).reshape(B, -1)
        assert m.shape[1] == T, "latent_mask is per frame: (B, T, 1, 1, 1)"
        return m.contiguous()

    def _vb_terms_bpd(self, model, x_start, x_t, t, clip_denoised=True, model_kwargs=None, latent_mask=None, _noise=None):
        """gaussian_diffusion.py:750-790 -> {'output': [N] bits/dim, 'pred_xstart'} (+ the two MSEs of
        calc_bpd_loop_subsampled when the noise x_t was drawn with is passed)."""
        model = self._bind(model)          # (predict_xstart=True: the engine takes the network output handed to vd_vb_terms as the x_0 prediction)
        dev = model.device
        xs, xt = _f32(x_start, dev), _f32(x_t, dev)
        B, T = xs.shape[:2]
        out = self.p_mean_variance(model, xt, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
        tt = t.to(device=dev, dtype=th.int64).contiguous()
        m = self._nll_mask(latent_mask, B, T, dev)
        nz = _f32(_noise, dev) if _noise is not None else None
        vb, xmse, mse = (th.empty(B, device=dev, dtype=th.float32) for _ in range(3))
        _lib.check(_lib.lib().vd_vb_terms(model._handle, B, T, _lib.ptr(xs), _lib.ptr(xt), _lib.ptr(out["eps"]), _lib.ptr(nz),
                                          _lib.ptr(tt), 1 if clip_denoised else 0, _lib.ptr(m), _lib.ptr(vb), _lib.ptr(xmse),
                                          _lib.ptr(mse) if nz is not None else None, None, _lib.current_stream()))
        res = {"output": vb, "pred_xstart": out["pred_xstart"], "xstart_mse": xmse}
        if nz is not None:
            res["mse"] = mse
        return res

    def _prior_bpd(self, x_start, latent_mask=None, model=None):
        """gaussian_diffusion.py:909-926."""
        model = self._bind(model if model is not None else self._last_model())
        dev = model.device
        xs = _f32(x_start, dev)
        B, T = xs.shape[:2]
        m = self._nll_mask(latent_mask, B, T, dev)
        out = th.empty(B, device=dev, dtype=th.float32)
        _lib.check(_lib.lib().vd_prior_bpd(model._handle, B, T, _lib.ptr(xs), _lib.ptr(m), _lib.ptr(out), _lib.current_stream()))
        return out

    def calc_bpd_loop_subsampled(self, model, x_start, clip_denoised=True, model_kwargs=None, latent_mask=None, t_seq=None):
        """gaussian_diffusion.py:928-1002 -> {'total_bpd','prior_bpd','vb','xstart_mse','mse'}; t_seq may be a list of
        timesteps or a 2-D array with one row of timesteps per batch item."""
        base = self._bind(model)
        self._model_hint = base
        dev = base.device
        xs = _f32(x_start, dev)
        B = xs.shape[0]
        if t_seq is None:
            t_seq = list(range(self.num_timesteps))[::-1]
        two_d = isinstance(t_seq, np.ndarray) and t_seq.ndim == 2
        if two_d:
            t_seq = t_seq.transpose()
        vb, xstart_mse, mse = [], [], []
        for t in t_seq:
            t_batch = th.tensor(t, device=dev) if two_d else th.tensor([t] * B, device=dev)
            noise = th.randn_like(xs)
            x_t = self.q_sample(xs, t_batch, noise=noise, model=base)
            out = self._vb_terms_bpd(model, x_start=xs, x_t=x_t, t=t_batch, clip_denoised=clip_denoised,
                                     model_kwargs=model_kwargs, latent_mask=latent_mask, _noise=noise)
            vb.append(out["output"])
            xstart_mse.append(out["xstart_mse"])
            mse.append(out["mse"])
        vb, xstart_mse, mse = th.stack(vb, dim=1), th.stack(xstart_mse, dim=1), th.stack(mse, dim=1)
        prior_bpd = self._prior_bpd(xs, latent_mask=latent_mask, model=base)
        return {"total_bpd": vb.sum(dim=1) + prior_bpd, "prior_bpd": prior_bpd, "vb": vb, "xstart_mse": xstart_mse, "mse": mse}

    def calc_bpd_loop(self, model, x_start, clip_denoised=True, model_kwargs=None, latent_mask=None):
        """gaussian_diffusion.py:1004-1016."""
        return self.calc_bpd_loop_subsampled(model, x_start, clip_denoised=clip_denoised, model_kwargs=model_kwargs,
                                             latent_mask=latent_mask)

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None,
                 return_attn_weights=False, use_gradient_method=False):
        """gaussian_diffusion.py:403-448.  `x` is not modified; returns fresh tensors."""
        sample, xstart = self._step(0, model, x, t, clip_denoised, denoised_fn, model_kwargs, 0.0, None,
                                    return_attn_weights, use_gradient_method)
        return {"sample": sample, "pred_xstart": xstart, "attn": self._last_attn if return_attn_weights else None}

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, eta=0.0):
        """gaussian_diffusion.py:597-634."""
        sample, xstart = self._step(1, model, x, t, clip_denoised, denoised_fn, model_kwargs, eta, None)
        return {"sample": sample, "pred_xstart": xstart}

    def q_sample(self, x_start, t, noise=None, model=None):
        """gaussian_diffusion.py:190-206.  Needs an engine for its tables: pass `model` (or call after
        any p_sample on the same diffusion object)."""
        model = self._bind(model if model is not None else self._last_model())
        dev = model.device
        xs = _f32(x_start, dev)
        if noise is None:
            noise = th.randn_like(xs)
        noise = _f32(noise, dev)
        assert noise.shape == xs.shape
        B = xs.shape[0]
        tt = t.to(device=dev, dtype=th.int64).reshape(-1)
        if tt.numel() == 1 and B != 1:
            tt = tt.expand(B)
        tt = tt.contiguous()
        out = th.empty_like(xs)
        _lib.check(_lib.lib().vd_q_sample(model._handle, B, xs[0].numel(), _lib.ptr(xs), _lib.ptr(tt), _lib.ptr(noise),
                                          _lib.ptr(out), _lib.current_stream()))
        return out

    def _last_model(self):
        m = getattr(self, "_model_hint", None)
        if m is None:
            raise RuntimeError("q_sample needs the model whose engine holds the schedule tables: pass model=")
        return m

    # -- loops -------------------------------------------------------------------------------------
    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, model_kwargs=None,
                      latent_mask=None, device=None, progress=False, return_attn_weights=False,
                      use_gradient_method=False):
        """gaussian_diffusion.py:450-526: returns (sample, attns).  With return_attn_weights, attns holds one running mean
        per (quartile of the schedule, attention type): 'attn/q<k>-temporal' / 'attn/q<k>-spatial' (:496-524) -- each
        block's head-averaged weights averaged over the non-attended axis, spatial maps resized (nearest) to the first
        block's size and renormalised to keep their mean, every step weighted 1 / (num_timesteps / 4)."""
        final, attns = None, {}
        steps = self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                               model_kwargs=model_kwargs, latent_mask=latent_mask, device=device,
                                               progress=progress, return_attn_weights=return_attn_weights,
                                               use_gradient_method=use_gradient_method)
        for k, out in enumerate(steps):
            final = out
            if not return_attn_weights:
                continue
            quartile = (4 * (self.num_timesteps - k - 1)) // self.num_timesteps
            for kind, maps in out["attn"].items():
                if not maps:
                    continue
                tag = f"attn/q{quartile}-{kind}"
                target = maps[0][0].shape                              # the first block's map size (largest resolution)
                acc = attns.get(tag, 0)
                for m in maps:
                    per_item = m.view(shape[0], m.shape[0] // shape[0], *m.shape[1:]).mean(dim=1)
                    if "temporal" not in kind:
                        r = th.nn.functional.interpolate(per_item.unsqueeze(0), size=target, mode="nearest").squeeze(0)
                        per_item = r / r.mean() * per_item.mean()
                    acc = acc + per_item / (self.num_timesteps / 4)
                attns[tag] = acc
        getattr(model, "check_device_errors", lambda: None)()       # waits for the device (every stream): a non-finite network output / bad index of ANY step surfaces here, not in a later loop
        return final["sample"], attns

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                  model_kwargs=None, latent_mask=None, device=None, progress=False,
                                  return_attn_weights=False, use_gradient_method=False):
        """gaussian_diffusion.py:528-595, including the per-step side draws that consume the global RNG
        (x_t_minus_1, random_t, x_random) so a seeded run walks the generator like the reference."""
        base = getattr(model, "model", model)
        if device is None:
            device = base.device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else th.randn(*shape, device=device)
        self._model_hint = base
        for i in list(range(self.num_timesteps))[::-1]:
            t = th.tensor([i] * shape[0], device=device)
            if "hybrid" in model_kwargs["observed_frames"]:
                raise NotImplementedError("observed_frames='hybrid_<k>' is a training-time option")
            model_kwargs["x_t_minus_1"] = self.q_sample(
                model_kwargs["x0"], t - 1, noise=th.randn_like(_f32(model_kwargs["x0"], device)) if noise is None else noise,
                model=base)
            model_kwargs["random_t"] = th.floor(t * th.rand(t.shape).to(device)).long()   # CPU generator, as :569-570
            if noise is None:
                th.randn_like(_f32(model_kwargs["x0"], device))   # x_random's draw: consumed, never read in eval (unet.py:962)
            out = self.p_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                model_kwargs=model_kwargs, return_attn_weights=return_attn_weights,
                                use_gradient_method=use_gradient_method)
            yield out
            img = out["sample"]

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, model_kwargs=None,
                         latent_mask=None, device=None, progress=False, eta=0.0):
        """gaussian_diffusion.py:670-700: returns the sample only."""
        final = None
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                        denoised_fn=denoised_fn, model_kwargs=model_kwargs,
                                                        device=device, progress=progress, eta=eta):
            final = sample
        getattr(model, "check_device_errors", lambda: None)()
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     model_kwargs=None, latent_mask=None, device=None, progress=False, eta=0.0):
        """gaussian_diffusion.py:702-748."""
        base = getattr(model, "model", model)
        if device is None:
            device = base.device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else th.randn(*shape, device=device)
        for i in list(range(self.num_timesteps))[::-1]:
            t = th.tensor([i] * shape[0], device=device)
            out = self.ddim_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                   model_kwargs=model_kwargs, eta=eta)
            yield out
            img = out["sample"]
