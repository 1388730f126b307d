"""ORACLE (test infrastructure, NOT product code) -- ancestral / DDIM sampling steps.

torch-CPU fp32 restatement of the reference's per-step posterior arithmetic
and its sampling loops.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s cpu_baseline leg may import it.

Pinned by tests/golden/psample_*.npz (imported reference, explicit noise).

Follows (reference file:line, relative to /root/reference/improved_diffusion):
  - _extract_into_tensor (f64 gather THEN cast to f32)  gaussian_diffusion.py:1019-1031
  - p_mean_variance (EPSILON, FIXED_*)                  gaussian_diffusion.py:229-372
  - _predict_xstart_from_eps / _predict_eps_from_xstart gaussian_diffusion.py:374-396
  - q_posterior_mean_variance                           gaussian_diffusion.py:208-227
  - p_sample                                            gaussian_diffusion.py:403-448
  - ddim_sample                                         gaussian_diffusion.py:597-634
  - q_sample                                            gaussian_diffusion.py:190-206
  - p_sample_loop_progressive (side draws)              gaussian_diffusion.py:528-595
  - ddim_sample_loop_progressive                        gaussian_diffusion.py:702-748
  - _WrappedModel.__call__                              respace.py:111-119
  - q_mean_variance / _prior_bpd                        gaussian_diffusion.py:174-188, 909-926
  - _vb_terms_bpd                                       gaussian_diffusion.py:750-790
  - calc_bpd_loop_subsampled                            gaussian_diffusion.py:928-1002

The loops draw from torch's global CPU generator in the reference's ORDER (p_sample_loop_progressive: the initial image,
then per step x_t_minus_1's noise, random_t's uniform, x_random's noise, p_sample's noise); tests/golden/loops_tiny.npz
holds what the imported reference produced from the same seeds.
"""
import math

import numpy as np
import torch

from .losses_ref import discretized_gaussian_log_likelihood, mean_flat, normal_kl


def _coef(table, t, like):
    v = torch.from_numpy(np.asarray(table))[t].float()
    return v.view(-1, *([1] * (like.dim() - 1)))


class SamplerRef:
    def __init__(self, sched, net):
        self.s, self.net = sched, net
        self.num_timesteps = sched.num_timesteps

    def eps(self, x, t, kw):
        tm = torch.tensor(self.s.timestep_map, dtype=t.dtype)[t]
        if self.s.rescale_timesteps:
            tm = tm.float() * (1000.0 / self.s.original_num_steps)
        return self.net(x, tm, **kw)

    def mean_variance(self, x, t, kw, clip=True, eps=None):
        s = self.s
        if eps is None:
            eps = self.eps(x, t, kw)
        x0 = _coef(s.sqrt_recip_alphas_cumprod, t, x) * x - _coef(s.sqrt_recipm1_alphas_cumprod, t, x) * eps
        if clip:
            x0 = x0.clamp(-1, 1)
        mean = _coef(s.posterior_mean_coef1, t, x) * x0 + _coef(s.posterior_mean_coef2, t, x) * x
        shape = x.shape
        return dict(mean=mean, pred_xstart=x0, eps=eps,
                    variance=_coef(s.model_variance, t, x).expand(shape),
                    log_variance=_coef(s.model_log_variance, t, x).expand(shape))

    def p_sample(self, x, t, kw, noise, clip=True, eps=None):
        o = self.mean_variance(x, t, kw, clip, eps)
        nz = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        o["sample"] = o["mean"] + nz * torch.exp(0.5 * o["log_variance"]) * noise
        return o

    def guided_mean_variance(self, x, t, kw, noise, clip=True):
        """p_mean_variance(..., use_gradient_method=True), gaussian_diffusion.py:264-271,350-364: the network sees every
        frame as latent; loss = sum(((mean + [t != 0] sigma z - x_t_minus_1) * obs_mask)^2) is differentiated w.r.t. x by
        autograd THROUGH THE ORACLE's own network; mean' = mean - 10 * alpha_t * grad / 2."""
        s = self.s
        obs = kw["obs_mask"]
        kw2 = dict(kw, obs_mask=torch.zeros_like(obs), latent_mask=obs + kw["latent_mask"])
        x = x.detach().clone().requires_grad_(True)
        tm = torch.tensor(s.timestep_map, dtype=t.dtype)[t]
        if s.rescale_timesteps:
            tm = tm.float() * (1000.0 / s.original_num_steps)
        with torch.enable_grad():
            eps = self.net.with_grad(x, tm, **kw2)
            x0 = _coef(s.sqrt_recip_alphas_cumprod, t, x) * x - _coef(s.sqrt_recipm1_alphas_cumprod, t, x) * eps
            if clip:
                x0 = x0.clamp(-1, 1)
            mean = _coef(s.posterior_mean_coef1, t, x) * x0 + _coef(s.posterior_mean_coef2, t, x) * x
            logvar = _coef(s.model_log_variance, t, x).expand(x.shape)
            nz = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
            smp = mean + nz * torch.exp(0.5 * logvar) * noise
            loss = (((smp - kw["x_t_minus_1"]) * obs) ** 2).sum()
            g, = torch.autograd.grad(loss, x)
        alpha = _coef(s.alphas, t, x)
        return dict(mean=(mean - 10 * alpha * g / 2).detach(), pred_xstart=x0.detach(), grad=g, log_variance=logvar.detach(),
                    variance=_coef(s.model_variance, t, x).expand(x.shape))

    def guided_p_sample(self, x, t, kw, noise, noise2, clip=True):
        o = self.guided_mean_variance(x, t, kw, noise, clip)
        nz = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        o["sample"] = o["mean"] + nz * torch.exp(0.5 * o["log_variance"]) * noise2
        return o

    def ddim_sample(self, x, t, kw, noise, eta=0.0, clip=True, eps=None):
        s = self.s
        o = self.mean_variance(x, t, kw, clip, eps)
        e = (_coef(s.sqrt_recip_alphas_cumprod, t, x) * x - o["pred_xstart"]) \
            / _coef(s.sqrt_recipm1_alphas_cumprod, t, x)
        ab, abp = _coef(s.alphas_cumprod, t, x), _coef(s.alphas_cumprod_prev, t, x)
        sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
        mean = o["pred_xstart"] * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * e
        nz = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        o["sample"] = mean + nz * sigma * noise
        return o

    def q_sample(self, x0, t, noise):
        s = self.s
        return _coef(s.sqrt_alphas_cumprod, t, x0) * x0 + _coef(s.sqrt_one_minus_alphas_cumprod, t, x0) * noise

    def p_sample_loop_progressive(self, shape, kw, observed_frames):
        """gaussian_diffusion.py:528-595 with noise=None: every draw comes from the global generator, in this order."""
        kw = dict(kw, observed_frames=observed_frames)
        img = torch.randn(*shape)
        B = shape[0]
        for i in range(self.num_timesteps)[::-1]:
            t = torch.tensor([i] * B)
            tm1 = torch.where(t - 1 < 0, t - 1 + self.num_timesteps, t - 1)          # table[-1] wraps (:565-568)
            kw["x_t_minus_1"] = self.q_sample(kw["x0"], tm1, torch.randn_like(kw["x0"]))
            kw["random_t"] = torch.floor(t * torch.rand(t.shape)).long()
            torch.randn_like(kw["x0"])                                            # x_random: drawn, unused in eval (unet.py:962)
            out = self.p_sample(img, t, kw, torch.randn_like(img))
            yield out, kw
            img = out["sample"]

    def ddim_sample_loop_progressive(self, shape, kw, eta=0.0):
        """gaussian_diffusion.py:702-748 with noise=None."""
        img = torch.randn(*shape)
        B = shape[0]
        for i in range(self.num_timesteps)[::-1]:
            out = self.ddim_sample(img, torch.tensor([i] * B), kw, torch.randn_like(img), eta=eta)
            yield out
            img = out["sample"]

    # ---- NLL path -------------------------------------------------------------------------------------------------
    def q_posterior_mean_variance(self, x_start, x_t, t):
        s = self.s
        mean = _coef(s.posterior_mean_coef1, t, x_t) * x_start + _coef(s.posterior_mean_coef2, t, x_t) * x_t
        return mean, _coef(s.posterior_variance, t, x_t).expand(x_t.shape), \
            _coef(s.posterior_log_variance_clipped, t, x_t).expand(x_t.shape)

    def vb_terms_bpd(self, x_start, x_t, t, kw, clip=True, latent_mask=None):
        true_mean, _, true_logvar = self.q_posterior_mean_variance(x_start, x_t, t)
        out = self.mean_variance(x_t, t, kw, clip)
        kl = mean_flat(normal_kl(true_mean, true_logvar, out["mean"], out["log_variance"]), latent_mask) / math.log(2.0)
        nll = -discretized_gaussian_log_likelihood(x_start, means=out["mean"], log_scales=0.5 * out["log_variance"])
        nll = mean_flat(nll, latent_mask) / math.log(2.0)
        return dict(output=torch.where(t == 0, nll, kl), pred_xstart=out["pred_xstart"])

    def prior_bpd(self, x_start, latent_mask=None):
        s = self.s
        t = torch.tensor([self.num_timesteps - 1] * x_start.shape[0])
        mean = _coef(s.sqrt_alphas_cumprod, t, x_start) * x_start
        logvar = _coef(np.log(1.0 - s.alphas_cumprod), t, x_start).expand(x_start.shape)
        return mean_flat(normal_kl(mean, logvar, 0.0, 0.0), latent_mask) / math.log(2.0)

    def calc_bpd_loop_subsampled(self, x_start, kw, clip=True, latent_mask=None, t_seq=None):
        s = self.s
        B = x_start.shape[0]
        if t_seq is None:
            t_seq = list(range(self.num_timesteps))[::-1]
        two_d = isinstance(t_seq, np.ndarray) and t_seq.ndim == 2
        if two_d:
            t_seq = t_seq.transpose()
        vb, xmse, mse = [], [], []
        for t in t_seq:
            tb = torch.tensor(t) if two_d else torch.tensor([t] * B)
            noise = torch.randn_like(x_start)
            x_t = self.q_sample(x_start, tb, noise)
            out = self.vb_terms_bpd(x_start, x_t, tb, kw, clip, latent_mask)
            vb.append(out["output"])
            xmse.append(mean_flat((out["pred_xstart"] - x_start) ** 2, latent_mask))
            eps = (_coef(s.sqrt_recip_alphas_cumprod, tb, x_t) * x_t - out["pred_xstart"]) / _coef(s.sqrt_recipm1_alphas_cumprod, tb, x_t)
            mse.append(mean_flat((eps - noise) ** 2, latent_mask))
        vb, xmse, mse = torch.stack(vb, 1), torch.stack(xmse, 1), torch.stack(mse, 1)
        prior = self.prior_bpd(x_start, latent_mask)
        return dict(total_bpd=vb.sum(1) + prior, prior_bpd=prior, vb=vb, xstart_mse=xmse, mse=mse)

    def window_loop(self, x_init, kw, noises, sampler="p", eta=0.0):
        """scripts/video_sample.py:149-168 -- the loop the drop-in target runs:
        start from x_init (= x0.clone()), one step per respaced index, high to low.
        `noises[i]` is the draw consumed at loop iteration i."""
        x = x_init.clone()
        B = x.shape[0]
        for i, step in enumerate(range(self.num_timesteps)[::-1]):
            t = torch.tensor([step] * B)
            if sampler == "p":
                x = self.p_sample(x, t, kw, noises[i])["sample"]
            else:
                x = self.ddim_sample(x, t, kw, noises[i], eta=eta)["sample"]
        return x

    def full_loop(self, batch, schedule_factory, obs_length, vertical_steps, observed_frames, noise_fn):
        """scripts/video_sample_full.py:50-323 (non-adaptive branch) -- the vertical + horizontal sampler:
        vertical phase = per window, the first `vertical_steps` timesteps in a row from the window's current frames
        (observed frames from x_0, :88-200); horizontal phase = per remaining timestep, the whole schedule again with
        one p_sample per window at that timestep and `observed_frames` as given (:202-315).
        `schedule_factory()` yields (obs_idx, latent_idx) lists; `noise_fn(shape)` is the next randn_like draw."""
        B = batch.shape[0]
        samples = torch.zeros_like(batch)
        samples[:, :obs_length] = batch[:, :obs_length]

        def window(obs_idx, lat_idx):
            x0 = torch.cat([samples[:, obs_idx], samples[:, lat_idx]], dim=1).clone()
            om = torch.zeros_like(x0[:, :, :1, :1, :1])
            om[:, :len(obs_idx)] = 1
            return x0, dict(x0=x0, obs_mask=om, latent_mask=1 - om, kinda_marg_mask=torch.zeros_like(om),
                            frame_indices=torch.tensor(list(obs_idx) + list(lat_idx)).repeat(B, 1))

        steps = list(range(self.num_timesteps))[::-1]
        if vertical_steps > 0:
            for obs_idx, lat_idx in schedule_factory():
                x0, kw = window(obs_idx, lat_idx)
                local = x0.clone()
                for ts in steps[:vertical_steps]:
                    local = self.p_sample(local, torch.tensor([ts] * B), kw, noise_fn(local.shape))["sample"]
                samples[:, lat_idx] = local[:, -len(lat_idx):]
        for ts in steps[vertical_steps:]:
            for obs_idx, lat_idx in schedule_factory():
                x0, kw = window(obs_idx, lat_idx)
                kw = dict(kw, observed_frames=observed_frames, x_t_minus_1=x0)
                local = self.p_sample(x0, torch.tensor([ts] * B), kw, noise_fn(x0.shape))["sample"]
                samples[:, lat_idx] = local[:, -len(lat_idx):]
        return samples
