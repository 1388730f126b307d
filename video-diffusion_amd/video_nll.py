"""NLL evaluation: the caller of `calc_bpd_loop_subsampled` (scripts/video_nll.py:142-186 of the reference).

`run_bpd_evaluation` builds the per-item (observed, latent) window exactly as the reference does -- observed frames first,
then latent ones, masks and frame_indices filled per batch item, ragged items padded with zeros -- and returns the
metrics summed over the timestep axis and multiplied by the number of frames (the reference reports bits per frame-dim
summed over frames).  The reference's script omits `observed_frames` / `x_t_minus_1` from model_kwargs, which its own
`CondMargVideoModel.forward` (unet.py:958-974) then fails to find; here they default to 'x_0' / x0.
"""
import torch


@torch.no_grad()
def run_bpd_evaluation(model, diffusion, batch, clip_denoised, obs_indices, lat_indices, t_seq=None, observed_frames="x_0"):
    max_frames = max(len(o) + len(l) for o, l in zip(obs_indices, lat_indices))
    dev = model.device
    x0 = torch.zeros_like(batch[:, :max_frames]).to(dev)
    obs_mask = torch.zeros_like(x0[:, :, :1, :1, :1])
    lat_mask = torch.zeros_like(x0[:, :, :1, :1, :1])
    kinda_marg_mask = torch.zeros_like(x0[:, :, :1, :1, :1])
    frame_indices = torch.zeros_like(x0[:, :, 0, 0, 0]).long()
    for i, (obs_i, lat_i) in enumerate(zip(obs_indices, lat_indices)):
        no, nl = len(obs_i), len(lat_i)
        x0[i, :no] = batch[i, obs_i].to(dev)
        obs_mask[i, :no] = 1.0
        frame_indices[i, :no] = torch.tensor(obs_i, device=dev)
        x0[i, no:no + nl] = batch[i, lat_i].to(dev)
        lat_mask[i, no:no + nl] = 1.0
        frame_indices[i, no:no + nl] = torch.tensor(lat_i, device=dev)
    model_kwargs = dict(frame_indices=frame_indices, x0=x0, obs_mask=obs_mask, latent_mask=lat_mask,
                        kinda_marg_mask=kinda_marg_mask, observed_frames=observed_frames, x_t_minus_1=x0)
    metrics = diffusion.calc_bpd_loop_subsampled(model, x0, clip_denoised=clip_denoised, model_kwargs=model_kwargs,
                                                 latent_mask=lat_mask, t_seq=t_seq)
    metrics = {k: v.sum(dim=1) if v.ndim > 1 else v for k, v in metrics.items()}
    metrics = {k: v * max_frames for k, v in metrics.items()}      # sum (rather than mean) over the frame dimension
    return {k: v.detach().cpu().numpy() for k, v in metrics.items()}
