"""NLL evaluation: the caller of `calc_bpd_loop_subsampled` (contract of scripts/video_nll.py:142-186 of the reference).

Per batch item the window is its observed frames followed by its latent frames; items are ragged, so shorter ones are
padded with zero frames that belong to neither mask.  The window is built here as ONE gather over a padded (item, slot) ->
frame table instead of per-item assignments.  Returned: every metric summed over the timestep axis and multiplied by the
window length (the reference reports bits per frame-dim summed, not averaged, over frames), as numpy arrays.  The reference's
script omits `observed_frames` / `x_t_minus_1` from model_kwargs, which its own `CondMargVideoModel.forward`
(unet.py:958-974) then fails to find; here they default to 'x_0' / x0.
"""
import torch
from torch.nn.utils.rnn import pad_sequence


def _window_table(obs_indices, lat_indices, n_items):
    """(frame table [n_items, F] long, observed [n_items, F] bool, latent [n_items, F] bool); F = the longest window."""
    pairs = list(zip(obs_indices, lat_indices))
    pairs += [((), ())] * (n_items - len(pairs))
    rows = [torch.as_tensor(list(o) + list(l), dtype=torch.long) for o, l in pairs]
    table = pad_sequence(rows, batch_first=True)
    slot = torch.arange(table.shape[1])[None, :]
    n_obs = torch.tensor([len(o) for o, _ in pairs])[:, None]
    n_all = torch.tensor([len(r) for r in rows])[:, None]
    return table, slot < n_obs, (slot >= n_obs) & (slot < n_all)


@torch.no_grad()
def run_bpd_evaluation(model, diffusion, batch, clip_denoised, obs_indices, lat_indices, t_seq=None, observed_frames="x_0"):
    dev = model.device
    table, observed, latent = _window_table(obs_indices, lat_indices, batch.shape[0])
    n_slots = table.shape[1]
    if n_slots > batch.shape[1]:
        raise ValueError(f"a window of {n_slots} frames does not fit videos of {batch.shape[1]}")
    item = torch.arange(batch.shape[0])[:, None]
    used = (observed | latent).to(batch.dtype)[:, :, None, None, None]
    x0 = (batch[item, table.to(batch.device)] * used.to(batch.device)).to(dev)
    as_mask = lambda m: m.to(device=dev, dtype=x0.dtype)[:, :, None, None, None]
    lat_mask = as_mask(latent)
    model_kwargs = dict(frame_indices=table.to(dev), x0=x0, obs_mask=as_mask(observed), latent_mask=lat_mask,
                        kinda_marg_mask=torch.zeros_like(lat_mask), observed_frames=observed_frames, x_t_minus_1=x0)
    per_t = diffusion.calc_bpd_loop_subsampled(model, x0, clip_denoised=clip_denoised, model_kwargs=model_kwargs,
                                               latent_mask=lat_mask, t_seq=t_seq)
    out = {}
    for name, v in per_t.items():
        total = v.sum(dim=1) if v.ndim > 1 else v
        out[name] = (total * n_slots).detach().cpu().numpy()
    return out
