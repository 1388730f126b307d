"""ORACLE (test infrastructure, NOT product code) -- likelihood helpers of the NLL path.

Restates (reference file:line, relative to /root/reference/improved_diffusion):
  - normal_kl                              losses.py:13-35
  - approx_standard_normal_cdf             losses.py:38-43
  - discretized_gaussian_log_likelihood    losses.py:46-76
  - mean_flat (with mask)                  nn.py:73-77
Pinned through tests/golden/nll_tiny.npz (imported reference), tests/test_oracle_golden.py.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this package.
"""
import math

import torch


def normal_kl(mean1, logvar1, mean2, logvar2):
    ref = next(o for o in (mean1, logvar1, mean2, logvar2) if torch.is_tensor(o))
    logvar1, logvar2 = [v if torch.is_tensor(v) else torch.tensor(v).to(ref) for v in (logvar1, logvar2)]
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))


def approx_standard_normal_cdf(x):
    return 0.5 * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    assert x.shape == means.shape == log_scales.shape
    centered = x - means
    inv_stdv = torch.exp(-log_scales)
    cdf_plus = approx_standard_normal_cdf(inv_stdv * (centered + 1.0 / 255.0))
    cdf_min = approx_standard_normal_cdf(inv_stdv * (centered - 1.0 / 255.0))
    log_cdf_plus = torch.log(cdf_plus.clamp(min=1e-12))
    log_one_minus_cdf_min = torch.log((1.0 - cdf_min).clamp(min=1e-12))
    delta = cdf_plus - cdf_min
    return torch.where(x < -0.999, log_cdf_plus,
                       torch.where(x > 0.999, log_one_minus_cdf_min, torch.log(delta.clamp(min=1e-12))))


def mean_flat(tensor, mask=None):
    if mask is not None:
        tensor = tensor * mask
    return tensor.mean(dim=list(range(1, tensor.dim())))
