"""Gaussian likelihood helpers (reference losses.py).  Only reached with learn_sigma / use_kl, which are
off by default and outside the native hot path; plain device tensor ops."""
import numpy as np
import torch as th


def normal_kl(mean1, logvar1, mean2, logvar2):
    tensor = next((o for o in (mean1, logvar1, mean2, logvar2) if isinstance(o, th.Tensor)), None)
    assert tensor is not None, "at least one argument must be a Tensor"
    logvar1, logvar2 = [x if isinstance(x, th.Tensor) else th.tensor(x).to(tensor) for x in (logvar1, logvar2)]
    return 0.5 * (-1.0 + logvar2 - logvar1 + th.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * th.exp(-logvar2))


def approx_standard_normal_cdf(x):
    return 0.5 * (1.0 + th.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * th.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    assert x.shape == means.shape == log_scales.shape
    centered = x - means
    inv_stdv = th.exp(-log_scales)
    cdf_plus = approx_standard_normal_cdf(inv_stdv * (centered + 1.0 / 255.0))
    cdf_min = approx_standard_normal_cdf(inv_stdv * (centered - 1.0 / 255.0))
    log_cdf_plus = th.log(cdf_plus.clamp(min=1e-12))
    log_one_minus_cdf_min = th.log((1.0 - cdf_min).clamp(min=1e-12))
    cdf_delta = cdf_plus - cdf_min
    return th.where(x < -0.999, log_cdf_plus,
                    th.where(x > 0.999, log_one_minus_cdf_min, th.log(cdf_delta.clamp(min=1e-12))))
