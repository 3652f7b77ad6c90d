"""The noise schedule of the hot path: log-linear total noise (reference models/noise_schedule.py:128-150, `get_noise` :13-28).

With total noise sigma(t) = -log(1 - (1 - eps) t) the absorbing-state move chance 1 - exp(-sigma) is simply (1 - eps) t, and the loss
weight sigma'(t) / expm1(sigma(t)) is 1 / t up to eps.  On the GPU step both values come out of `udm_sample_t_noise` together with t itself
(kernels.sample_t_noise); the functions below are the same two formulas for the callers that hold a `t` already (sampler loops, CPU tests).
"""
from __future__ import annotations

import torch

DEFAULT_EPS = 1e-3


def loglinear_total_noise(t: torch.Tensor, eps: float = DEFAULT_EPS) -> torch.Tensor:
    """sigma(t) = -log1p(-(1 - eps) t)   (noise_schedule.py:145-146)"""
    return torch.log1p(t * -(1 - eps)).neg()


def loglinear_rate_noise(t: torch.Tensor, eps: float = DEFAULT_EPS) -> torch.Tensor:
    """sigma'(t) = (1 - eps) / (1 - (1 - eps) t)   (noise_schedule.py:142-143)"""
    keep = 1 - eps
    return keep / (1 - keep * t)


class LogLinearNoise:
    """Callable holder of `eps`: `noise(t) -> (sigma, sigma')` like the reference module (it has no parameters or buffers, so it is not an nn.Module here
    and does not show up in a state dict - the reference's does not either)."""

    def __init__(self, eps: float = DEFAULT_EPS):
        self.eps = float(eps)

    def total_noise(self, t):
        return loglinear_total_noise(t, self.eps)

    def rate_noise(self, t):
        return loglinear_rate_noise(t, self.eps)

    def __call__(self, t):
        return loglinear_total_noise(t, self.eps), loglinear_rate_noise(t, self.eps)


def get_noise(config, dtype=torch.float32):
    """Every shipped configuration uses `noise.type: loglinear`; other schedules are not on the denoising hot path."""
    from .dit import cfg_get

    kind = cfg_get(cfg_get(config, "noise"), "type", "loglinear")
    if kind != "loglinear":
        raise NotImplementedError(f"unidisc_amd: noise.type={kind} is not on the denoising hot path (every shipped config uses loglinear)")
    return LogLinearNoise()
