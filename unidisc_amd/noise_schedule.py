"""Noise schedules on the hot path (reference models/noise_schedule.py)."""
from __future__ import annotations

import torch


class LogLinearNoise(torch.nn.Module):
    """models/noise_schedule.py:128-157: total noise -log1p(-(1-eps) t), so the move chance is (1-eps) t."""

    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps
        self.sigma_max = self.total_noise(torch.tensor(1.0, dtype=torch.float32))
        self.sigma_min = self.eps + self.total_noise(torch.tensor(0.0, dtype=torch.float32))

    def rate_noise(self, t):
        return (1 - self.eps) / (1 - (1 - self.eps) * t)

    def total_noise(self, t):
        return -torch.log1p(-(1 - self.eps) * t)

    def forward(self, t):  # noise_schedule.py:37-43
        return self.total_noise(t), self.rate_noise(t)

    def importance_sampling_transformation(self, t):
        f_T = torch.log1p(-torch.exp(-self.sigma_max))
        f_0 = torch.log1p(-torch.exp(-self.sigma_min))
        sigma_t = -torch.log1p(-torch.exp(t * f_T + (1 - t) * f_0))
        return -torch.expm1(-sigma_t) / (1 - self.eps)


def get_noise(config, dtype=torch.float32):
    """models/noise_schedule.py:13-28 — only the default log-linear schedule is on the hot path."""
    from .dit import cfg_get

    kind = cfg_get(cfg_get(config, "noise"), "type", "loglinear")
    if kind != "loglinear":
        raise NotImplementedError(f"unidisc_amd: noise.type={kind} is not on the denoising hot path (every shipped config uses loglinear)")
    return LogLinearNoise()
