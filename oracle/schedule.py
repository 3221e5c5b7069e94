"""Noise-schedule tables and sampling-schedule matrices (oracle, CPU).

Restates:
  * cosine_simple_diffusion_schedule  -- algorithms/dfot/diffusion/noise_schedule.py:48-81
  * make_beta_schedule (beta clip)    -- noise_schedule.py:6-33
  * DiscreteDiffusion._build_buffer   -- algorithms/dfot/diffusion/discrete_diffusion.py:94-168
  * ddim_idx_to_noise_level           -- discrete_diffusion.py:379-384
  * _generate_scheduling_matrix       -- algorithms/common/base_pytorch_video_algo.py:877-913
  * pyramid / interleaved variants    -- base_pytorch_video_algo.py:915-947
  * CosineNoiseSchedule (training)    -- algorithms/dfot/diffusion/continuous_diffusion.py:46-92
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class ScheduleTables:
    """fp32 lookup tables indexed by the integer noise level k in [0, timesteps)."""

    alphas_cumprod: torch.Tensor
    sqrt_alphas_cumprod: torch.Tensor
    sqrt_one_minus_alphas_cumprod: torch.Tensor
    logsnr: torch.Tensor
    timesteps: int
    snr: torch.Tensor = None  # alphas_cumprod / (1 - alphas_cumprod) formed in float64 like the reference's buffer
    # posterior q(x_{k-1} | x_k, x_0) of DDPM sampling (discrete_diffusion.py:137-158), formed in float64 like the reference's buffers
    posterior_mean_coef1: torch.Tensor = None
    posterior_mean_coef2: torch.Tensor = None
    posterior_log_variance_clipped: torch.Tensor = None


def _alphas_cumprod_cosine_simple(timesteps: int, logsnr_min: float, logsnr_max: float,
                                  shifted: float, interpolated: bool) -> torch.Tensor:
    f64 = torch.float64
    lo = torch.atan(torch.exp(torch.tensor(-0.5 * logsnr_max, dtype=f64)))
    hi = torch.atan(torch.exp(torch.tensor(-0.5 * logsnr_min, dtype=f64)))
    u = torch.linspace(0, 1, timesteps, dtype=f64)
    logsnr = -2.0 * torch.log(torch.tan(lo + u * (hi - lo)))
    if shifted != 1.0:
        moved = logsnr + 2.0 * math.log(shifted)
        logsnr = u * logsnr + (1 - u) * moved if interpolated else moved
    return torch.sigmoid(logsnr)


def _alphas_cumprod_cosine(timesteps: int, s: float = 0.008) -> torch.Tensor:
    # noise_schedule.py:36-45 (used by the discrete K600 configuration)
    u = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    ac = torch.cos((u + s) / (1 + s) * math.pi * 0.5) ** 2
    return (ac / ac[0])[1:]


def build_tables(timesteps: int = 1000, beta_schedule: str = "cosine_simple_diffusion",
                 shifted: float = 0.125, interpolated: bool = False,
                 logsnr_min: float = -15.0, logsnr_max: float = 15.0,
                 clip_min: float = 1e-9) -> ScheduleTables:
    """alphas_cumprod is rebuilt from clipped betas exactly as the reference does
    (schedule -> alpha ratios -> betas clipped to [1e-9, 1] -> cumprod), all in
    float64, then cast to float32."""
    if beta_schedule == "cosine_simple_diffusion":
        ac = _alphas_cumprod_cosine_simple(timesteps, logsnr_min, logsnr_max, shifted, interpolated)
    elif beta_schedule == "cosine":
        ac = _alphas_cumprod_cosine(timesteps)
    else:
        raise ValueError(f"oracle: unsupported beta schedule {beta_schedule}")
    ratio = torch.cat([ac[:1], ac[1:] / ac[:-1]])
    betas = torch.clip(1.0 - ratio, clip_min, 1.0)
    ac = torch.cumprod(1.0 - betas, dim=0)
    snr = ac / (1.0 - ac)
    f32 = torch.float32
    ac_prev = torch.cat([torch.ones(1, dtype=ac.dtype), ac[:-1]])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return ScheduleTables(
        posterior_mean_coef1=(betas * torch.sqrt(ac_prev) / (1.0 - ac)).to(f32),
        posterior_mean_coef2=((1.0 - ac_prev) * torch.sqrt(1.0 - betas) / (1.0 - ac)).to(f32),
        posterior_log_variance_clipped=torch.log(post_var.clamp(min=1e-20)).to(f32),
        alphas_cumprod=ac.to(f32),
        sqrt_alphas_cumprod=torch.sqrt(ac).to(f32),
        sqrt_one_minus_alphas_cumprod=torch.sqrt(1.0 - ac).to(f32),
        logsnr=torch.log(snr).to(f32),
        snr=snr.to(f32),
        timesteps=timesteps,
    )


def ddim_levels(timesteps: int, sampling_timesteps: int) -> torch.Tensor:
    """Table of sampling_timesteps+1 integer noise levels, entry 0 == -1 (clean)."""
    return torch.linspace(-1, timesteps - 1, sampling_timesteps + 1).long()


def scheduling_matrix(kind: str, horizon: int, padding: int, timesteps: int,
                      sampling_timesteps: int) -> torch.Tensor:
    """(M, horizon+padding) int64 noise levels; padded columns are pure noise."""
    s = sampling_timesteps
    if kind == "full_sequence":
        idx = np.arange(s, -1, -1)[:, None].repeat(horizon, axis=1)
    elif kind == "autoregressive":
        rows = s + (horizon - 1) + 1
        m = np.arange(rows)[:, None]
        t = np.arange(horizon)[None, :]
        idx = np.clip(s + t - m, 0, s)
    elif kind == "interleaved":
        # base_pytorch_video_algo.py:914-936 with interleaved_size 3: token i waits (i % 3) + 1 rows at pure noise, then drops
        # three DDIM indices every three rows until it reaches 0
        size = 3
        cols = []
        for i in range(horizon):
            start = i % size + 1
            col = [s] * start
            j = 0
            while len(col) < s + size:
                level = max(s - start - size * j, 0)
                col += [level] * (size if level > 0 else s + size - len(col))
                j += 1
            cols.append(col[: s + size])
        idx = np.array(cols).T
    elif kind == "gibbs":
        # base_pytorch_video_algo.py:884-905: every DDIM step becomes `horizon` rows that move one more token to the new level
        base = np.arange(s, -1, -1)
        idx = np.zeros(((s + 1) * horizon, horizon), dtype=np.int64)
        for i in range(s + 1):
            for j in range(horizon):
                idx[i * horizon + j] = base[i] if i == 0 else np.where(np.arange(horizon) <= j, base[i], base[i - 1])
    else:
        raise ValueError(f"oracle: unsupported scheduling matrix {kind}")
    levels = ddim_levels(timesteps, s)[torch.from_numpy(idx).long()]
    if padding > 0:
        pad = torch.full((levels.shape[0], padding), timesteps - 1, dtype=torch.long)
        levels = torch.cat([levels, pad], dim=1)
    return levels


def refine_scheduling_matrix(horizon: int, goback_length: int, n_goback: int, padding: int, timesteps: int,
                             sampling_timesteps: int) -> torch.Tensor:
    """_generate_refine_scheduling_matrix (base_pytorch_video_algo.py:949-976): the full-sequence ladder; at every DDIM index in
    range(1, S - goback_length, goback_length) it climbs goback_length indices back up and down again, n_goback times."""
    s = sampling_timesteps
    marks = list(range(1, s - goback_length, goback_length))
    seq = []
    for t in range(s, -1, -1):
        seq.append(t)
        if t in marks:
            for _ in range(n_goback):
                seq.extend(range(t + 1, t + goback_length + 1))
                seq.extend(range(t + goback_length - 1, t - 1, -1))
    levels = ddim_levels(timesteps, s)[torch.tensor(seq).long()][:, None].repeat(1, horizon)
    if padding > 0:
        levels = torch.cat([levels, torch.full((levels.shape[0], padding), timesteps - 1, dtype=torch.long)], dim=1)
    return levels


def training_logsnr(t: torch.Tensor, shift: float = 0.125, logsnr_min: float = -15.0,
                    logsnr_max: float = 15.0) -> torch.Tensor:
    """Continuous-time cosine logSNR(t), t in [0,1] (fp32 like the reference buffers)."""
    lo = torch.atan(torch.exp(-0.5 * torch.tensor(logsnr_max, dtype=torch.float32)))
    hi = torch.atan(torch.exp(-0.5 * torch.tensor(logsnr_min, dtype=torch.float32)))
    sh = 2 * torch.log(torch.tensor(shift, dtype=torch.float32))
    return -2 * torch.log(torch.tan(lo + t * (hi - lo))) + sh
