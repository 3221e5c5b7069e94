"""Host side of the diffusion wrapper: schedule tables and the per-step coefficient plan.

Mirrors (host logic only; the arithmetic on frames runs in libdfot_hip.so):
  * DiscreteDiffusion._build_buffer / make_beta_schedule / cosine_simple_diffusion_schedule
      algorithms/dfot/diffusion/discrete_diffusion.py:94-168, noise_schedule.py:6-81
  * ddim_idx_to_noise_level                         discrete_diffusion.py:379-384
  * ContinuousDiffusion.model_predictions (level -> 0.125*logsnr[k])   continuous_diffusion.py:118-121
  * ddim_sample_step coefficient algebra            discrete_diffusion.py:454-483,527-536
  * posterior q(x_{k-1}|x_k,x_0) tables, ddpm_sample_step   discrete_diffusion.py:137-158,423-452
  * q_sample coefficients                           discrete_diffusion.py:242-250
  * _generate_scheduling_matrix                     algorithms/common/base_pytorch_video_algo.py:877-913
Everything here is numpy on the host, evaluated once per sampling call; the device kernels
only ever see small fp32 coefficient tables (one row per (branch-batch, token)).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class DiffusionConfig:
    timesteps: int = 1000
    sampling_timesteps: int = 50
    beta_schedule: str = "cosine_simple_diffusion"
    shifted: float = 0.125
    interpolated: bool = False
    logsnr_min: float = -15.0
    logsnr_max: float = 15.0
    clip_min: float = 1e-9
    ddim_sampling_eta: float = 0.0
    clip_noise: float = 20.0
    precond_scale: float = 0.125
    objective: str = "pred_v"
    # dfot_video.py:161,700-723 / discrete_diffusion.py:485-513: > 0 pulls the predicted clean context towards the given one through
    # d prediction / d x_t (one backbone backward per step); 0.0 in every shipped configuration
    reconstruction_guidance: float = 0.0
    # False: DiscreteDiffusion (the backbone receives the integer level index, discrete_diffusion.py:173-174);
    # True: ContinuousDiffusion (precond_scale * logsnr[k])
    is_continuous: bool = True
    # continuous training schedule and loss weighting (cfg.training_schedule {name: cosine, shift}, cfg.loss_weighting {strategy: sigmoid,
    # sigmoid_bias}; continuous_diffusion.py:53-74,106-116) -- read by the training / validation loss paths; the cosine logSNR limits
    # are logsnr_min / logsnr_max above
    training_schedule_name: str = "cosine"
    training_schedule_shift: float = 0.125
    loss_weighting_strategy: str = "sigmoid"
    loss_sigmoid_bias: float = -1.0

    def training_logsnr_tables(self, t):
        """t in [0,1] (torch tensor, any shape) -> (logsnr, alpha, sigma, loss weight) of ContinuousDiffusion.forward in fp32 torch, from
        THIS config (a schedule the engine does not implement raises instead of silently training on another one)"""
        import torch
        if self.training_schedule_name != "cosine":
            raise ValueError(f"unsupported training schedule '{self.training_schedule_name}' (only 'cosine')")
        if self.loss_weighting_strategy != "sigmoid":
            raise ValueError(f"unsupported continuous loss weighting '{self.loss_weighting_strategy}' (only 'sigmoid')")
        tt = t.detach().float().cpu()
        lo = torch.atan(torch.exp(-0.5 * torch.tensor(float(self.logsnr_max))))
        hi = torch.atan(torch.exp(-0.5 * torch.tensor(float(self.logsnr_min))))
        logsnr = -2 * torch.log(torch.tan(lo + tt * (hi - lo))) + 2 * torch.log(torch.tensor(float(self.training_schedule_shift)))
        return logsnr, torch.sigmoid(logsnr).sqrt(), torch.sigmoid(-logsnr).sqrt(), torch.sigmoid(float(self.loss_sigmoid_bias) - logsnr)


class Schedule:
    """float64 construction, float32 tables -- the dtype path of the reference's buffers."""

    def __init__(self, cfg: DiffusionConfig):
        self.cfg = cfg
        n = cfg.timesteps
        if cfg.beta_schedule == "cosine_simple_diffusion":
            t_lo = np.arctan(np.exp(-0.5 * cfg.logsnr_max))
            t_hi = np.arctan(np.exp(-0.5 * cfg.logsnr_min))
            u = np.linspace(0.0, 1.0, n, dtype=np.float64)
            lam = -2.0 * np.log(np.tan(t_lo + u * (t_hi - t_lo)))
            if cfg.shifted != 1.0:
                lam_s = lam + 2.0 * np.log(cfg.shifted)
                lam = u * lam + (1.0 - u) * lam_s if cfg.interpolated else lam_s
            abar = 1.0 / (1.0 + np.exp(-lam))
        elif cfg.beta_schedule == "cosine":
            u = np.linspace(0, n, n + 1, dtype=np.float64) / n
            f = np.cos((u + 0.008) / 1.008 * np.pi * 0.5) ** 2
            abar = (f / f[0])[1:]
        else:
            raise ValueError(f"unknown beta schedule {cfg.beta_schedule}")
        alphas = np.concatenate([abar[:1], abar[1:] / abar[:-1]])
        betas = np.clip(1.0 - alphas, cfg.clip_min, 1.0)
        abar = np.cumprod(1.0 - betas)
        self.alphas_cumprod = abar.astype(np.float32)
        self.sqrt_alphas_cumprod = np.sqrt(abar).astype(np.float32)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - abar).astype(np.float32)
        self.snr = (abar / (1.0 - abar)).astype(np.float32)  # float64 quotient, like the reference's buffer
        with np.errstate(divide="ignore"):  # the plain cosine schedule ends at abar = 0 (zero terminal SNR)
            self.logsnr = np.log(abar / (1.0 - abar)).astype(np.float32)
        # posterior q(x_{k-1} | x_k, x_0): DDPM sampling (sampling_timesteps == timesteps)
        abar_prev = np.concatenate([[1.0], abar[:-1]])
        post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
        self.posterior_mean_coef1 = (betas * np.sqrt(abar_prev) / (1.0 - abar)).astype(np.float32)
        self.posterior_mean_coef2 = ((1.0 - abar_prev) * np.sqrt(1.0 - betas) / (1.0 - abar)).astype(np.float32)
        self.posterior_log_variance_clipped = np.log(np.maximum(post_var, 1e-20)).astype(np.float32)

    @property
    def is_ddim_sampling(self) -> bool:  # discrete_diffusion.py:108
        return self.cfg.sampling_timesteps < self.cfg.timesteps

    # ---- index tables ---------------------------------------------------------------------
    def ddim_idx_to_noise_level(self, indices: np.ndarray) -> np.ndarray:
        c = self.cfg
        # the reference builds this table with torch.linspace(float32).long(); reuse torch so that
        # truncation of non-integer steps (e.g. 3 sampling steps) is bit-identical
        import torch
        steps = torch.linspace(-1, c.timesteps - 1, c.sampling_timesteps + 1).long().numpy()
        return steps[np.asarray(indices, dtype=np.int64)]

    def scheduling_matrix(self, kind: str, horizon: int, padding: int = 0) -> np.ndarray:
        s = self.cfg.sampling_timesteps
        if kind in ("full_sequence",):
            idx = np.repeat(np.arange(s, -1, -1)[:, None], horizon, axis=1)
        elif kind == "autoregressive":
            height = s + (horizon - 1) + 1
            idx = np.clip(s + np.arange(horizon)[None, :] - np.arange(height)[:, None], 0, s)
        elif kind == "interleaved":
            # _generate_interleaved_scheduling_matrix(horizon, 3, S): row r of token i (start = i % 3 + 1) is S while r < start,
            # then max(S - start - 3 * ((r - start) // 3), 0)
            r = np.arange(s + 3)[:, None]
            start = (np.arange(horizon) % 3 + 1)[None, :]
            idx = np.where(r < start, s, np.maximum(s - start - 3 * ((r - start) // 3), 0))
        elif kind == "gibbs":
            # each DDIM step i >= 1 is spread over `horizon` rows; row (i, j) has tokens 0..j at step i and the rest at i - 1
            step = np.repeat(np.arange(s + 1), horizon)[:, None]
            sub = np.tile(np.arange(horizon), s + 1)[:, None]
            tok = np.arange(horizon)[None, :]
            idx = s - np.where((tok <= sub) | (step == 0), step, step - 1)
        else:
            raise ValueError(f"unsupported scheduling matrix '{kind}'")
        levels = self.ddim_idx_to_noise_level(idx)
        if padding > 0:
            levels = np.concatenate([levels, np.full((levels.shape[0], padding), self.cfg.timesteps - 1, np.int64)], 1)
        return levels

    def refine_scheduling_matrix(self, horizon: int, goback_length: int, n_goback: int, padding: int = 0) -> np.ndarray:
        """_generate_refine_scheduling_matrix (base_pytorch_video_algo.py:943-970): the full-sequence ladder with, at every
        goback_length-th index, n_goback excursions goback_length steps back up and down again.
        The sampler that walks it is ``sampler._sample_sequence_refine``."""
        s = self.cfg.sampling_timesteps
        marks = set(range(1, s - goback_length, goback_length))
        seq = []
        for t in range(s, -1, -1):
            seq.append(t)
            if t in marks:
                for _ in range(n_goback):
                    seq += list(range(t + 1, t + goback_length + 1)) + list(range(t + goback_length - 1, t - 1, -1))
        levels = np.repeat(self.ddim_idx_to_noise_level(np.asarray(seq))[:, None], horizon, axis=1)
        if padding > 0:
            levels = np.concatenate([levels, np.full((levels.shape[0], padding), self.cfg.timesteps - 1, np.int64)], 1)
        return levels

    # ---- per-step coefficient tables (all float32, shape = levels.shape) -------------------
    def model_level(self, k: np.ndarray) -> np.ndarray:
        """what the backbone receives as `noise_levels`: precond_scale * logsnr[clamp(k,0)] (continuous) or the
        clamped level index itself (discrete; exact in float32)"""
        if not self.cfg.is_continuous:
            return np.clip(k, 0, None).astype(np.float32)
        return (np.float32(self.cfg.precond_scale) * self.logsnr[np.clip(k, 0, None)]).astype(np.float32)

    def q_sample_coef(self, k: np.ndarray):
        """x_k = a*x0 + b*noise; negative k indexes from the end like the reference's a[t]"""
        return self.sqrt_alphas_cumprod[k], self.sqrt_one_minus_alphas_cumprod[k]

    def ddim_coef(self, curr: np.ndarray, nxt: np.ndarray):
        """returns sa, s1 (v -> x0/eps at the clamped current level), an, cn (DDIM update) and keep"""
        f32 = np.float32
        kc = np.clip(curr, 0, None)
        alpha = self.alphas_cumprod[kc]
        alpha_next = np.where(nxt < 0, f32(1.0), self.alphas_cumprod[np.clip(nxt, 0, None)]).astype(f32)
        eta = f32(self.cfg.ddim_sampling_eta)
        with np.errstate(divide="ignore", invalid="ignore"):
            sig = eta * np.sqrt((f32(1) - alpha / alpha_next) * (f32(1) - alpha_next) / (f32(1) - alpha))
        sigma = np.where(nxt < 0, f32(0.0), sig).astype(f32)
        if eta == 0:
            sigma = np.zeros_like(alpha)
        cn = np.sqrt(f32(1) - alpha_next - sigma ** 2).astype(f32)
        return (self.sqrt_alphas_cumprod[kc], self.sqrt_one_minus_alphas_cumprod[kc], np.sqrt(alpha_next).astype(f32),
                cn, (curr == nxt).astype(f32), sigma)


    def ddpm_coef(self, curr: np.ndarray):
        """ddpm_sample_step in the ddim_coef form: with x0 = sa*x - s1*v and eps = sa*v + s1*x one has x = sa*x0 + s1*eps, so the
        posterior mean coef1*x0 + coef2*x = (coef1 + coef2*sa)*x0 + (coef2*s1)*eps.  keep = tokens at level -1 (clean);
        sigma = sqrt(posterior variance), 0 at level 0 (no noise there)."""
        f32 = np.float32
        kc = np.clip(curr, 0, None)
        sa, s1 = self.sqrt_alphas_cumprod[kc], self.sqrt_one_minus_alphas_cumprod[kc]
        c1, c2 = self.posterior_mean_coef1[kc], self.posterior_mean_coef2[kc]
        sigma = np.where(kc > 0, np.exp(f32(0.5) * self.posterior_log_variance_clipped[kc]), f32(0)).astype(f32)
        return sa, s1, (c1 + c2 * sa).astype(f32), (c2 * s1).astype(f32), (curr == -1).astype(f32), sigma

    # ---- training-loss weights of DiscreteDiffusion (compute_loss_weights, discrete_diffusion.py:274-343), objective pred_v ----
    def loss_weights(self, k: np.ndarray, strategy: str = "fused_min_snr", snr_clip: float = 5.0, cum_snr_decay: float = 0.9,
                     sigmoid_bias: float = -1.0, causal: bool = False) -> np.ndarray:
        """k (B,T) integer levels -> (B,T) float32 weights of the v-space squared error."""
        f32 = np.float32
        if strategy == "uniform":
            return np.ones(k.shape, f32)
        snr_tab = self.snr
        snr = snr_tab[k]
        floor = f32(1e-8)
        if strategy == "sigmoid":
            with np.errstate(divide="ignore"):
                eps_w = f32(1) / (f32(1) + np.exp(np.log(snr_tab)[k] - f32(sigmoid_bias)))
        elif strategy == "min_snr":
            eps_w = np.minimum(snr_tab, f32(snr_clip))[k] / np.maximum(snr, floor)
        elif strategy == "fused_min_snr":
            d = f32(cum_snr_decay)
            nclip = np.minimum(snr_tab, f32(snr_clip))[k] / f32(snr_clip)
            nsnr = snr / f32(snr_clip)

            def history(x):  # EMA over the tokens BEFORE each position (zero at the first one)
                out = np.zeros_like(x)
                ema = x[:, 0].copy()
                for t in range(1, x.shape[1]):
                    out[:, t] = ema
                    ema = d * ema + (f32(1) - d) * x[:, t]
                return out
            cum = history(nclip) if causal else f32(0.5) * (history(nclip[:, ::-1])[:, ::-1] + history(nclip))
            keep = f32(1) - cum * d
            clipped = (f32(1) - keep * (f32(1) - nclip)) * f32(snr_clip)
            snr = (f32(1) - keep * (f32(1) - nsnr)) * f32(snr_clip)
            eps_w = clipped / np.maximum(snr, floor)
        else:
            raise ValueError(f"unknown loss weighting strategy {strategy}")
        return (eps_w * snr / (snr + f32(1))).astype(f32)
