"""Forward half of the reference's ``training_step`` on the MI355X engine (no backward / optimizer yet).

  * ``TrainingNoise``                -- ``BaseVideoAlgo._get_training_noise_levels``
                                        (algorithms/common/base_pytorch_video_algo.py:778-874): per-token independent levels
                                        (Diffusion Forcing), uniform, interleaved; fixed / variable context with binary dropout;
                                        uniform future; unavailable frames -> full noise.  Host-side, driven by a torch.Generator
                                        exactly like the reference, so a seeded CPU generator reproduces its draws.
  * ``training_step_forward``        -- ``DFoTVideo.training_step`` (algorithms/dfot/dfot_video.py:41-75) up to the loss:
                                        noise levels -> noised forward -> weighted loss -> ``_reweight_loss`` with the masks.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch


@dataclass
class ContextTraining:
    enabled: bool = False
    prob: float = 0.25            # variable_context only
    indices: Optional[Sequence[int]] = None  # fixed_context only (default: the first n_context_tokens)
    dropout: float = 0.0


@dataclass
class TrainingNoise:
    noise_level: str = "random_independent"  # | "random_uniform" | "interleaved"
    is_continuous: bool = True
    timesteps: int = 1000
    n_context_tokens: int = 1
    uniform_future: bool = False
    fixed_context: ContextTraining = field(default_factory=ContextTraining)
    variable_context: ContextTraining = field(default_factory=ContextTraining)

    def _rand(self, shape, generator):
        if self.is_continuous:
            return torch.rand(shape, generator=generator)
        return torch.randint(0, self.timesteps, shape, generator=generator)

    def sample(self, batch_size: int, n_tokens: int, masks: torch.Tensor, generator: Optional[torch.Generator] = None,
               training: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """masks: (B, T, ...) availability of the frames (bool or 0/1).  Returns (noise_levels (B,T), masks for the loss)."""
        bern = lambda shape, p: torch.bernoulli(torch.full(shape, float(p)), generator=generator)
        context_mask = None
        if self.variable_context.enabled:
            assert not self.fixed_context.enabled, "Cannot use both fixed and variable context"
            context_mask = bern((batch_size, n_tokens), self.variable_context.prob).bool()
        elif self.fixed_context.enabled:
            idx = list(self.fixed_context.indices) if self.fixed_context.indices else list(range(self.n_context_tokens))
            context_mask = torch.zeros(batch_size, n_tokens, dtype=torch.bool)
            context_mask[:, idx] = True

        if self.noise_level == "random_independent":
            levels = self._rand((batch_size, n_tokens), generator)
        elif self.noise_level == "random_uniform":
            levels = self._rand((batch_size, 1), generator).repeat(1, n_tokens)
        elif self.noise_level == "interleaved":
            first, second = self._rand((batch_size, 1), generator), self._rand((batch_size, 1), generator)
            levels = torch.zeros(batch_size, n_tokens, dtype=first.dtype)
            levels[:, ::2] = first
            levels[:, 1::2] = second
        else:
            raise ValueError(f"unknown noise_level '{self.noise_level}'")
        if self.uniform_future:
            c = self.n_context_tokens
            levels[:, c:] = self._rand((batch_size, 1), generator).repeat(1, n_tokens - c)

        avail = masks.bool().flatten(2).any(-1) if masks.ndim > 2 else masks.bool()
        full = 1 if self.is_continuous else self.timesteps - 1
        levels = torch.where(avail, levels, torch.full_like(levels, full))

        if context_mask is not None:
            ctx = self.variable_context if self.variable_context.enabled else self.fixed_context
            drop = bern((batch_size, 1), ctx.dropout if training else 0.0)
            if not self.is_continuous:
                drop = drop.long() * (self.timesteps - 1)
            levels = torch.where(context_mask, drop.to(levels.dtype), levels)
            cm = context_mask.reshape(batch_size, n_tokens, *([1] * (masks.ndim - 2)))
            masks = torch.where(cm, torch.zeros_like(masks), masks)
        return levels, masks


@torch.no_grad()
def training_step_forward(sampler, xs: torch.Tensor, masks: torch.Tensor, noise_cfg: TrainingNoise,
                          conditions: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                          noise: Optional[torch.Tensor] = None, loss_weighting: Optional[Dict] = None) -> Dict[str, torch.Tensor]:
    """DFoTVideo.training_step without the backward: returns {"loss", "xs_pred", "noise_levels", "per_token"}."""
    b, t = xs.shape[:2]
    levels, loss_masks = noise_cfg.sample(b, t, masks, generator)
    lm = loss_masks.float().flatten(2).mean(-1) if loss_masks.ndim > 2 else loss_masks.float()
    if noise_cfg.is_continuous:
        x_pred, _, per_token = sampler.denoising_loss(xs, conditions, levels, noise=noise)
    else:
        x_pred, _, per_token = sampler.discrete_denoising_loss(xs, levels, noise=noise, loss_weighting=loss_weighting)
    per_token = per_token * lm.to(per_token.device)
    return {"loss": per_token.mean(), "xs_pred": x_pred, "noise_levels": levels, "per_token": per_token}


def lr_at_step(step: int, base_lr: float, name: str = "constant_with_warmup", num_warmup_steps: int = 10000, num_training_steps: int = 0,
               num_processes: int = 1) -> float:
    """Learning rate of optimizer step `step` (0-based) under the reference's schedulers (`transformers.get_scheduler(name=cfg.lr_scheduler.name,
    num_warmup_steps=...)` stepped once per optimizer step: experiments/simple_video_generation.py:271, realestate10k_video_generation.yaml:19-22
    `constant_with_warmup`, 10000 warm-up steps): linear warm-up from 0, then constant / linear decay / cosine decay.  A LambdaLR is
    at epoch s when optimizer step s runs (it is stepped AFTER each optimizer step), so step s runs at base * s / warmup: the first
    step has learning rate 0 and step `num_warmup_steps` is the first at the base rate.
    num_processes: the reference hands its scheduler to `accelerator.prepare` (simple_video_generation.py:183); the AcceleratedScheduler
    then steps the wrapped scheduler `num_processes` times per optimizer step (split_batches = False) and not at all on gradient-
    accumulation micro-steps, so on N GPUs the schedule runs N times faster: optimizer step s is at scheduler epoch s * N (the
    10000-step warm-up of the RE10K recipe takes 10000 / 12 optimizer steps on its 12 GPUs).  Pass the world size to reproduce that;
    1 (default) is the bare LambdaLR."""
    import math
    if num_processes < 1:
        raise ValueError("num_processes must be >= 1")
    step = step * int(num_processes)
    warm = min(1.0, step / max(1, num_warmup_steps)) if num_warmup_steps > 0 else 1.0
    if name in ("constant", "constant_with_warmup"):
        return base_lr * (warm if name == "constant_with_warmup" else 1.0)
    if step < num_warmup_steps:
        return base_lr * warm
    if num_training_steps <= num_warmup_steps:
        raise ValueError("lr schedule: num_training_steps must exceed num_warmup_steps for a decaying schedule")
    prog = (step - num_warmup_steps) / (num_training_steps - num_warmup_steps)
    if name == "linear":
        return base_lr * max(0.0, 1.0 - prog)
    if name == "cosine":
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    raise ValueError(f"unknown lr scheduler '{name}'")
