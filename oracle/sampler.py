"""Diffusion step + sampler drivers (oracle, CPU fp32).

Restates:
  * extract / q_sample / predict_start_from_v / predict_noise_from_v
                                           -- algorithms/dfot/diffusion/discrete_diffusion.py:21-24,213-250
  * ContinuousDiffusion.model_predictions  -- diffusion/continuous_diffusion.py:118-138
  * ddim_sample_step (eta, sigma, c, keep-mask) -- discrete_diffusion.py:454-538
  * DFoTVideo._sample_sequence             -- algorithms/dfot/dfot_video.py:516-763
  * DFoTVideo._predict_sequence            -- dfot_video.py:362-514
  * DFoTVideo._predict_videos              -- dfot_video.py:114-179
  * DFoTVideo._interpolate_videos (planner + batched execution) -- dfot_video.py:181-360
  * _pad_to_max_tokens                     -- algorithms/common/base_pytorch_video_algo.py:666-682
  * ContinuousDiffusion.forward (training loss) -- continuous_diffusion.py:140-167

Randomness is injected: every place the reference draws noise calls
``noise_fn(tag, shape)`` so that golden traces can be replayed exactly.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import guidance as hg
from . import schedule as sch

ModelFn = Callable[[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]], torch.Tensor]
CondFn = Callable[[torch.Tensor], torch.Tensor]


def default_noise_fn(generator: Optional[torch.Generator] = None, clip: float = 20.0) -> hg.NoiseFn:
    def fn(tag: str, shape: tuple) -> torch.Tensor:
        return torch.randn(shape, generator=generator).clamp(-clip, clip)
    return fn


def replay_noise_fn(recorded: List[torch.Tensor], clip: float = 20.0) -> hg.NoiseFn:
    """Pops recorded raw normal draws in call order (shapes must agree) and applies the
    clamp the reference applies at that call site (none for the 'excluded' draw)."""
    queue = list(recorded)

    def fn(tag: str, shape: tuple) -> torch.Tensor:
        t = queue.pop(0)
        if tuple(t.shape) != tuple(shape):
            raise AssertionError(f"noise replay shape mismatch at {tag}: {tuple(t.shape)} vs {shape}")
        return t if tag == "excluded" else t.clamp(-clip, clip)
    fn.queue = queue  # type: ignore[attr-defined]
    return fn


def _ext(a: torch.Tensor, nd: int) -> torch.Tensor:
    return a.reshape(*a.shape, *([1] * (nd - a.ndim)))


@dataclass
class Diffusion:
    tables: sch.ScheduleTables
    model: ModelFn
    sampling_timesteps: int = 50
    eta: float = 0.0
    clip_noise: float = 20.0
    precond_scale: float = 0.125
    is_continuous: bool = True  # False: DiscreteDiffusion.model_predictions hands the integer level to the backbone

    def q_sample(self, x0, k, noise):
        t = self.tables
        return (_ext(t.sqrt_alphas_cumprod[k], x0.ndim) * x0
                + _ext(t.sqrt_one_minus_alphas_cumprod[k], x0.ndim) * noise)

    def predictions(self, x, k, cond, cond_mask):
        t = self.tables
        # continuous_diffusion.py model_predictions: precond_scale * logsnr[k]; discrete_diffusion.py:173-174: k itself
        v = self.model(x, self.precond_scale * t.logsnr[k] if self.is_continuous else k, cond, cond_mask)
        a, s = _ext(t.sqrt_alphas_cumprod[k], x.ndim), _ext(t.sqrt_one_minus_alphas_cumprod[k], x.ndim)
        return v, a * x - s * v, a * v + s * x  # v, x0, eps

    def ddim_step(self, x, curr, nxt, cond, cond_mask, noise=None, guidance_fn=None):
        ac = self.tables.alphas_cumprod
        kc = curr.clamp(min=0)
        alpha = ac[kc]
        alpha_next = torch.where(nxt < 0, torch.ones_like(alpha), ac[nxt.clamp(min=0)])
        sigma = torch.where(nxt < 0, torch.zeros_like(alpha),
                            self.eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt())
        c = (1 - alpha_next - sigma ** 2).sqrt()
        if guidance_fn is not None:
            # discrete_diffusion.py:485-513: the prediction is differentiated w.r.t. x_t; the negative gradient of the guidance
            # "likelihood" shifts the predicted noise, and x0 is recomputed from it (kept where alphas_cumprod = 0)
            with torch.enable_grad():
                xg = x.detach().requires_grad_()
                _, x0_raw, eps = self.predictions(xg, kc, cond, cond_mask)
                like = guidance_fn(xk=xg, pred_x0=x0_raw, alpha_cumprod=_ext(alpha, x.ndim))
                grad = torch.nan_to_num(-torch.autograd.grad(like, xg)[0], nan=0.0)
            t = self.tables
            eps = eps.detach() + _ext((1 - alpha).sqrt(), x.ndim) * grad
            x0 = torch.where(_ext(alpha, x.ndim) > 0,
                             (x - _ext(t.sqrt_one_minus_alphas_cumprod[kc], x.ndim) * eps) / _ext(t.sqrt_alphas_cumprod[kc], x.ndim),
                             x0_raw.detach())
        else:
            _, x0, eps = self.predictions(x, kc, cond, cond_mask)
        out = x0 * _ext(alpha_next.sqrt(), x.ndim) + eps * _ext(c, x.ndim)
        if self.eta != 0.0:
            out = out + _ext(sigma, x.ndim) * noise
        return torch.where(_ext(curr == nxt, x.ndim), x, out)

    def ddpm_step(self, x, curr, cond, cond_mask, noise):
        """ddpm_sample_step (discrete_diffusion.py:423-452): posterior mean of (x0 prediction, x) + sqrt(posterior variance) * noise;
        no noise at level 0; tokens at level -1 (clean) are kept.  Chosen when sampling_timesteps == timesteps (:395-420)."""
        t = self.tables
        kc = curr.clamp(min=0)
        _, x0, _ = self.predictions(x, kc, cond, cond_mask)
        mean = _ext(t.posterior_mean_coef1[kc], x.ndim) * x0 + _ext(t.posterior_mean_coef2[kc], x.ndim) * x
        nz = torch.where(_ext(kc > 0, x.ndim), noise, torch.zeros_like(noise)).clamp(-self.clip_noise, self.clip_noise)
        out = mean + torch.exp(0.5 * _ext(t.posterior_log_variance_clipped[kc], x.ndim)) * nz
        return torch.where(_ext(curr == -1, x.ndim), x, out)


@dataclass
class SamplerConfig:
    x_shape: Tuple[int, int, int] = (3, 256, 256)
    max_tokens: int = 8
    timesteps: int = 1000
    sampling_timesteps: int = 50
    clip_noise: float = 20.0
    scheduling_matrix: str = "full_sequence"
    is_full_sequence: bool = False
    prediction_guidance: Dict = field(default_factory=lambda: {"name": "conditional"})
    interpolation_guidance: Dict = field(default_factory=lambda: {"name": "conditional"})
    keyframe_density: Optional[float] = None
    sliding_context_len: Optional[int] = None
    interpolation_max_batch_size: Optional[int] = None
    reconstruction_guidance: float = 0.0  # dfot_video.py:700-723 (needs a differentiable model_fn)


class Sampler:
    """cond_fn maps raw conditions (B,T,16) -> processed (B,T,180,H,W) (ray encoding);
    it is re-applied each step on the NFE-replicated raw conditions, as the reference does."""

    def __init__(self, cfg: SamplerConfig, diffusion: Diffusion, cond_fn: CondFn, noise_fn: hg.NoiseFn):
        self.cfg, self.diff, self.cond_fn, self.noise_fn = cfg, diffusion, cond_fn, noise_fn
        self.trace: List[dict] = []

    # ------------------------------------------------------------------ one window
    def sample_sequence(self, batch_size: int, context: torch.Tensor, context_mask: torch.Tensor,
                        conditions: Optional[torch.Tensor], scheme: hg.Scheme,
                        length: Optional[int] = None) -> torch.Tensor:
        cfg = self.cfg
        length = context.shape[1] if length is None else length
        if length > cfg.max_tokens:
            raise ValueError(f"length is expected to <={cfg.max_tokens}, got {length}.")
        if context.shape[0] != batch_size:
            raise ValueError("context batch size mismatch")
        if context.shape[1] != length:
            raise ValueError("context length mismatch")
        if tuple(context.shape[2:]) != tuple(cfg.x_shape):
            raise ValueError("context shape not compatible with x_shape")
        if tuple(context.shape[:2]) != tuple(context_mask.shape):
            raise ValueError("context and context_mask must have the same shape.")
        horizon = cfg.max_tokens
        padding = horizon - length
        xs = self.noise_fn("init", (batch_size, horizon, *cfg.x_shape))
        if padding > 0:
            context = torch.cat([context, context.new_zeros(batch_size, padding, *cfg.x_shape)], 1)
            context_mask = torch.cat([context_mask, -torch.ones(batch_size, padding, dtype=torch.long)], 1)
        nd = xs.ndim
        xs = torch.where(_ext(context_mask, nd) >= 1, context, xs)
        sm = sch.scheduling_matrix(cfg.scheduling_matrix, horizon - padding, padding,
                                   cfg.timesteps, cfg.sampling_timesteps)
        sm = sm[:, None, :].repeat(1, batch_size, 1)
        if not cfg.is_full_sequence:
            sm = torch.where(context_mask[None] >= 1, -1, sm)
        changed = ~(sm[1:] == sm[:-1]).flatten(1).all(dim=1)
        sm = sm[int(torch.argmax(changed.float())):]
        self.trace.append({"context_mask": context_mask.clone(), "batch": batch_size, "rows": sm.shape[0]})
        for m in range(sm.shape[0] - 1):
            frm, to = sm[m], sm[m + 1]
            context_mask = torch.where((context_mask == 0) & (frm == -1), 2, context_mask)
            prev = xs.clone()
            g = hg.Guidance(scheme, context_mask)
            x_in, f_in, t_in, cmask = g.prepare(xs, frm, to, self.diff.q_sample, self.noise_fn,
                                                replacement_only=cfg.is_full_sequence)
            cond = None
            if conditions is not None:
                rep = conditions.repeat_interleave(g.nfe, dim=0)
                # dfot_video_pose.py:75-83: under temporal guidance the pose processing also sees the branch's noise levels
                cond = self.cond_fn(rep, f_in) if getattr(self, "cond_uses_levels", False) else self.cond_fn(rep)
            step_noise = self.noise_fn("ddim", tuple(x_in.shape))
            gfn = None
            if cfg.reconstruction_guidance > 0:
                # dfot_video.py:700-723: squared error of the predicted clean context against the given one, weighted by
                # sqrt(alphas_cumprod), summed over everything that is not "to be generated" (mask != 0: the reference's
                # `.bool()` also counts generated context and padding, whose `context` rows are zeros) and divided by that count
                cm = _ext(context_mask.bool(), nd).to(context.dtype)
                rg = float(cfg.reconstruction_guidance)

                def gfn(xk, pred_x0, alpha_cumprod, context=context, cm=cm, rg=rg):
                    loss = (pred_x0 - context) ** 2 * alpha_cumprod.sqrt()
                    loss = torch.sum(loss * cm / cm.sum(dim=1, keepdim=True).clamp(min=1))
                    return -rg * 0.5 * loss
            if self.diff.sampling_timesteps < self.diff.tables.timesteps:  # is_ddim_sampling (discrete_diffusion.py:108)
                x_out = self.diff.ddim_step(x_in, f_in, t_in, cond, cmask, step_noise, guidance_fn=gfn)
            else:
                x_out = self.diff.ddpm_step(x_in, f_in, cond, cmask, step_noise)
            xs = g.compose(x_out)
            xs = torch.where(_ext(context_mask, nd) == 0, xs, prev)
            if getattr(self, "step_hook", None) is not None:  # test instrumentation: the window state after step m
                self.step_hook(m, xs)
        return xs[:, :length] if padding > 0 else xs

    # ------------------------------------------------------------------ one window, refinement ladder
    def sample_sequence_refine(self, batch_size: int, goback_length: int, n_goback: int, context: torch.Tensor,
                               context_mask: torch.Tensor, conditions: Optional[torch.Tensor], scheme: hg.Scheme,
                               length: Optional[int] = None) -> torch.Tensor:
        """_sample_sequence_refine (dfot_video.py:765-1008) for one-branch guidance: a row whose last token (sample 0) moves
        down is a DDIM step (followed by a q_sample of the context whose result the reference discards -- the draw is kept so a
        replayed noise list stays aligned); every other row re-noises all tokens (q_sample_from_x_k, discrete_diffusion.py:252-260)."""
        cfg = self.cfg
        length = context.shape[1] if length is None else length
        horizon = cfg.max_tokens
        padding = horizon - length
        xs = self.noise_fn("init", (batch_size, horizon, *cfg.x_shape))
        if padding > 0:
            context = torch.cat([context, context.new_zeros(batch_size, padding, *cfg.x_shape)], 1)
            context_mask = torch.cat([context_mask, -torch.ones(batch_size, padding, dtype=torch.long)], 1)
        nd = xs.ndim
        xs = torch.where(_ext(context_mask, nd) >= 1, context, xs)
        sm = sch.refine_scheduling_matrix(horizon - padding, goback_length, n_goback, padding, cfg.timesteps, cfg.sampling_timesteps)
        sm = sm[:, None, :].repeat(1, batch_size, 1)
        sm = torch.where(context_mask[None] >= 1, -1, sm)
        ac = self.diff.tables.alphas_cumprod
        for m in range(sm.shape[0] - 1):
            frm, to = sm[m], sm[m + 1]
            if frm[0, -1].item() > to[0, -1].item():
                context_mask = torch.where((context_mask == 0) & (frm == -1), 2, context_mask)
                prev = xs.clone()
                g = hg.Guidance(scheme, context_mask)
                if g.nfe != 1:
                    raise ValueError("the reference's refinement step is only well-formed for one-branch guidance")
                x_in, f_in, t_in, cmask = g.prepare(xs, frm, to, self.diff.q_sample, self.noise_fn, replacement_only=cfg.is_full_sequence)
                cond = None if conditions is None else self.cond_fn(conditions)
                x_out = self.diff.ddim_step(x_in, f_in, t_in, cond, cmask, self.noise_fn("ddim", tuple(x_in.shape)))
                xs = g.compose(x_out)
                xc_t = self.diff.q_sample(context, t_in, self.noise_fn("refine_context", tuple(context.shape)))
                xs = torch.where(_ext(context_mask, nd) == 0, xs, xc_t)
                xs = torch.where(_ext(context_mask, nd) == 0, xs, prev)
            else:
                noise = self.noise_fn("renoise", tuple(xs.shape))
                scale = ac[to] / ac[frm]  # index -1 wraps to the last entry, as in the reference's gather
                scale = torch.where(to == cfg.timesteps - 1, torch.ones_like(scale), scale)
                xs = _ext(scale.sqrt(), nd) * xs + _ext((1 - scale).sqrt(), nd) * noise
        return xs[:, :length] if padding > 0 else xs

    # ------------------------------------------------------------------ sliding window
    def predict_sequence(self, context: torch.Tensor, length: int, conditions: Optional[torch.Tensor],
                         scheme: hg.Scheme, sliding_context_len: Optional[int]) -> torch.Tensor:
        cfg = self.cfg
        mt = cfg.max_tokens
        if sliding_context_len is None:
            if mt < length:
                raise ValueError("when length > max_tokens, sliding_context_len must be specified.")
            sliding_context_len = mt - 1
        if sliding_context_len == -1:
            sliding_context_len = mt - 1
        b, gt_len = context.shape[:2]
        if sliding_context_len < gt_len:
            raise ValueError("sliding_context_len is expected to be >= length of initial context")
        xs = context
        cur = gt_len
        while cur < length:
            c = min(sliding_context_len, cur)
            h = min(length - cur, mt - c)
            window = torch.cat([xs[:, -c:], xs.new_zeros(b, h, *cfg.x_shape)], 1)
            n_generated = cur - max(cur - c, gt_len)
            cmask = torch.ones(b, c, dtype=torch.long)
            if n_generated > 0:
                cmask[:, -n_generated:] = 2
            cmask = torch.cat([cmask, torch.zeros(b, h, dtype=torch.long)], 1)
            cond = None if conditions is None else conditions[:, cur - c: cur - c + mt]
            new = self.sample_sequence(b, window, cmask, cond, scheme, length=c + h)
            xs = torch.cat([xs, new[:, -h:]], 1)
            cur = xs.shape[1]
        return xs

    # ------------------------------------------------------------------ interpolation
    def interpolation_plan(self, known: torch.Tensor) -> List[List[torch.Tensor]]:
        """known (T,) bool -> list of stages, each a list of frame-index windows."""
        mt = self.cfg.max_tokens
        known = known.clone()
        plan: List[List[torch.Tensor]] = []
        while not bool(known.all()):
            keys = torch.where(known)[0].tolist()
            stage: List[torch.Tensor] = []
            chunk: Optional[List[int]] = None
            for left, right in zip(keys[:-1], keys[1:]):
                if chunk is not None:
                    if len(chunk) + right - left <= mt:
                        chunk = chunk + list(range(left + 1, right + 1))
                        continue
                    stage.append(torch.tensor(chunk))
                    chunk = None
                gap = right - left
                if gap == 1:
                    continue
                if gap >= mt - 1:
                    stage.append(torch.linspace(left, right, mt).round().long())
                else:
                    chunk = list(range(left, right + 1))
            if chunk is not None:
                stage.append(torch.tensor(chunk))
            for w in stage:
                known[w] = True
            plan.append(stage)
        return plan

    def _pad(self, y: torch.Tensor) -> torch.Tensor:
        mt = self.cfg.max_tokens
        if y.shape[1] >= mt:
            return y
        tail = y[:, -1:].expand(-1, mt - y.shape[1], *y.shape[2:])
        return torch.cat([y, tail], 1)

    def interpolate_videos(self, context: torch.Tensor, context_mask: Optional[torch.Tensor],
                           conditions: Optional[torch.Tensor]) -> torch.Tensor:
        cfg = self.cfg
        if context_mask is None:
            context_mask = torch.zeros(context.shape[:2], dtype=torch.bool)
            context_mask[:, [0, -1]] = True
        elif not bool(context_mask[:, [0, -1]].all()):
            raise AssertionError("The first and last frames must be known to interpolate.")
        scheme = hg.make_scheme(timesteps=cfg.timesteps, **cfg.interpolation_guidance)
        xs, known = context.clone(), context_mask.clone()
        for stage in self.interpolation_plan(context_mask[0]):
            ctx = torch.cat([self._pad(xs[:, w]) for w in stage], 0)
            msk = torch.cat([self._pad(known[:, w]) for w in stage], 0)
            cnd = None if conditions is None else torch.cat([self._pad(conditions[:, w]) for w in stage], 0)
            mb = cfg.interpolation_max_batch_size or ctx.shape[0]
            outs = []
            for i in range(0, ctx.shape[0], mb):
                sl = slice(i, i + mb)
                outs.append(self.sample_sequence(ctx[sl].shape[0], ctx[sl], msk[sl].long(),
                                                 None if cnd is None else cnd[sl], scheme))
            out = torch.cat(outs, 0)
            for w, pred in zip(stage, out.chunk(len(stage), 0)):
                xs[:, w] = pred[:, : len(w)]
                known[:, w] = True
        return xs

    # ------------------------------------------------------------------ top level
    def predict_videos(self, xs: torch.Tensor, n_context_tokens: int,
                       conditions: Optional[torch.Tensor]) -> torch.Tensor:
        cfg = self.cfg
        out = xs.clone()
        scheme = hg.make_scheme(timesteps=cfg.timesteps, **cfg.prediction_guidance)
        density = cfg.keyframe_density or 1
        if density > 1:
            raise ValueError("tasks.prediction.keyframe_density must be <= 1")
        n = out.shape[1]
        keys = torch.linspace(0, n - 1, round(density * n)).round().long()
        keys = torch.cat([torch.arange(n_context_tokens), keys]).unique()
        kc = None if conditions is None else conditions[:, keys]
        pred = self.predict_sequence(out[:, :n_context_tokens], len(keys), kc, scheme,
                                     cfg.sliding_context_len or cfg.max_tokens // 2)
        out[:, keys] = pred.to(out.dtype)
        if len(keys) < n:
            known = torch.zeros(out.shape[:2], dtype=torch.bool)
            known[:, keys] = True
            out = self.interpolate_videos(out, known, conditions)
        return out


def training_loss(model: ModelFn, x: torch.Tensor, cond: Optional[torch.Tensor], t: torch.Tensor,
                  noise: torch.Tensor, precond_scale: float = 0.125, shift: float = 0.125,
                  sigmoid_bias: float = -1.0):
    """Continuous-time v-prediction loss; returns (x_pred, per-element weighted loss)."""
    logsnr = sch.training_logsnr(t, shift)
    a = _ext(torch.sigmoid(logsnr).sqrt(), x.ndim)
    s = _ext(torch.sigmoid(-logsnr).sqrt(), x.ndim)
    x_t = a * x + s * noise
    v = model(x_t, precond_scale * logsnr, cond, None)
    eps_hat = a * v + s * x_t
    w = _ext(torch.sigmoid(sigmoid_bias - logsnr), x.ndim)
    return a * x_t - s * v, (eps_hat - noise) ** 2 * w


def discrete_loss_weights(tables: sch.ScheduleTables, k: torch.Tensor, strategy: str = "fused_min_snr", snr_clip: float = 5.0,
                          cum_snr_decay: float = 0.96, sigmoid_bias: float = -1.0, causal: bool = False) -> torch.Tensor:
    """DiscreteDiffusion.compute_loss_weights for objective pred_v (discrete_diffusion.py:274-343); k (B,T) long."""
    if strategy == "uniform":
        return torch.ones_like(k, dtype=torch.float32)
    snr_all = tables.snr  # float64 quotient cast to fp32 (discrete_diffusion.py:160-161)
    snr = snr_all[k]
    if strategy == "sigmoid":
        eps_w = torch.sigmoid(sigmoid_bias - torch.log(snr_all)[k])
    elif strategy == "min_snr":
        eps_w = snr_all.clamp(max=snr_clip)[k] / snr.clamp(min=1e-8)
    elif strategy == "fused_min_snr":
        nclip = snr_all.clamp(max=snr_clip)[k] / snr_clip
        nsnr = snr / snr_clip

        def running(x: torch.Tensor) -> torch.Tensor:  # exponential moving average of the PREVIOUS tokens, 0 for the first
            out = torch.zeros_like(x)
            acc = torch.zeros_like(x[:, 0])
            for t in range(x.shape[1]):
                out[:, t] = acc if t else 0.0
                acc = x[:, t] if t == 0 else cum_snr_decay * acc + (1 - cum_snr_decay) * x[:, t]
            return out
        cum = running(nclip) if causal else 0.5 * (running(nclip) + running(nclip.flip(1)).flip(1))
        clipped = (1 - (1 - cum * cum_snr_decay) * (1 - nclip)) * snr_clip
        snr = (1 - (1 - cum * cum_snr_decay) * (1 - nsnr)) * snr_clip
        eps_w = clipped / snr.clamp(min=1e-8)
    else:
        raise ValueError(f"unknown loss weighting strategy {strategy}")
    return eps_w * snr / (snr + 1)


def discrete_training_loss(model: ModelFn, tables: sch.ScheduleTables, x: torch.Tensor, k: torch.Tensor, noise: torch.Tensor,
                           **weighting):
    """DiscreteDiffusion.forward, objective pred_v (discrete_diffusion.py:345-377); returns (x_pred, per-element weighted loss)."""
    a = _ext(tables.sqrt_alphas_cumprod[k], x.ndim)
    s = _ext(tables.sqrt_one_minus_alphas_cumprod[k], x.ndim)
    x_k = a * x + s * noise
    v = model(x_k, k, None, None)
    w = _ext(discrete_loss_weights(tables, k, **weighting), x.ndim)
    return a * x_k - s * v, (v - (a * noise - s * x)) ** 2 * w
