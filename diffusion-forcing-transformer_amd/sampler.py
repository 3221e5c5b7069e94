"""DFoT sampler driver on the MI355X engine (host side).

Same method names, arguments and error behaviour as the reference's sampling path:
  * ``_process_conditions``  algorithms/dfot/dfot_video_pose.py:64-110   -> dfot_ray_encode
  * ``_sample_sequence``     algorithms/dfot/dfot_video.py:516-763       -> dfot_hg_prepare /
                             backbone / dfot_ddim_compose per step
  * ``_predict_sequence``    dfot_video.py:362-514   (sliding window)
  * ``_interpolate_videos``  dfot_video.py:181-360   (planner + batched windows)
  * ``_predict_videos``      dfot_video.py:114-179   (keyframes then interpolation)
  * ``_pad_to_max_tokens``   algorithms/common/base_pytorch_video_algo.py:666-682
Differences by design: the per-step History-Guidance bookkeeping is planned on the host in numpy
(no device syncs inside the step loop), the camera-ray encoding of a window is computed once per
window instead of once per step (it does not depend on the step), and all frame arithmetic of a
step is two fused kernels around the backbone call.
Randomness is drawn through ``noise_fn(tag, shape)`` (tags: "init", "q_sample", "ddim") so tests can
replay the reference's draws; the default draws on the GPU with ``torch.randn``.
By default (``use_graph``) the steps of a window after the first are captured as ONE hipGraph (``torch.cuda.CUDAGraph``) and replayed;
windows the capture cannot serve run eagerly.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import capi, parallel
from .diffusion import DiffusionConfig, Schedule
from .guidance import HistoryGuidance


@dataclass
class SamplerConfig:
    x_shape: Tuple[int, int, int] = (3, 256, 256)
    max_tokens: int = 8
    diffusion: DiffusionConfig = field(default_factory=DiffusionConfig)
    scheduling_matrix: str = "full_sequence"
    is_full_sequence: bool = False
    prediction_guidance: Dict = field(default_factory=lambda: {"name": "conditional"})
    interpolation_guidance: Dict = field(default_factory=lambda: {"name": "conditional"})
    keyframe_density: Optional[float] = None
    sliding_context_len: Optional[int] = None
    interpolation_max_batch_size: Optional[int] = None
    # refinement_sampling of dfot_video.yaml:41-44 ({enabled, goback_length, n_goback}); honoured by the difference sampler only,
    # as in the reference (difference_dfot_video.py:433-440, 576-586)
    refinement_sampling: Optional[Dict] = None
    # camera_pose_conditioning of dfot_video_pose.yaml:7-9
    camera_pose_normalize_by: str = "first"
    camera_pose_bound: Optional[float] = None


NoiseFn = Callable[[str, tuple], torch.Tensor]


def device_noise_fn(generator: Optional[torch.Generator] = None, clip: float = 20.0) -> NoiseFn:
    def fn(tag: str, shape: tuple) -> torch.Tensor:
        return torch.randn(shape, device="cuda", generator=generator).clamp_(-clip, clip)
    return fn


class DFoTVideoPoseSampler:
    def __init__(self, cfg: SamplerConfig, backbone, noise_fn: Optional[NoiseFn] = None):
        self.cfg = cfg
        self.model = backbone
        self.schedule = Schedule(cfg.diffusion)
        self.noise_fn = noise_fn or device_noise_fn(clip=cfg.diffusion.clip_noise)
        self.timesteps = cfg.diffusion.timesteps
        self.sampling_timesteps = cfg.diffusion.sampling_timesteps
        self.max_tokens = cfg.max_tokens
        self.x_shape = tuple(cfg.x_shape)
        self.trace: List[dict] = []
        self.window_forwards = 0
        self.shard_windows = False  # True: shard interpolation windows over torch.distributed ranks (parallel.py)
        # True: the History-Guidance branches of the (sequential, replicated) key-frame windows are split over ranks -- rank r runs
        # branch r % NFE of every sample and one all-gather per step returns all branches (parallel.exchange_branches)
        self.branch_parallel = False
        self._branch_split_active = False
        self.device = "cuda"        # where the rollout state lives; "cpu" only together with dry_run (planner inspection / host tests)
        self.dry_run = False        # True: plan every window (trace, noise draws in the reference's order) but launch nothing
        # hipGraph execution of the step loop (the default): the steps of a window after the first are ONE captured graph, one replay per
        # window (_run_steps_graph).  Windows the capture cannot serve -- stochastic steps, strict-order noise replay, per-step
        # conditioning, branch-parallel key frames, reconstruction guidance, a step_hook -- run the same kernels eagerly; False forces that.
        self.use_graph = True
        self.skip_frozen_frames = True  # ... nor the down path of frames whose input equals the previous step's (clean context)
        self.skip_dead_frames = True  # the backbone does not compute the output of tokens the composition step ignores (context / padding)
        self.graph_replays = 0
        self.graph_captures = 0
        self._graphs: Dict[tuple, dict] = {}
        self._graphs_generation = 0
        if cfg.diffusion.sampling_timesteps > cfg.diffusion.timesteps:
            raise ValueError("sampling_timesteps must be <= timesteps")

    # ------------------------------------------------------------------ conditions
    @torch.no_grad()
    def _process_conditions(self, conditions: Optional[torch.Tensor], noise_levels=None) -> Optional[torch.Tensor]:
        """raw poses (B,T,16) -> ray encoding (B,T,180,H,W).  ``noise_levels`` (B,T) is only looked at where the reference looks
        at it: under temporal History Guidance the poses of tokens at pure noise are replaced by interpolated ones."""
        if conditions is None:
            return None
        if conditions.shape[-1] != 16:
            raise ValueError(f"raw camera poses must have 16 values per frame, got {conditions.shape[-1]}")
        cfg = self.cfg
        if cfg.camera_pose_normalize_by not in ("first", "mean"):
            raise ValueError(f"Unknown camera pose normalization method: {cfg.camera_pose_normalize_by}")
        interp = None
        if noise_levels is not None and getattr(self, "_interpolate_masked_poses", False):
            interp = np.asarray(noise_levels.detach().cpu().numpy() if torch.is_tensor(noise_levels) else noise_levels) == self.timesteps - 1
        if cfg.camera_pose_normalize_by == "first" and cfg.camera_pose_bound is None and interp is None:
            return torch.ops.dfot.ray_encoding(conditions, int(self.x_shape[-1]))
        from . import pose
        world = pose.normalize_poses(conditions.detach().float().cpu().numpy(), cfg.camera_pose_normalize_by, cfg.camera_pose_bound, interp)
        return torch.ops.dfot.ray_encoding(torch.from_numpy(world), int(self.x_shape[-1]), True)

    # ------------------------------------------------------------------ denoising loss (no backward)
    @torch.no_grad()
    def denoising_loss(self, xs: torch.Tensor, conditions: Optional[torch.Tensor], t: torch.Tensor,
                       noise: Optional[torch.Tensor] = None, masks: Optional[torch.Tensor] = None):
        """One noised forward + sigmoid-weighted v-prediction loss: ``ContinuousDiffusion.forward``
        (diffusion/continuous_diffusion.py:140-167) followed by ``_reweight_loss``
        (algorithms/common/base_pytorch_video_algo.py:684-693) -- what ``training_step`` and the validation
        denoising loss evaluate (dfot_video.py:41-75).  t: (B,T) in [0,1] per-token noise levels.
        Returns (x_pred, loss scalar, per-token loss (B,T))."""
        b, tk = xs.shape[:2]
        f = int(np.prod(xs.shape[2:]))
        # cosine logSNR schedule of the reference in fp32 (CosineNoiseSchedule, continuous_diffusion.py:46-92), limits / shift / loss
        # weighting from the DiffusionConfig this sampler was built with (the same object the sampling Schedule honours)
        logsnr, alpha, sigma, weight = self.cfg.diffusion.training_logsnr_tables(t)
        tab = torch.stack([alpha, sigma, weight, self.cfg.diffusion.precond_scale * logsnr]).float().cuda().contiguous()
        x = xs.to(device="cuda", dtype=torch.float32).contiguous()
        if noise is None:
            noise = self.noise_fn("train", tuple(x.shape))
        eps = noise.to(device="cuda", dtype=torch.float32).clamp(-self.cfg.diffusion.clip_noise, self.cfg.diffusion.clip_noise).contiguous()
        ones = torch.ones(b, tk, device="cuda")
        x_t = torch.empty_like(x)
        capi.check(capi.lib.dfot_hg_prepare(capi.ptr(x), capi.ptr(eps), capi.ptr(tab[0]), capi.ptr(tab[1]), capi.ptr(x_t),
                                            b, 1, tk, f, capi.stream_ptr()))
        v = self.model(x_t, tab[3], self._process_conditions(conditions), None)
        x_pred = torch.empty_like(x)
        per_token = torch.empty(b, tk, device="cuda")
        scratch = torch.empty(int(capi.lib.dfot_vpred_loss_scratch_floats(b, tk, f)), device="cuda")
        capi.check(capi.lib.dfot_vpred_loss(capi.ptr(x), capi.ptr(eps), capi.ptr(v), capi.ptr(tab[0]), capi.ptr(tab[1]),
                                            capi.ptr(tab[2]), capi.ptr(x_pred), capi.ptr(scratch), capi.ptr(per_token), b, tk, f,
                                            capi.stream_ptr()))
        del ones
        if masks is not None:
            per_token = per_token * masks.to(device="cuda", dtype=torch.float32).view(b, tk)
        return x_pred, per_token.mean(), per_token

    @torch.no_grad()
    def discrete_denoising_loss(self, xs: torch.Tensor, k: torch.Tensor, noise: Optional[torch.Tensor] = None,
                                masks: Optional[torch.Tensor] = None, loss_weighting: Optional[Dict] = None):
        """``DiscreteDiffusion.forward`` (diffusion/discrete_diffusion.py:345-377, objective pred_v) + ``_reweight_loss``: noise
        every token to its integer level k (B,T), one backbone forward, weighted v-space squared error (weights:
        ``Schedule.loss_weights``, default = dfot_video.yaml's fused_min_snr / snr_clip 5 / cum_snr_decay 0.9).  No backward.
        Returns (x_pred, loss scalar, per-token loss (B,T))."""
        if self.cfg.diffusion.is_continuous:
            raise ValueError("discrete_denoising_loss needs DiffusionConfig(is_continuous=False)")
        b, tk = xs.shape[:2]
        f = int(np.prod(xs.shape[2:]))
        kk = k.detach().cpu().numpy().astype(np.int64)
        sch = self.schedule
        tab = np.stack([sch.sqrt_alphas_cumprod[kk], sch.sqrt_one_minus_alphas_cumprod[kk],
                        sch.loss_weights(kk, **(loss_weighting or {}))]).astype(np.float32)
        tab = torch.from_numpy(tab).cuda().contiguous()
        x = xs.to(device="cuda", dtype=torch.float32).contiguous()
        if noise is None:
            noise = self.noise_fn("train", tuple(x.shape))
        eps = noise.to(device="cuda", dtype=torch.float32).clamp(-self.cfg.diffusion.clip_noise, self.cfg.diffusion.clip_noise).contiguous()
        x_k = torch.empty_like(x)
        capi.check(capi.lib.dfot_hg_prepare(capi.ptr(x), capi.ptr(eps), capi.ptr(tab[0]), capi.ptr(tab[1]), capi.ptr(x_k),
                                            b, 1, tk, f, capi.stream_ptr()))
        v = self.model(x_k, k.to(device="cuda", dtype=torch.int32), None, None)
        x_pred = torch.empty_like(x)
        per_token = torch.empty(b, tk, device="cuda")
        scratch = torch.empty(int(capi.lib.dfot_vpred_loss_scratch_floats(b, tk, f)), device="cuda")
        capi.check(capi.lib.dfot_vspace_loss(capi.ptr(x), capi.ptr(eps), capi.ptr(v), capi.ptr(tab[0]), capi.ptr(tab[1]),
                                             capi.ptr(tab[2]), capi.ptr(x_pred), capi.ptr(scratch), capi.ptr(per_token), b, tk, f,
                                             capi.stream_ptr()))
        if masks is not None:
            per_token = per_token * masks.to(device="cuda", dtype=torch.float32).view(b, tk)
        return x_pred, per_token.mean(), per_token

    # data (un)normalisation of the reference (algorithms/common/base_pytorch_video_algo.py:491-502)
    def _normalize_x(self, xs: torch.Tensor, mean, std) -> torch.Tensor:
        m = torch.as_tensor(mean, dtype=xs.dtype, device=xs.device).view(-1, 1, 1)
        s = torch.as_tensor(std, dtype=xs.dtype, device=xs.device).view(-1, 1, 1)
        return (xs - m) / s

    def _unnormalize_x(self, xs: torch.Tensor, mean, std) -> torch.Tensor:
        m = torch.as_tensor(mean, dtype=xs.dtype, device=xs.device).view(-1, 1, 1)
        s = torch.as_tensor(std, dtype=xs.dtype, device=xs.device).view(-1, 1, 1)
        return xs * s + m

    def _pad_to_max_tokens(self, y: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if y is None or y.shape[1] >= self.max_tokens:
            return y
        tail = y[:, -1:].expand(-1, self.max_tokens - y.shape[1], *y.shape[2:])
        return torch.cat([y, tail], dim=1)

    # ------------------------------------------------------------------ one window
    @torch.no_grad()
    def _sample_sequence(self, batch_size: int, length: Optional[int] = None, context: Optional[torch.Tensor] = None,
                         context_mask: Optional[torch.Tensor] = None, conditions: Optional[torch.Tensor] = None,
                         history_guidance: Optional[HistoryGuidance] = None, _refine: Optional[Tuple[int, int]] = None,
                         **_) -> Tuple[torch.Tensor, None]:
        cfg, sch = self.cfg, self.schedule
        x_shape = self.x_shape
        # temporal History Guidance + camera poses: the poses of tokens shown as pure noise are re-interpolated, per branch and
        # per step (dfot_video_pose.py:75-83) -- one encoding per distinct mask pattern, cached for the window
        pose_interp = conditions is not None and history_guidance is not None and history_guidance.needs_pose_interpolation
        if length is None:
            length = self.max_tokens if context is None else context.shape[1]
        if length > self.max_tokens:
            raise ValueError(f"length is expected to <={self.max_tokens}, got {length}.")
        if context is not None:
            if context_mask is None:
                raise ValueError("context_mask must be provided if context is given.")
            if context.shape[0] != batch_size:
                raise ValueError(f"context batch size is expected to be {batch_size} but got {context.shape[0]}.")
            if context.shape[1] != length:
                raise ValueError(f"context length is expected to be {length} but got {context.shape[1]}.")
            if tuple(context.shape[2:]) != tuple(x_shape):
                raise ValueError(f"context shape not compatible with x_stacked_shape {x_shape}.")
        if context_mask is not None:
            if context is None:
                raise ValueError("context must be provided if context_mask is given. ")
            if tuple(context.shape[:2]) != tuple(context_mask.shape):
                raise ValueError("context and context_mask must have the same shape.")
        horizon = self.max_tokens
        padding = horizon - length
        f = int(np.prod(x_shape))
        dev = self.device
        if dev != "cuda" and not self.dry_run:
            raise RuntimeError("the sampler runs on the GPU; device='cpu' is only valid together with dry_run (no CPU fallback)")
        xs = self.noise_fn("init", (batch_size, horizon, *x_shape)).to(device=dev, dtype=torch.float32)
        xs = xs.clamp(-cfg.diffusion.clip_noise, cfg.diffusion.clip_noise)
        if context is None:
            mask = np.zeros((batch_size, horizon), np.int64)
        else:
            mask = context_mask.detach().cpu().numpy().astype(np.int64)
            ctx = context.to(device=dev, dtype=torch.float32)
            if padding > 0:
                ctx = torch.cat([ctx, ctx.new_zeros(batch_size, padding, *x_shape)], 1)
                mask = np.concatenate([mask, -np.ones((batch_size, padding), np.int64)], 1)
            keep = torch.from_numpy(mask >= 1).to(dev).view(batch_size, horizon, 1, 1, 1)
            xs = torch.where(keep, ctx, xs)
        rg = float(getattr(cfg.diffusion, "reconstruction_guidance", 0.0) or 0.0)
        if rg > 0 and not sch.is_ddim_sampling:
            raise ValueError("reconstruction guidance is a DDIM-step feature (discrete_diffusion.py:455-513); sampling_timesteps == timesteps selects DDPM")
        rg_ctx = (ctx if context is not None else torch.zeros_like(xs)) if rg > 0 and not self.dry_run else None
        if history_guidance is None:
            history_guidance = HistoryGuidance.conditional(timesteps=self.timesteps)

        if _refine is None:
            sm = sch.scheduling_matrix(cfg.scheduling_matrix, horizon - padding, padding)
            sm = np.repeat(sm[:, None, :], batch_size, axis=1)
            if not cfg.is_full_sequence:
                sm = np.where(mask[None] >= 1, -1, sm)
            changed = ~(sm[1:] == sm[:-1]).reshape(sm.shape[0] - 1, -1).all(axis=1)
            sm = sm[int(np.argmax(changed)):]
        else:  # refinement ladder: context tokens always at -1, no pruning of leading rows (dfot_video.py:881-890)
            if cfg.scheduling_matrix != "full_sequence":
                raise ValueError("Refining only support full_sequence scheduling matrix")
            sm = sch.refine_scheduling_matrix(horizon - padding, _refine[0], _refine[1], padding)
            sm = np.repeat(sm[:, None, :], batch_size, axis=1)
            sm = np.where(mask[None] >= 1, -1, sm)
        self.trace.append({"context_mask": mask.copy(), "batch": batch_size, "rows": sm.shape[0]})

        # ---- plan every step on the host first and upload the coefficient tables ONCE: the step loop below then
        # contains no host<->device synchronisation, so the CPU runs ahead of the GPU and launch latency stays hidden
        plans = []
        for m in range(sm.shape[0] - 1):
            frm, to = sm[m], sm[m + 1]
            if _refine is not None and not frm[0, -1] > to[0, -1]:
                # going back up the ladder: every token is re-noised from its level to the next one,
                # q_sample_from_x_k (discrete_diffusion.py:252-260); index -1 (clean tokens) wraps to the last table entry
                # exactly as the reference's gather does, which makes their scale 1
                ac = sch.alphas_cumprod.astype(np.float32)
                with np.errstate(divide="ignore", invalid="ignore"):
                    scale = np.where(to == self.timesteps - 1, np.float32(1), ac[to] / ac[frm]).astype(np.float32)
                if not (np.isfinite(scale).all() and (scale <= 1).all()):
                    # 0/0 when the schedule ends at alphas_cumprod = 0 (K600 cosine) and the window has context tokens; scale > 1
                    # when a DESCENDING row is taken for a re-noising row (last token of sample 0 is context or padding)
                    raise ValueError("refinement sampling: this window / schedule makes q_sample_from_x_k produce NaN in the reference "
                                     "(alphas_cumprod ratio not in [0, 1]); refused instead of returning NaN")
                tables = np.zeros((8, batch_size, horizon), np.float32)
                tables[0], tables[1] = np.sqrt(scale), np.sqrt(np.float32(1) - scale)
                plans.append(dict(renoise=True, nfe=1, bm=batch_size, tables=tables, gen=(mask == 0).astype(np.uint8), cmask=None))
                continue
            mask = np.where((mask == 0) & (frm == -1), 2, mask)
            plan = history_guidance.plan(mask, frm, to, replacement_only=cfg.is_full_sequence)
            bm = batch_size * plan.nfe
            lv = plan.levels.reshape(bm, horizon)
            tl = plan.to_levels.reshape(bm, horizon)
            repl = plan.replace.reshape(bm, horizon)
            qa_c, qb_c = sch.q_sample_coef(lv)
            qa = np.where(repl, qa_c, np.float32(1)).astype(np.float32)
            qb = np.where(repl, qb_c, np.float32(0)).astype(np.float32)
            excl = None
            if plan.excluded is not None:  # gen tokens outside the branch's gen segment: x_in = fresh (unclamped) noise
                excl = plan.excluded.reshape(bm, horizon)
                qa, qb = np.where(excl, np.float32(0), qa), np.where(excl, np.float32(1), qb)
            # deterministic DDIM (eta = 0, the BASELINE configs), stochastic DDIM (eta > 0) or DDPM (sampling_timesteps == timesteps):
            # the same fused step with other coefficients; the stochastic ones add sigma * noise per branch (dfot_ddim_noise)
            sa, s1, an, cn, keepf, sigma = sch.ddim_coef(lv, tl) if sch.is_ddim_sampling else sch.ddpm_coef(lv)
            sigma = np.where(keepf != 0, np.float32(0), sigma).astype(np.float32)
            tables = np.stack([qa, qb, sa, s1, an, cn, keepf, sch.model_level(lv)]).astype(np.float32)
            plans.append(dict(plan=plan, nfe=plan.nfe, bm=bm, need_noise=bool(repl.any()) or excl is not None, excl=excl, tables=tables,
                              sigma=sigma if bool((sigma != 0).any()) else None,
                              gen=(mask == 0).astype(np.uint8),
                              cmask=None if plan.cond_masked is None else np.tile(plan.cond_masked, batch_size)))
        if not plans:
            return (xs[:, :-padding] if padding > 0 else xs), None
        strict = bool(getattr(self.noise_fn, "strict_order", False))

        def draw_noise(p_):
            """noise for re-noised history tokens; with a strict-order noise source (golden replay) every draw the
            reference makes is consumed, used or not (history_guidance.py:505,530; discrete_diffusion.py:525)"""
            nfe, bm, need = p_["nfe"], p_["bm"], p_["need_noise"]
            noise = None
            if history_guidance.is_simple:
                if nfe == 2 and (need or strict):
                    drawn = self.noise_fn("q_sample", (batch_size, horizon, *x_shape))
                    noise = torch.zeros(batch_size, 2, horizon, *x_shape, device=dev, dtype=torch.float32)
                    noise[:, 0] = drawn.to(device=dev, dtype=torch.float32)
            else:
                g = p_["plan"].n_gen
                bh = bm // g  # the reference draws the history noise per (sample, history branch), before the gen-segment split
                if need or strict:
                    noise = self.noise_fn("q_sample", (bh, horizon, *x_shape)).to(device=dev, dtype=torch.float32)
                    if g > 1:
                        noise = noise.repeat_interleave(g, dim=0)
                if strict or p_["excl"] is not None:
                    fresh = self.noise_fn("excluded", (bh, g, horizon, *x_shape))
                    if p_["excl"] is not None:  # torch.randn_like(x) for the excluded gen tokens (:529-533), not clamped
                        fresh = fresh.to(device=dev, dtype=torch.float32).reshape(bm, horizon, *x_shape)
                        em = torch.from_numpy(p_["excl"]).to(dev).view(bm, horizon, *([1] * len(x_shape)))
                        noise = torch.where(em, fresh, noise if noise is not None else torch.zeros_like(fresh))
            return None if noise is None else noise.contiguous().view(bm, horizon, *x_shape)

        # History-Guidance branch parallelism (replicated key-frame windows only): a step's `nfe` branches are split over nfe-rank
        # sub-groups when the world divides into them; otherwise EVERY rank evaluates all branches of that step (same model batch on
        # every rank, so the replicas stay bit-identical).  The sub-groups are created here, before the step loop, by every rank of
        # the world (dist.new_group is a world-wide call): all ranks plan the same window, hence the same set of branch counts.
        split_nfe = set()
        if self._branch_split_active:
            world_n = parallel.world_info()[0]
            for n_ in sorted({p_["nfe"] for p_ in plans if not p_.get("renoise")}):
                if n_ > 1 and world_n > 1 and world_n % n_ == 0:
                    parallel.branch_group(n_)
                    split_nfe.add(n_)
        if self.dry_run:
            # planner inspection: every host decision of the window has been taken (trace, per-step plans); consume the noise
            # draws in the order the device path would and hand back the context-filled window -- nothing is launched
            for p_ in plans:
                if p_["nfe"] in split_nfe:  # host plumbing of the branch exchange: same groups, same collective, a marker payload
                    hb = parallel.world_info()[1] % p_["nfe"]
                    got = parallel.exchange_branches(torch.full((batch_size, 1), float(hb)), p_["nfe"])
                    if not torch.equal(got.view(batch_size, p_["nfe"]), torch.arange(p_["nfe"], dtype=got.dtype).repeat(batch_size, 1)):
                        raise RuntimeError("branch exchange returned the branches out of (sample, branch) order")
                    self.branch_exchanges = getattr(self, "branch_exchanges", 0) + 1
                if p_.get("renoise"):
                    self.noise_fn("renoise", (batch_size, horizon, *x_shape))
                    continue
                draw_noise(p_)
                if strict or p_["sigma"] is not None:
                    self.noise_fn("ddim", (p_["bm"], horizon, *x_shape))
                    if _refine is not None:
                        self.noise_fn("refine_context", (batch_size, horizon, *x_shape))
            self.window_forwards += sum(p_["bm"] for p_ in plans if not p_.get("renoise"))
            return (xs[:, :-padding] if padding > 0 else xs), None
        flat_dev = torch.from_numpy(np.concatenate([p_["tables"].ravel() for p_ in plans])).cuda()
        gens_dev = torch.from_numpy(np.stack([p_["gen"] for p_ in plans])).cuda()
        off = 0
        cmask_cache: Dict[bytes, torch.Tensor] = {}
        weight_cache: Dict[bytes, torch.Tensor] = {}
        lives_by_nfe: Dict[int, torch.Tensor] = {}
        for i, p_ in enumerate(plans):
            n = p_["tables"].size
            p_["tables_dev"] = flat_dev[off:off + n].view(8, p_["bm"], horizon)
            off += n
            p_["gen_dev"] = gens_dev[i]
            if p_.get("renoise"):
                continue
            if p_["nfe"] not in lives_by_nfe:  # one expansion per branch count of the window, not one per step
                lives_by_nfe[p_["nfe"]] = gens_dev if p_["nfe"] == 1 else gens_dev.repeat_interleave(p_["nfe"], dim=1)
            p_["live_dev"] = lives_by_nfe[p_["nfe"]][i]
            wsrc = p_["plan"].weights if p_["plan"].tok_weights is None else p_["plan"].tok_weights
            wkey = wsrc.tobytes()
            if wkey not in weight_cache:
                weight_cache[wkey] = torch.from_numpy(np.ascontiguousarray(wsrc, dtype=np.float32)).cuda()
            p_["weights_dev"] = weight_cache[wkey]
            p_["tokw"] = p_["plan"].tok_weights is not None
            p_["sigma_dev"] = None if p_["sigma"] is None else torch.from_numpy(p_["sigma"]).cuda()
            p_["cmask_dev"] = None
            if p_["cmask"] is not None:
                # one device tensor per distinct mask pattern: the backbone keys its per-window pose caches on the
                # identity of (external_cond, external_cond_mask)
                ckey = p_["cmask"].tobytes()
                if ckey not in cmask_cache:
                    cmask_cache[ckey] = torch.from_numpy(p_["cmask"]).cuda()
                p_["cmask_dev"] = cmask_cache[ckey]

        cond_full = None if pose_interp else self._process_conditions(conditions)
        cond_rep, cond_nfe = None, 0
        if pose_interp:
            by_mask: Dict[bytes, torch.Tensor] = {}
            self._interpolate_masked_poses = True
            try:
                for p_ in plans:
                    if p_.get("renoise"):
                        continue
                    lv = p_["plan"].levels.reshape(p_["bm"], horizon)
                    key = (lv == self.timesteps - 1).tobytes() + bytes([p_["nfe"]])
                    if key not in by_mask:
                        by_mask[key] = self._process_conditions(conditions.repeat_interleave(p_["nfe"], dim=0), lv)
                    p_["cond"] = by_mask[key]
            finally:
                self._interpolate_masked_poses = False
        fresh_dev = torch.from_numpy(np.concatenate([a.ravel() for a in self._fresh_flags(plans, batch_size, horizon)])).cuda()
        foff = 0
        for p_ in plans:
            p_["fresh_dev"] = fresh_dev[foff:foff + p_["bm"] * horizon].view(p_["bm"], horizon)
            foff += p_["bm"] * horizon
        xs = xs.contiguous()
        s = capi.stream_ptr
        branch_cache: Dict[tuple, tuple] = {}

        # frames whose model output the composition never reads (context / padding tokens: gen == 0): the U-ViT backbone skips them past
        # its last transformer block (backbone.live_frames).  A backbone without the attribute (DiT3D) computes everything.
        skip_dead = self.skip_dead_frames and hasattr(self.model, "live_frames")
        # ... and frames whose backbone input did not change since the previous step (clean context of the conditional branch): their
        # frame-local down-path activations are still in the backbone's workspace (backbone.fresh_frames, _fresh_flags below)
        skip_frozen = skip_dead and self.skip_frozen_frames and hasattr(self.model, "fresh_frames")

        def step(p_, xs, noise, tables, gen_dev, xs_next=None, live_dev=None, fresh_dev=None):
            nonlocal cond_rep, cond_nfe
            nfe, bm = p_["nfe"], p_["bm"]
            x_in = torch.empty(bm, horizon, *x_shape, device="cuda", dtype=torch.float32)
            capi.check(capi.lib.dfot_hg_prepare(capi.ptr(xs), capi.ptr(noise), capi.ptr(tables[0]), capi.ptr(tables[1]),
                                                capi.ptr(x_in), batch_size, nfe, horizon, f, s()))
            if cond_full is not None and cond_nfe != nfe:
                cond_rep = cond_full if nfe == 1 else cond_full.repeat_interleave(nfe, dim=0)
                cond_nfe = nfe
            if "cond" in p_:
                cond_rep = p_["cond"]
            # discrete diffusion hands the backbone integer level indices (exact in the float32 table)
            lvl = tables[7] if cfg.diffusion.is_continuous else tables[7].to(torch.int32)
            world, rank = parallel.world_info()
            if nfe in split_nfe:  # this rank evaluates branch rank % nfe; its nfe-rank sub-group returns all of them
                # this rank's branch only; the conditioning slices are cached per window so that the backbone's pose cache (keyed on
                # tensor identity) still hits on every step
                hb = rank % nfe
                key = (id(cond_rep), id(p_["cmask_dev"]), hb)
                if key not in branch_cache:
                    c_h = None if cond_rep is None else cond_rep.view(batch_size, nfe, *cond_rep.shape[1:])[:, hb].contiguous()
                    m_h = None if p_["cmask_dev"] is None else p_["cmask_dev"].view(batch_size, nfe)[:, hb].contiguous()
                    branch_cache[key] = (c_h, m_h, cond_rep, p_["cmask_dev"])  # keep the keyed tensors alive
                c_h, m_h = branch_cache[key][:2]
                x_h = x_in.view(batch_size, nfe, horizon, *x_shape)[:, hb].contiguous()
                l_h = lvl.view(batch_size, nfe, horizon)[:, hb].contiguous()
                if skip_dead:
                    self.model.live_frames = gen_dev
                try:
                    v = parallel.exchange_branches(self.model(x_h, l_h, c_h, m_h), nfe)
                    self.branch_exchanges = getattr(self, "branch_exchanges", 0) + 1
                finally:
                    if skip_dead:
                        self.model.live_frames = None
            elif rg > 0:
                v = self._reconstruction_guided_v(x_in, lvl, cond_rep, p_["cmask_dev"], tables, gen_dev, rg_ctx, rg, nfe)
            else:
                if skip_dead:  # rows of the model batch are (sample, branch): every branch of a sample shares the sample's flags
                    self.model.live_frames = live_dev if live_dev is not None else p_["live_dev"]
                if skip_frozen:
                    self.model.fresh_frames = fresh_dev if fresh_dev is not None else p_["fresh_dev"]
                try:
                    v = self.model(x_in, lvl, cond_rep, p_["cmask_dev"])
                finally:
                    if skip_dead:
                        self.model.live_frames = None
                    if skip_frozen:
                        self.model.fresh_frames = None
            step_noise = None
            if strict or p_["sigma"] is not None:  # the reference draws it every step; with sigma = 0 it is multiplied by 0
                step_noise = self.noise_fn("ddim", (bm, horizon, *x_shape))
            if xs_next is None:
                xs_next = torch.empty_like(xs)
            compose = capi.lib.dfot_ddim_compose_tokw if p_["tokw"] else capi.lib.dfot_ddim_compose
            capi.check(compose(capi.ptr(xs), capi.ptr(x_in), capi.ptr(v), capi.ptr(tables[2]),
                                                  capi.ptr(tables[3]), capi.ptr(tables[4]), capi.ptr(tables[5]),
                                                  capi.ptr(tables[6]), capi.ptr(p_["weights_dev"]), capi.ptr(gen_dev),
                                                  capi.ptr(xs_next), batch_size, nfe, horizon, f, s()))
            if p_["sigma"] is not None:
                nz = step_noise.to(device="cuda", dtype=torch.float32).clamp(-cfg.diffusion.clip_noise, cfg.diffusion.clip_noise).contiguous()
                capi.check(capi.lib.dfot_ddim_noise(capi.ptr(nz), capi.ptr(p_["sigma_dev"]), capi.ptr(p_["weights_dev"]), capi.ptr(gen_dev),
                                                    capi.ptr(xs_next), batch_size, nfe, horizon, f, int(p_["tokw"]), s()))
            return xs_next

        def renoise(p_, xs):
            noise = self.noise_fn("renoise", (batch_size, horizon, *x_shape)).to(device="cuda", dtype=torch.float32)
            noise = noise.clamp(-cfg.diffusion.clip_noise, cfg.diffusion.clip_noise).contiguous()
            out = torch.empty_like(xs)
            capi.check(capi.lib.dfot_hg_prepare(capi.ptr(xs), capi.ptr(noise), capi.ptr(p_["tables_dev"][0]), capi.ptr(p_["tables_dev"][1]),
                                                capi.ptr(out), batch_size, 1, horizon, f, s()))
            return out

        if _refine is not None:
            for p_ in plans:
                if p_.get("renoise"):
                    xs = renoise(p_, xs)
                    continue
                xs = step(p_, xs, draw_noise(p_), p_["tables_dev"], p_["gen_dev"])
                if strict:  # the reference re-noises the context here (q_sample) and then discards it (dfot_video.py:984-992)
                    self.noise_fn("refine_context", (batch_size, horizon, *x_shape))
            self.window_forwards += sum(p_["bm"] for p_ in plans if not p_.get("renoise"))
            return (xs[:, :-padding] if padding > 0 else xs), None
        # one captured step body serves every step only if nothing but the tables changes: same branch batch, conditioning mask,
        # composition weights AND conditioning tensor (temporal guidance re-interpolates the poses of pure-noise tokens per step)
        uniform = all(p_["bm"] == plans[0]["bm"] and p_["cmask_dev"] is plans[0]["cmask_dev"]
                      and p_["weights_dev"] is plans[0]["weights_dev"] and p_.get("cond") is plans[0].get("cond") for p_ in plans)
        hook = getattr(self, "step_hook", None)  # test instrumentation (drift per step); None on every product path
        if (self.use_graph and uniform and not strict and len(plans) > 2 and all(p_["sigma"] is None for p_ in plans) and not self._branch_split_active
                and rg == 0 and hook is None):
            xs = self._run_steps_graph(plans, xs, draw_noise, step, flat_dev, gens_dev, horizon)
        else:
            for i, p_ in enumerate(plans):
                xs = step(p_, xs, draw_noise(p_), p_["tables_dev"], p_["gen_dev"])
                if hook is not None:
                    hook(i, xs)
        self.window_forwards += sum(p_["bm"] for p_ in plans)
        if padding > 0:
            xs = xs[:, :-padding]
        return xs, None

    @torch.no_grad()
    def _sample_sequence_refine(self, batch_size: int, goback_length: int, n_goback: int, length: Optional[int] = None,
                                context: Optional[torch.Tensor] = None, context_mask: Optional[torch.Tensor] = None,
                                conditions: Optional[torch.Tensor] = None, history_guidance: Optional[HistoryGuidance] = None,
                                **_) -> Tuple[torch.Tensor, None]:
        """Refinement sampling of the fork (dfot_video.py:765-1008): the full-sequence ladder with excursions back up
        (``Schedule.refine_scheduling_matrix``).  A row whose last token of the first sample moves DOWN is an ordinary History-
        Guidance DDIM step; any other row re-noises every token from its level to the next (``q_sample_from_x_k``) with fresh
        clamped noise (noise tag "renoise").  In the reference a window whose last token is context or padding therefore only
        ever re-noises, also on descending rows, and returns NaN (so does a schedule ending at alphas_cumprod 0 with context
        tokens): those cases raise ValueError here.  The reference's denoising branch is only well-formed for one-branch guidance (it re-noises the
        (B,..) context with (B*NFE,..) levels and discards the result); with more branches this does the ordinary composed step."""
        return self._sample_sequence(batch_size, length=length, context=context, context_mask=context_mask, conditions=conditions,
                                     history_guidance=history_guidance, _refine=(int(goback_length), int(n_goback)))

    def _window_sampler(self):
        """which per-window sampler the rollout / interpolation drivers call (the base classes always use _sample_sequence)"""
        return self._sample_sequence

    def _reconstruction_guided_v(self, x_in, lvl, cond, cmask, tables, gen_dev, ctx, rg: float, nfe: int):
        """Reconstruction guidance (dfot_video.py:700-723, discrete_diffusion.py:485-513): the model prediction is differentiated w.r.t.
        x_t -- the backbone runs in its saved-activation form and its hand-written backward returns d / d x (ops.py) -- and the
        gradient of  rg/2 * sum( (pred_x0 - context)^2 * sqrt(alphas_cumprod) * [mask != 0] / #[mask != 0] )  shifts the predicted noise:
        eps' = eps + sqrt(1 - ac) * grad,  x0' = (x - sqrt(1 - ac) eps') / sqrt(ac).  For the v-objective that is exactly the ordinary
        step on  v' = v + sqrt(1 - ac) / sqrt(ac) * grad  (eps = sqrt(ac) v + sqrt(1 - ac) x), so v' goes into the fused composition
        kernel (the alphas_cumprod = 0 level is handled through the step's output, see the end of this function).  The few elementwise
        steps around the backbone run as torch ops under autograd, as in the reference; one-branch guidance only (the reference
        compares the (B * NFE, ...) prediction with the (B, ...) context)."""
        if nfe != 1:
            raise ValueError("reconstruction guidance needs one-branch history guidance (the reference's loss compares the "
                             "(B * NFE, ...) prediction with the (B, ...) context)")
        if self.cfg.diffusion.objective != "pred_v":
            raise ValueError("reconstruction guidance is implemented for the v objective")
        nd = x_in.ndim
        ext = lambda a: a.reshape(*a.shape, *([1] * (nd - a.ndim)))
        sa, s1 = ext(tables[2]), ext(tables[3])  # sqrt(alphas_cumprod), sqrt(1 - alphas_cumprod) of the step's (clamped) levels
        with torch.enable_grad():
            x = x_in.detach().requires_grad_(True)
            v = self.model(x, lvl, cond, cmask)
            pred_x0 = sa * x - s1 * v
            cm = ext((gen_dev == 0).to(torch.float32))
            loss = torch.sum((pred_x0 - ctx) ** 2 * sa * cm / cm.sum(dim=1, keepdim=True).clamp(min=1))
            grad = torch.nan_to_num(-torch.autograd.grad(-rg * 0.5 * loss, x)[0], nan=0.0)
        # sa > 0: v' = v + (s1 / sa) grad reproduces the reference's guided eps AND the x0 recomputed from it.  sa = 0 (zero terminal SNR,
        # the first step of the K600 cosine schedule): the reference keeps the unguided x0 but still uses the guided eps' = eps + s1 grad,
        # which no v can express (eps = sa v + s1 x does not depend on v there).  The step's OUTPUT an x0 + cn eps' is linear in v,
        # d out / d v = cn sa - an s1 = -an s1, so v' = v - (cn / an) grad yields exactly the reference's x_pred (an = sqrt of the next
        # level's alphas_cumprod > 0); x0 / eps are not used anywhere else in a one-branch step.
        an, cn = ext(tables[4]), ext(tables[5])
        scale = torch.where(sa > 0, s1 / sa.clamp(min=1e-30), torch.where(an > 0, -cn / an.clamp(min=1e-30), torch.zeros_like(sa)))
        return (v.detach() + scale * grad).contiguous()

    @staticmethod
    def _fresh_flags(plans, batch_size: int, horizon: int):
        """per step, uint8 (bm, horizon): 0 where the backbone input of (row, frame) is bit-for-bit the previous step's.  That holds when
        in BOTH steps the token enters as x_in = 1 * xs + 0 * noise (tables rows 0, 1) at the same model level (row 7), the previous
        composition left it alone (gen == 0: dfot_ddim_compose copies such tokens) and the steps share the branch batch and the
        conditioning tensors -- the clean context frames of a conditional History-Guidance branch (history_guidance.py:474-533,
        dfot_video.py:682-752).  Every frame of a window's first step, and of a step after a re-noising row, is fresh."""
        out, prev = [], None
        for p_ in plans:
            fresh = np.ones((p_["bm"], horizon), np.uint8)
            if p_.get("renoise"):
                out.append(fresh)
                prev = None
                continue
            if (prev is not None and prev["bm"] == p_["bm"] and prev["nfe"] == p_["nfe"] and prev.get("cond") is p_.get("cond")
                    and ((prev["cmask"] is None and p_["cmask"] is None)
                         or (prev["cmask"] is not None and p_["cmask"] is not None and np.array_equal(prev["cmask"], p_["cmask"])))):
                a, b = prev["tables"], p_["tables"]
                same_in = (a[0] == 1) & (a[1] == 0) & (b[0] == 1) & (b[1] == 0) & (a[7] == b[7])
                untouched = np.repeat(prev["gen"].reshape(batch_size, horizon) == 0, p_["nfe"], axis=0)
                fresh = np.where(same_in & untouched, 0, 1).astype(np.uint8)
            out.append(fresh)
            prev = p_
        return out

    def _run_steps_graph(self, plans, xs, draw_noise, step, flat_dev, gens_dev, horizon):
        """hipGraph execution of the step loop: step 0 runs eagerly (lazy initialisation, pose caches), then ALL remaining steps
        [hg_prepare -> backbone -> ddim/compose/clamp] x (n_steps - 1) are captured as ONE graph and launched with one replay.
        (Round 1 captured one step with a device-side step counter and replayed it 49 times: every replay of the same executable
        graph waited on the host for the previous one, 1.4 ms per step -- graph 8.0 vs eager 8.7 frames/s, profiles/r02_l_*.)
        Each captured step reads ITS slice of the static per-step tables, so nothing is indexed at run time.  The graph only
        touches static buffers owned by a cache entry keyed by the window shape, so windows of the same shape (the interpolation
        windows of a long rollout, successive samples of a benchmark) re-use it: a new window copies its tables / noise / guidance
        weights into the entry and replays -- capture is paid once per shape."""
        p0 = plans[0]
        bm, n_steps, nfe = p0["bm"], len(plans), p0["nfe"]
        need_noise = any(p_["need_noise"] for p_ in plans)
        self.model.reserve(bm)  # before looking at the generation: a growing workspace invalidates every captured pointer
        if hasattr(self.model, "sync_weights"):
            self.model.sync_weights()  # ... and so do re-loaded weights (kernel choices follow them) or a changed engine option
        gen = getattr(self.model, "reserve_generation", 0)
        if gen != self._graphs_generation:
            self._graphs.clear()
            self._graphs_generation = gen
        key = (bm, n_steps, nfe, horizon, tuple(xs.shape), need_noise, None if p0["cmask"] is None else p0["cmask"].tobytes(),
               id(self.model))
        ent = self._graphs.get(key)
        if ent is None:
            ent = dict(tables=torch.empty(n_steps, 8, bm, horizon, device="cuda", dtype=torch.float32),
                       gens=torch.empty_like(gens_dev), weights=torch.empty_like(p0["weights_dev"]),
                       lives=torch.empty(n_steps, bm, horizon, device="cuda", dtype=torch.uint8),
                       fresh=torch.empty(n_steps, bm, horizon, device="cuda", dtype=torch.uint8),
                       noise=torch.empty(n_steps, bm, *xs.shape[1:], device="cuda") if need_noise else None,
                       xs=torch.empty_like(xs), out=None, graph=None)
        ent["tables"].copy_(flat_dev.view(n_steps, 8, bm, horizon))
        ent["gens"].copy_(gens_dev)
        ent["lives"].copy_(gens_dev if nfe == 1 else gens_dev.repeat_interleave(nfe, dim=1))
        ent["fresh"].copy_(torch.stack([p_["fresh_dev"] for p_ in plans]))
        ent["weights"].copy_(p0["weights_dev"])
        if need_noise:
            for i, p_ in enumerate(plans):
                if p_["need_noise"]:
                    ent["noise"][i].copy_(draw_noise(p_))
                else:
                    ent["noise"][i].zero_()
        p_static = dict(p0, weights_dev=ent["weights"])
        ent["xs"].copy_(step(p_static, xs, None if not need_noise else ent["noise"][0], ent["tables"][0], ent["gens"][0], live_dev=ent["lives"][0],
                            fresh_dev=ent["fresh"][0]))
        if ent["graph"] is None:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread-local capture mode: with a process group alive (sharded rollouts) the RCCL watchdog thread polls events while this
            # thread captures; in the default "global" mode such calls from other threads invalidate the capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                x = ent["xs"]
                for i in range(1, n_steps):
                    x = step(p_static, x, ent["noise"][i] if need_noise else None, ent["tables"][i], ent["gens"][i], live_dev=ent["lives"][i],
                             fresh_dev=ent["fresh"][i])
                ent["out"] = x
            ent["graph"] = graph
            if len(self._graphs) >= 4:  # small LRU: each entry owns a private memory pool
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = ent
            self.graph_captures += 1
        ent["graph"].replay()
        self.graph_replays += n_steps - 1
        return ent["out"].clone()

    # ------------------------------------------------------------------ sliding window
    @torch.no_grad()
    def _predict_sequence(self, context: torch.Tensor, length: Optional[int] = None,
                          conditions: Optional[torch.Tensor] = None, history_guidance: Optional[HistoryGuidance] = None,
                          sliding_context_len: Optional[int] = None, **_) -> Tuple[torch.Tensor, None]:
        mt = self.max_tokens
        if length is None:
            length = mt
        if sliding_context_len is None:
            if mt < length:
                raise ValueError("when length > max_tokens, sliding_context_len must be specified.")
            sliding_context_len = mt - 1
        if sliding_context_len == -1:
            sliding_context_len = mt - 1
        batch_size, gt_len = context.shape[:2]
        if sliding_context_len < gt_len:
            raise ValueError("sliding_context_len is expected to be >= length of initial context,"
                             f"got {sliding_context_len}. If you are trying to use max context, "
                             "consider specifying sliding_context_len=-1.")
        xs = context.to(device=self.device, dtype=torch.float32)
        cur = gt_len
        n_window = 0
        while cur < length:
            c = min(sliding_context_len, cur)
            h = min(length - cur, mt - c)
            window = torch.cat([xs[:, -c:], xs.new_zeros(batch_size, h, *self.x_shape)], 1)
            generated = cur - max(cur - c, gt_len)
            cmask = torch.ones(batch_size, c, dtype=torch.long)
            if generated > 0:
                cmask[:, -generated:] = 2
            cmask = torch.cat([cmask, torch.zeros(batch_size, h, dtype=torch.long)], 1)
            cond = None if conditions is None else conditions[:, cur - c: cur - c + mt]
            if hasattr(self.noise_fn, "set_windows"):
                # key-frame windows are replicated on every rank: key their noise by the sliding-window index (rank-independent),
                # never by whatever the previous interpolation batch of this rank left behind
                self.noise_fn.set_windows([n_window])
            n_window += 1
            new, _ = self._window_sampler()(batch_size, length=c + h, context=window, context_mask=cmask,
                                           conditions=cond, history_guidance=history_guidance)
            xs = torch.cat([xs, new[:, -h:]], 1)
            cur = xs.shape[1]
        return xs, None

    # ------------------------------------------------------------------ interpolation
    def _interpolation_plan(self, known: np.ndarray) -> List[List[np.ndarray]]:
        mt = self.max_tokens
        known = known.copy()
        plan: List[List[np.ndarray]] = []
        while not known.all():
            keys = np.where(known)[0]
            stage: List[np.ndarray] = []
            chunk: Optional[np.ndarray] = None
            for left, right in zip(keys[:-1], keys[1:]):
                left, right = int(left), int(right)
                if chunk is not None:
                    if len(chunk) + right - left <= mt:
                        chunk = np.concatenate([chunk, np.arange(left + 1, right + 1)])
                        continue
                    stage.append(chunk)
                    chunk = None
                if right - left == 1:
                    continue
                if right - left >= mt - 1:
                    stage.append(torch.linspace(left, right, mt).round().long().numpy())
                else:
                    chunk = np.arange(left, right + 1)
            if chunk is not None:
                stage.append(chunk)
            for w in stage:
                known[w] = True
            plan.append(stage)
        return plan

    @torch.no_grad()
    def _interpolate_videos(self, context: torch.Tensor, context_mask: Optional[torch.Tensor] = None,
                            conditions: Optional[torch.Tensor] = None, **_) -> torch.Tensor:
        cfg = self.cfg
        if context_mask is None:
            context_mask = torch.zeros(context.shape[0], context.shape[1], dtype=torch.bool)
            context_mask[:, [0, -1]] = True
        else:
            context_mask = context_mask.detach().cpu().bool()
            assert bool(context_mask[:, [0, -1]].all()), "The first and last frames must be known to interpolate."
        hg = HistoryGuidance.from_config(cfg.interpolation_guidance, timesteps=self.timesteps)
        xs = context.to(device=self.device, dtype=torch.float32).clone()
        known = context_mask.clone()
        for si, stage in enumerate(self._interpolation_plan(context_mask[0].numpy())):
            ctx = torch.cat([self._pad_to_max_tokens(xs[:, w]) for w in stage], 0)
            msk = torch.cat([self._pad_to_max_tokens(known[:, w]) for w in stage], 0)
            cnd = None if conditions is None else torch.cat([self._pad_to_max_tokens(conditions[:, w]) for w in stage], 0)

            def sample_batch(ids, ctx=ctx, msk=msk, cnd=cnd, si=si):
                if not ids:
                    return ctx.new_zeros((0, *ctx.shape[1:]))
                if hasattr(self.noise_fn, "set_windows"):
                    self.noise_fn.set_windows([(si + 1) * 100000 + i for i in ids])
                o, _ = self._window_sampler()(len(ids), context=ctx[ids], context_mask=msk[ids].long(),
                                             conditions=None if cnd is None else cnd[ids], history_guidance=hg)
                return o

            mb = cfg.interpolation_max_batch_size or ctx.shape[0]
            if self.shard_windows:
                # windows of one plan stage are independent units: round-robin over ranks, one all-gather per stage
                out = parallel.run_sharded(ctx.shape[0], sample_batch, mb)
            else:
                out = torch.cat([sample_batch(list(range(i, min(i + mb, ctx.shape[0])))) for i in range(0, ctx.shape[0], mb)], 0)
            for w, pred in zip(stage, out.chunk(len(stage), 0)):
                xs[:, w] = pred[:, : len(w)]
                known[:, w] = True
        return xs

    # ------------------------------------------------------------------ top level
    @torch.no_grad()
    def _predict_videos(self, xs: torch.Tensor, n_context_tokens: int,
                        conditions: Optional[torch.Tensor] = None) -> torch.Tensor:
        cfg = self.cfg
        out = xs.to(device=self.device, dtype=torch.float32).clone()
        hg = HistoryGuidance.from_config(cfg.prediction_guidance, timesteps=self.timesteps)
        density = cfg.keyframe_density or 1
        if density > 1:
            raise ValueError("tasks.prediction.keyframe_density must be <= 1")
        n = out.shape[1]
        keys = torch.linspace(0, n - 1, round(density * n)).round().long()
        keys = torch.cat([torch.arange(n_context_tokens), keys]).unique()
        kc = None if conditions is None else conditions[:, keys]
        if self.branch_parallel and parallel.world_info()[0] > 1 and not (hasattr(self.noise_fn, "set_windows") or getattr(self.noise_fn, "replicated", False)):
            # the gathered v mixes branches evaluated on DIFFERENT ranks: they must all hold bit-identical xs and noise
            raise ValueError("branch_parallel needs a rank-independent noise source: parallel.WindowKeyedNoise (set_windows) or a noise_fn "
                             "whose attribute `replicated` is True (every rank draws the same tensors)")
        self._branch_split_active = bool(self.branch_parallel)  # key-frame windows are replicated on every rank: split their branches
        try:
            pred, _ = self._predict_sequence(out[:, :n_context_tokens], length=len(keys), conditions=kc, history_guidance=hg,
                                             sliding_context_len=cfg.sliding_context_len or self.max_tokens // 2)
        finally:
            self._branch_split_active = False  # interpolation windows are sharded whole (both branches on one GPU)
        out[:, keys.to(self.device)] = pred
        if len(keys) < n:
            known = torch.zeros(out.shape[0], n, dtype=torch.bool)
            known[:, keys] = True
            out = self._interpolate_videos(out, known, conditions)
        return out


class DFoTVideoSampler(DFoTVideoPoseSampler):
    """The reference's base algorithm (algorithms/dfot/dfot_video.py: DFoTVideo), used by the un-conditioned video
    configurations such as Kinetics-600 (DiT3D + DiscreteDiffusion): identical sampling path, no camera poses."""

    def _process_conditions(self, conditions: Optional[torch.Tensor], noise_levels=None) -> Optional[torch.Tensor]:
        if conditions is not None:
            raise ValueError("DFoTVideoSampler takes no external conditions; use DFoTVideoPoseSampler for camera poses")
        return None


class DifferenceDFoTVideoSampler(DFoTVideoSampler):
    """The bash/k600 algorithm (algorithms/dfot/difference_dfot_video.py: DifferenceDFoTVideo): every frame travels with its
    temporal difference as a second token, interleaved (difference_t, frame_t); the sampler itself is the DFoTVideo path run
    on the 2T merged tokens (its `_sample_sequence` / `_predict_sequence` / `_predict_videos` differ from the base class
    only by `max_tokens * 2`, difference_dfot_video.py:214-278,463-607,609-846).  Construct it with
    ``SamplerConfig(max_tokens=2 * T)`` and a ``DifferenceDiT3D`` backbone built with ``max_tokens=T``."""

    def _window_sampler(self):
        r = self.cfg.refinement_sampling
        if r and r.get("enabled"):
            import functools
            return functools.partial(self._sample_sequence_refine, goback_length=int(r["goback_length"]), n_goback=int(r["n_goback"]))
        return self._sample_sequence

    @staticmethod
    def merge_tensors(x: Optional[torch.Tensor], y: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """merge_type 'interleaved' (difference_dfot_video.py:45-61): (x_0, y_0, x_1, y_1, ...) along the token axis."""
        if x is None or y is None:
            return None
        assert x.shape == y.shape, "Tensors must have the same shape to be merged."
        return torch.stack([x, y], dim=2).reshape(x.shape[0], 2 * x.shape[1], *x.shape[2:])

    @staticmethod
    def unmerge_tensors(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """inverse of merge_tensors (difference_dfot_video.py:63-75)."""
        return x[:, 0::2], x[:, 1::2]

    @torch.no_grad()
    def _sample_all_videos(self, xs: torch.Tensor, n_context_tokens: int) -> Dict[str, torch.Tensor]:
        """difference_dfot_video.py:166-212 without the logging/VAE tail: frames -> (difference, frame) tokens -> prediction
        on the merged sequence with doubled context -> {"prediction", "prediction_diff"}."""
        difference = torch.diff(xs, dim=1, prepend=xs[:, :1])
        merged = self.merge_tensors(difference, xs)
        out = self._predict_videos(merged, n_context_tokens=2 * n_context_tokens, conditions=None)
        gen_diff, gen = self.unmerge_tensors(out)
        return {"gt": xs.clone(), "prediction": gen, "prediction_diff": gen_diff}
