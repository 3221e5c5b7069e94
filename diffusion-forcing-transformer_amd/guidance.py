"""History Guidance -- host-side branch planner of the MI355X engine.

Same public constructors as the reference's ``HistoryGuidance``
(algorithms/dfot/history_guidance.py:571-900): ``from_config``, ``conditional``,
``stabilized_conditional``, ``vanilla``, ``stabilized_vanilla``, ``fractional``,
``stabilized_fractional``.  Instead of materialising NFE-replicated tensors with a chain of
torch ops per step (``prepare`` :446-543 / :929-973, ``compose`` :545-568 / :978-982), the
planner emits, once per sampling step, a tiny table per (sample, branch, token):

    level[b,h,t]   noise level the branch shows the model for that token
    weight[h]      composition weight            cond_masked[h]  drop the camera pose?

from which the sampler derives the q_sample / DDIM coefficient tables consumed by
``dfot_hg_prepare`` and ``dfot_ddim_compose``.  Temporal / custom guidance (history sub-sequences via ``time_indices`` and
several ``gen_segments``: :105-149, :357-568) adds per-(branch, token) composition weights and an "excluded" flag for tokens a
gen segment leaves out (shown to the model as pure noise at level T-1, weight 0).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

ALL = "all"


@dataclass
class HistorySegment:
    time_indices: object = ALL
    freq_ranges: Sequence = (ALL,)
    freq_ranges_if_generated: Optional[Sequence] = None

    def ranges(self, generated: bool) -> Sequence:
        if generated and self.freq_ranges_if_generated is not None:
            return self.freq_ranges_if_generated
        return self.freq_ranges


@dataclass
class BranchPlan:
    levels: np.ndarray       # (B, H, T) int64; for non-history tokens: the incoming `from` level
    to_levels: np.ndarray    # (B, H, T) int64
    weights: np.ndarray      # (H,) float32
    cond_masked: Optional[np.ndarray]  # (H,) bool or None (no pose guidance at all)
    replace: np.ndarray      # (B, H, T) bool: history token re-noised to levels[b,h,t]
    # several gen segments only (H = hist branches x gen segments, gen segment fastest):
    excluded: Optional[np.ndarray] = None     # (B, H, T) bool: token to be generated but outside the branch's gen segment
    tok_weights: Optional[np.ndarray] = None  # (H, T) float32: weight / (number of gen segments covering the token), 0 if excluded
    n_gen: int = 1

    @property
    def nfe(self) -> int:
        return int(self.weights.shape[0])


class HistoryGuidance:
    def __init__(self, hist_segments: List[HistorySegment], hist_weights: List[float], gen_segments=None,
                 timesteps: int = 1000, use_external_cond_guidance: bool = False, visualize: bool = False):
        if len(hist_segments) != len(hist_weights):
            raise AssertionError("Length of hist_segments and hist_weights should be the same")
        self.needs_pose_interpolation = False
        self.gen_segments = [ALL] if gen_segments is None else [g if g == ALL else list(g) for g in gen_segments]
        self.hist_segments, self.hist_weights = hist_segments, list(hist_weights)
        self.timesteps = timesteps
        self.use_external_cond_guidance = use_external_cond_guidance

    # ---------------------------------------------------------------- constructors
    @classmethod
    def from_config(cls, config: Dict, timesteps: int = 1000) -> "HistoryGuidance":
        cfg = dict(config)
        name = cfg.pop("name")
        cfg.pop("visualize", None)
        if not hasattr(cls, name) or name.startswith("_"):
            raise AttributeError(f"unknown history guidance scheme '{name}'")
        return getattr(cls, name)(**cfg, timesteps=timesteps)

    @classmethod
    def conditional(cls, timesteps: int = 1000, **_):
        return cls([HistorySegment()], [1], timesteps=timesteps, use_external_cond_guidance=False)

    @classmethod
    def stabilized_conditional(cls, stabilization_level: float, timesteps: int = 1000, **_):
        seg = HistorySegment(ALL, (ALL,), ((stabilization_level, 1.0),))
        return cls([seg], [1], timesteps=timesteps, use_external_cond_guidance=False)

    @classmethod
    def vanilla(cls, guidance_scale: float, timesteps: int = 1000, use_external_cond_guidance: bool = True, **_):
        return cls([HistorySegment()], [guidance_scale], timesteps=timesteps,
                   use_external_cond_guidance=use_external_cond_guidance)

    @classmethod
    def stabilized_vanilla(cls, guidance_scale: float, stabilization_level: float, timesteps: int = 1000,
                           use_external_cond_guidance: bool = True, **_):
        seg = HistorySegment(ALL, (ALL,), ((stabilization_level, 1.0),))
        return cls([seg], [guidance_scale], timesteps=timesteps, use_external_cond_guidance=use_external_cond_guidance)

    @classmethod
    def fractional(cls, guidance_scale: float, freq_scale: float, timesteps: int = 1000,
                   use_external_cond_guidance: bool = True, **_):
        segs = [HistorySegment(), HistorySegment(ALL, ((freq_scale, 1.0),))]
        return cls(segs, [1, guidance_scale - 1], timesteps=timesteps,
                   use_external_cond_guidance=use_external_cond_guidance)

    @classmethod
    def stabilized_fractional(cls, guidance_scale: float, freq_scale: float, stabilization_level: float,
                              timesteps: int = 1000, use_external_cond_guidance: bool = True, **_):
        segs = [HistorySegment(ALL, (ALL,), ((stabilization_level, 1.0),)), HistorySegment(ALL, ((freq_scale, 1.0),))]
        return cls(segs, [1, guidance_scale - 1], timesteps=timesteps,
                   use_external_cond_guidance=use_external_cond_guidance)

    @classmethod
    def temporal(cls, hist_subsequences, hist_weights, gen_segments=None, timesteps: int = 1000,
                 use_external_cond_guidance: bool = True, **_):
        """Temporal History Guidance (HG-t, :832-859): one full-frequency segment per history sub-sequence."""
        segs = [HistorySegment(time_indices=ALL if sub == ALL else list(sub)) for sub in hist_subsequences]
        hg = cls(segs, hist_weights, gen_segments=gen_segments, timesteps=timesteps,
                 use_external_cond_guidance=use_external_cond_guidance)
        # with camera poses the reference re-interpolates the poses of fully masked frames (dfot_video_pose.py:77-84,
        # CameraPose.replace_with_interpolation: quaternion slerp): the pose sampler does so per branch (pose.py)
        hg.needs_pose_interpolation = True
        return hg

    @classmethod
    def custom(cls, hist_segments, hist_weights, gen_segments=None, timesteps: int = 1000,
               use_external_cond_guidance: bool = True, **_):
        """The most flexible constructor (:861-900): hist_segments = dicts with time_indices / freq_ranges /
        freq_ranges_if_generated."""
        def ranges(r):
            return None if r is None else tuple(ALL if x == ALL else tuple(x) for x in r)
        segs = [HistorySegment(time_indices=ALL if d["time_indices"] == ALL else list(d["time_indices"]),
                               freq_ranges=ranges(d["freq_ranges"]) or (ALL,),
                               freq_ranges_if_generated=ranges(d.get("freq_ranges_if_generated")))
                for d in hist_segments]
        return cls(segs, hist_weights, gen_segments=gen_segments, timesteps=timesteps,
                   use_external_cond_guidance=use_external_cond_guidance)

    # ---------------------------------------------------------------- planning
    @property
    def is_simple(self) -> bool:
        """the reference's dispatch to SimpleHistoryGuidanceManager (:635-653)"""
        s = self.hist_segments[0]
        gen = s.freq_ranges if s.freq_ranges_if_generated is None else s.freq_ranges_if_generated
        return (len(self.hist_weights) == 1 and len(s.freq_ranges) == 1 and s.freq_ranges[0] == ALL and gen[0] == ALL
                and s.time_indices == ALL and self.gen_segments == [ALL])

    @staticmethod
    def _range_for(ranges: Sequence, i: int, n: int) -> Tuple[float, float]:
        rs = [(0.0, 1.0) if r == ALL else (float(r[0]), float(r[1])) for r in ranges]
        if len(rs) == n:
            return rs[i]
        if len(rs) == 2:
            if n == 1:
                return rs[1]
            (a0, a1), (b0, b1) = rs
            return (a0 + (b0 - a0) * i / (n - 1), a1 + (b1 - a1) * i / (n - 1))
        if len(rs) == 1:
            return rs[0]
        raise ValueError(f"The length of the history is {n}, but the length of freq_ranges is {len(rs)}.")

    def _to_level(self, frac: np.ndarray) -> np.ndarray:
        # torch.tensor(float list) * timesteps - 1 evaluated in float32, then truncated (:428-432)
        return (frac.astype(np.float32) * np.float32(self.timesteps) - np.float32(1)).astype(np.int64)

    def plan(self, mask: np.ndarray, frm: np.ndarray, to: np.ndarray, replacement_only: bool = False) -> BranchPlan:
        """mask/frm/to: (B,T) int64 (mask: 0 generate, 1 GT history, 2 generated history, -1 padding)."""
        b, t = mask.shape
        hist = mask >= 1
        if self.is_simple:
            scale = float(self.hist_weights[0])
            if scale == 1.0:
                lv, tl = frm[:, None, :].copy(), to[:, None, :].copy()
                return BranchPlan(lv, tl, np.ones(1, np.float32), None, np.zeros((b, 1, t), bool))
            top = self.timesteps - 1
            lv = np.stack([np.where(hist, top, frm), frm], axis=1)
            tl = np.stack([np.where(hist, top, to), to], axis=1)
            repl = np.stack([hist, np.zeros_like(hist)], axis=1)
            cm = np.array([True, False]) if self.use_external_cond_guidance else None
            return BranchPlan(lv, tl, np.array([-(scale - 1.0), scale], np.float32), cm, repl)

        if not (mask == mask[:1]).all():
            raise AssertionError("`mask` should be the same across the batch to use history guidance.")
        row = mask[0]
        hidx = np.where(row >= 1)[0]
        n = len(hidx)
        generated = row[hidx] == 2
        acc: Dict[tuple, float] = {}
        order: List[tuple] = []

        def add(key, w):
            if key not in acc:
                acc[key] = 0.0
                order.append(key)
            acc[key] += w

        ucg = bool(self.use_external_cond_guidance)
        add((1.0,) * n + (ucg,), 1.0)
        for seg, w in zip(self.hist_segments, self.hist_weights):
            # HistorySegment.to_noise_levels (:105-149): tokens outside time_indices are fully masked (1.0, 1.0); the freq
            # ranges are indexed by the position inside the chosen sub-sequence
            chosen = list(range(n)) if seg.time_indices == ALL else [i if i >= 0 else n + i for i in seg.time_indices]
            if any(not 0 <= i < n for i in chosen):
                raise AssertionError("time_indices should be between 0 and hist_len.")
            pairs = [(1.0, 1.0)] * n
            for j, tok in enumerate(chosen):
                pairs[tok] = self._range_for(seg.ranges(bool(generated[tok])), j, len(chosen))
            start = tuple(p[0] for p in pairs)
            end = tuple(p[1] for p in pairs)
            add(start + (False,), float(w))
            add(end + (ucg,), -float(w))
        keys = [k for k in order if acc[k] != 0]
        hlev = self._to_level(np.array([k[:-1] for k in keys], dtype=np.float64).reshape(len(keys), n))
        h = len(keys)
        lv = np.repeat(frm[:, None, :], h, axis=1).copy()
        tl = np.repeat(to[:, None, :], h, axis=1).copy()
        if not replacement_only:
            lv[:, :, hidx] = hlev[None]
            tl[:, :, hidx] = hlev[None]
        repl = (lv >= 0) & hist[:, None, :]
        weights = np.array([acc[k] for k in keys], np.float32)
        cond = np.array([bool(k[-1]) for k in keys])
        if self.gen_segments == [ALL]:
            return BranchPlan(lv, tl, weights, cond, repl)
        # several gen segments (:386-395, :512-536, :545-568): every history branch is evaluated once per segment; tokens
        # to be generated but outside the segment are shown as pure noise at level T-1 and contribute nothing
        gidx = np.where(row == 0)[0]
        g = len(self.gen_segments)
        gen_mask = np.zeros((g, t), bool)
        for i, seg in enumerate(self.gen_segments):
            gen_mask[i, gidx if seg == ALL else gidx[np.asarray(seg, dtype=np.int64)]] = True
        cover = np.maximum(gen_mask.sum(0), 1).astype(np.float32)
        lv, tl, repl = (np.repeat(a, g, axis=1) for a in (lv, tl, repl))          # (B, h*g, T), gen segment fastest
        excl = np.tile(~gen_mask, (h, 1))[None] & (mask == 0)[:, None, :]          # (B, h*g, T)
        top = self.timesteps - 1
        lv, tl = np.where(excl, top, lv), np.where(excl, top, tl)
        tokw = (np.repeat(weights, g)[:, None] * np.tile(gen_mask, (h, 1)) / cover[None]).astype(np.float32)
        return BranchPlan(lv, tl, np.repeat(weights, g), np.repeat(cond, g), repl, excluded=excl, tok_weights=tokw, n_gen=g)
