"""History Guidance (oracle, CPU): branch construction, prepare and compose.

Restates algorithms/dfot/history_guidance.py:
  * HistorySegment.to_noise_levels / _process_freq_ranges   -- :70-149
  * HistoryGuidanceManager.__enter__ / prepare / compose    -- :357-568
  * SimpleHistoryGuidanceManager.prepare / compose          -- :929-982
  * manager dispatch rule                                   -- :635-653
  * scheme constructors conditional / vanilla / stabilized_* / fractional / temporal / custom -- :700-900

A *scheme* is (segments, weights, use_cond_guidance) with each segment =
(time_indices|"all", freq_ranges, freq_ranges_if_generated).  A *branch* is one model
evaluation: per-history-token noise levels, a weight, and whether the external
condition (camera pose) is masked.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch

ALL = "all"
NoiseFn = Callable[[str, tuple], torch.Tensor]


@dataclass
class Segment:
    time_indices: object = ALL
    freq_ranges: Sequence = (ALL,)
    freq_ranges_if_generated: Optional[Sequence] = None

    def gen_ranges(self):
        return self.freq_ranges if self.freq_ranges_if_generated is None else self.freq_ranges_if_generated


@dataclass
class Scheme:
    segments: List[Segment]
    weights: List[float]
    use_cond_guidance: bool = False
    gen_segments: Sequence = (ALL,)
    timesteps: int = 1000

    @property
    def is_simple(self) -> bool:
        s = self.segments[0]
        return (len(self.weights) == 1 and len(s.freq_ranges) == 1 and s.freq_ranges[0] == ALL
                and s.gen_ranges()[0] == ALL)


def make_scheme(name: str, timesteps: int = 1000, **kw) -> Scheme:
    if name == "conditional":
        return Scheme([Segment()], [1], False, timesteps=timesteps)
    if name == "stabilized_conditional":
        return Scheme([Segment(ALL, (ALL,), ((kw["stabilization_level"], 1.0),))], [1], False, timesteps=timesteps)
    ucg = kw.get("use_external_cond_guidance", True)
    if name == "vanilla":
        return Scheme([Segment()], [kw["guidance_scale"]], ucg, timesteps=timesteps)
    if name == "stabilized_vanilla":
        seg = Segment(ALL, (ALL,), ((kw["stabilization_level"], 1.0),))
        return Scheme([seg], [kw["guidance_scale"]], ucg, timesteps=timesteps)
    if name == "fractional":
        segs = [Segment(), Segment(ALL, ((kw["freq_scale"], 1.0),))]
        return Scheme(segs, [1, kw["guidance_scale"] - 1], ucg, timesteps=timesteps)
    if name == "stabilized_fractional":
        segs = [Segment(ALL, (ALL,), ((kw["stabilization_level"], 1.0),)),
                Segment(ALL, ((kw["freq_scale"], 1.0),))]
        return Scheme(segs, [1, kw["guidance_scale"] - 1], ucg, timesteps=timesteps)
    if name == "temporal":  # :832-859
        segs = [Segment(time_indices=ALL if sub == ALL else list(sub)) for sub in kw["hist_subsequences"]]
        return Scheme(segs, list(kw["hist_weights"]), ucg, gen_segments=kw.get("gen_segments") or (ALL,), timesteps=timesteps)
    if name == "custom":  # :861-900
        tup = lambda r: None if r is None else tuple(ALL if x == ALL else tuple(x) for x in r)
        segs = [Segment(ALL if d["time_indices"] == ALL else list(d["time_indices"]), tup(d["freq_ranges"]),
                        tup(d.get("freq_ranges_if_generated"))) for d in kw["hist_segments"]]
        return Scheme(segs, list(kw["hist_weights"]), ucg, gen_segments=kw.get("gen_segments") or (ALL,), timesteps=timesteps)
    raise ValueError(f"oracle: unsupported history guidance scheme {name}")


def _expand_ranges(ranges: Sequence, n: int) -> List[Tuple[float, float]]:
    rs = [(0.0, 1.0) if r == ALL else tuple(r) for r in ranges]
    if len(rs) == n:
        return rs
    if len(rs) == 2:
        if n == 1:
            return [rs[1]]
        (a0, a1), (b0, b1) = rs
        return [(a0 + (b0 - a0) * i / (n - 1), a1 + (b1 - a1) * i / (n - 1)) for i in range(n)]
    if len(rs) == 1:
        return rs * n
    raise ValueError("freq_ranges length does not match the history length")


def _segment_levels(seg: Segment, generated: List[bool]):
    n = len(generated)
    idx = list(range(n)) if seg.time_indices == ALL else [i if i >= 0 else n + i for i in seg.time_indices]
    gt_r = _expand_ranges(seg.freq_ranges, len(idx))
    gen_r = _expand_ranges(seg.gen_ranges(), len(idx))
    final = [(1.0, 1.0)] * n
    for j, tok in enumerate(idx):
        final[tok] = gen_r[j] if generated[tok] else gt_r[j]
    if n == 0:
        return (), ()
    start, end = zip(*final)
    return tuple(start), tuple(end)


@dataclass
class Branches:
    hist_idx: torch.Tensor      # (hist_len,)
    gen_idx: torch.Tensor       # (gen_len,)
    levels: torch.Tensor        # (H, hist_len) int64
    weights: torch.Tensor       # (H,) float32
    cond_masked: torch.Tensor   # (H,) bool
    gen_mask: torch.Tensor      # (G, T) bool


def build_branches(scheme: Scheme, mask_row: torch.Tensor) -> Branches:
    """mask_row (T,): 0 generate, 1 GT history, 2 generated history, -1 padding."""
    hist_idx = torch.where(mask_row >= 1)[0]
    gen_idx = torch.where(mask_row == 0)[0]
    n_hist, n_gen = len(hist_idx), len(gen_idx)
    gsegs = [list(range(n_gen)) if g == ALL else list(g) for g in scheme.gen_segments]
    gen_mask = torch.zeros(len(gsegs), len(mask_row), dtype=torch.bool)
    for i, g in enumerate(gsegs):
        gen_mask[i, gen_idx[g]] = True
    acc: "OrderedDict[tuple, float]" = OrderedDict()
    acc[(1.0,) * n_hist + (scheme.use_cond_guidance,)] = 1.0
    generated = (mask_row[hist_idx] == 2).tolist()
    for seg, w in zip(scheme.segments, scheme.weights):
        start, end = _segment_levels(seg, generated)
        acc[start + (False,)] = acc.get(start + (False,), 0.0) + w
        key = end + (scheme.use_cond_guidance,)
        acc[key] = acc.get(key, 0.0) - w
    keep = [(k, w) for k, w in acc.items() if w != 0]
    lv = torch.tensor([k[:-1] for k, _ in keep], dtype=torch.float32).reshape(len(keep), n_hist)
    levels = (lv * scheme.timesteps - 1).long()
    return Branches(hist_idx, gen_idx, levels,
                    torch.tensor([w for _, w in keep], dtype=torch.float32),
                    torch.tensor([bool(k[-1]) for k, _ in keep]), gen_mask)


def _ext(a: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    return a.reshape(*a.shape, *([1] * (x.ndim - a.ndim)))


class Guidance:
    """One sampling step's prepare/compose pair (the reference's context manager)."""

    def __init__(self, scheme: Scheme, mask: torch.Tensor):
        self.scheme, self.mask = scheme, mask
        self.simple = scheme.is_simple
        if self.simple:
            self.scale = scheme.weights[0]
            self.nfe = 1 if self.scale == 1 else 2
        else:
            if not bool((mask == mask[0]).all()):
                raise AssertionError("mask must be identical across the batch for this scheme")
            self.br = build_branches(scheme, mask[0])
            self.nfe = self.br.gen_mask.shape[0] * len(self.br.weights)

    # -- prepare ------------------------------------------------------------------
    def prepare(self, x, frm, to, q_sample, noise_fn: NoiseFn, replacement_only: bool = False):
        """returns x (B*nfe,...), from, to, cond_mask (B*nfe,) | None"""
        if self.simple:
            return self._prepare_simple(x, frm, to, q_sample, noise_fn)
        br, ts = self.br, self.scheme.timesteps
        b, h, g = x.shape[0], len(br.weights), br.gen_mask.shape[0]
        rep = lambda y: y.unsqueeze(1).repeat(1, h, *([1] * (y.ndim - 1))).clone()
        x, frm, to, mask = rep(x), rep(frm), rep(to), rep(self.mask)
        if not replacement_only:
            frm[:, :, br.hist_idx] = br.levels
            to[:, :, br.hist_idx] = br.levels
        replace = (frm >= 0) & (mask >= 1)
        flat = x.flatten(0, 1)
        noisy = q_sample(flat, frm.flatten(0, 1), noise_fn("q_sample", tuple(flat.shape))).view_as(x)
        x = torch.where(_ext(replace, x), noisy, x)
        repg = lambda y: y.flatten(0, 1).unsqueeze(1).repeat(1, g, *([1] * (y.ndim - 2))).clone()
        x, frm, to, mask = repg(x), repg(frm), repg(to), repg(mask)
        self.excluded = (~br.gen_mask) & (mask == 0)
        frm = torch.where(self.excluded, ts - 1, frm)
        to = torch.where(self.excluded, ts - 1, to)
        x = torch.where(_ext(self.excluded, x), noise_fn("excluded", tuple(x.shape)), x)
        cond_mask = br.cond_masked.view(1, h, 1).expand(b, h, g).reshape(-1).clone()
        return x.flatten(0, 1), frm.flatten(0, 1), to.flatten(0, 1), cond_mask

    def _prepare_simple(self, x, frm, to, q_sample, noise_fn):
        if self.scale == 1:
            return x, frm, to, None
        ts = self.scheme.timesteps
        hist = self.mask >= 1
        frm_u = torch.where(hist, ts - 1, frm)
        to_u = torch.where(hist, ts - 1, to)
        noisy = q_sample(x, frm_u, noise_fn("q_sample", tuple(x.shape)))
        x_u = torch.where(_ext(hist, x), noisy, x)
        stack = lambda u, c: torch.stack([u, c], dim=1).flatten(0, 1)
        cond_mask = None
        if self.scheme.use_cond_guidance:
            cond_mask = torch.tensor([True, False]).repeat(x.shape[0])
        return stack(x_u, x), stack(frm_u, frm), stack(to_u, to), cond_mask

    # -- compose ------------------------------------------------------------------
    def compose(self, x: torch.Tensor) -> torch.Tensor:
        if self.simple:
            if self.scale == 1:
                return x
            x = x.view(-1, 2, *x.shape[1:])
            return x[:, 1] * self.scale - x[:, 0] * (self.scale - 1)
        br = self.br
        h, g = len(br.weights), br.gen_mask.shape[0]
        x = x.view(-1, g, *x.shape[1:])
        x = torch.where(_ext(self.excluded, x), torch.zeros_like(x), x)
        x = x.view(-1, h, g, *x.shape[2:])
        x = (x * _ext(br.weights.view(1, h, 1), x)).sum(1).sum(1)
        denom = br.gen_mask.long().sum(0).clamp(min=1)
        return x / _ext(denom.view(1, -1), x)
