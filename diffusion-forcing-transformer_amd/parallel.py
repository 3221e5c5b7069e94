"""Multi-GPU sharding of independent sampling units (one process per GPU, torch.distributed over RCCL).

The DFoT path has no sequence parallelism: a window is one 8-frame forward, and the windows of
one interpolation plan stage (dfot_video.py:284-358) are independent.  They are dealt round-robin to ranks and the
finished windows are exchanged with ONE all-gather per plan stage.  Randomness is keyed by window id, not by rank
or batch position, so the sharded result is identical to the single-GPU result.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_windows(n_windows: int, world: int, rank: int) -> List[int]:
    """Window ids owned by `rank` (round-robin keeps both HG branches of a window on one GPU)."""
    return list(range(rank, n_windows, world))


def gather_windows(local: torch.Tensor, n_windows: int, group=None) -> torch.Tensor:
    """local: [n_local, ...] results of shard_windows(n_windows, world, rank) in order -> [n_windows, ...] on every
    rank.  One collective: ranks pad to ceil(n/world) rows, all-gather, and rows are re-interleaved."""
    world, rank = world_info(group)
    if world == 1:
        return local
    per = (n_windows + world - 1) // world
    if local.shape[0] == per:
        pad = local.contiguous()
    else:
        pad = local.new_zeros((per, *local.shape[1:]))
        pad[: local.shape[0]] = local
    # one collective straight into one buffer [world][per][...] (no per-rank list + stack copies)
    out = pad.new_empty((world * per, *local.shape[1:]))
    dist.all_gather_into_tensor(out, pad, group=group)
    # rank-major [world][per] -> window id = i*world + r
    return out.view(world, per, *local.shape[1:]).transpose(0, 1).reshape(per * world, *local.shape[1:])[:n_windows].contiguous()


class WindowKeyedNoise:
    """noise_fn whose draws depend only on (seed, window id, tag, draw counter of that window): the same window gets the
    same noise whichever rank or batch slot it is sampled in.  `set_windows` is called by the sampler per batch."""

    def __init__(self, seed: int, device: str = "cuda", clip: float = 20.0):
        self.seed, self.device, self.clip = int(seed), device, clip
        self.keys: Sequence[int] = (0,)
        self.counters = {}

    def set_windows(self, keys: Sequence[int]) -> None:
        self.keys = tuple(int(k) for k in keys)

    def __call__(self, tag: str, shape: tuple) -> torch.Tensor:
        rows = shape[0]
        if rows % len(self.keys) != 0:
            raise ValueError(f"WindowKeyedNoise: a draw of {rows} rows cannot be split over the {len(self.keys)} window keys set by "
                             "set_windows (stale keys from another batch?)")
        per = rows // len(self.keys)
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        tag_id = {"init": 1, "q_sample": 2, "excluded": 3, "ddim": 4}.get(tag, 9)
        for i, key in enumerate(self.keys):
            c = self.counters.get((key, tag_id), 0)
            self.counters[(key, tag_id)] = c + 1
            g = torch.Generator(device=self.device)
            g.manual_seed((self.seed * 1000003 + key * 8191 + tag_id * 131 + c) % (2 ** 63 - 1))
            out[i * per:(i + 1) * per] = torch.randn((per, *shape[1:]), device=self.device, generator=g)
        return out.clamp_(-self.clip, self.clip) if tag != "excluded" else out


def run_sharded(n_windows: int, sample_batch: Callable[[List[int]], torch.Tensor], max_batch: Optional[int],
                group=None) -> torch.Tensor:
    """Runs sample_batch(ids) -> [len(ids), ...] over this rank's windows in chunks of <= max_batch and returns
    all n_windows results on every rank."""
    world, rank = world_info(group)
    mine = shard_windows(n_windows, world, rank)
    mb = max_batch or max(len(mine), 1)
    outs = [sample_batch(mine[i:i + mb]) for i in range(0, len(mine), mb)]
    if outs:
        local = torch.cat(outs, 0)
    else:  # more ranks than windows: contribute an empty shard of the right trailing shape
        probe = sample_batch([])
        local = probe
    return gather_windows(local, n_windows, group)


def allreduce_mean_(flat: torch.Tensor, group=None, bucket_numel: int = 1 << 26) -> torch.Tensor:
    """Data-parallel gradient averaging of ONE flat buffer (trainer.DiT3DTrainer.grads): in-place sum over the ranks in buckets
    of `bucket_numel` elements issued back to back (async) and awaited together, then scaled by 1/world.  xGMI is point-to-point,
    so a ring all-reduce is bound per link: few large buckets (256 MiB of fp32 by default) rather than per-tensor calls."""
    world, _ = world_info(group)
    if world == 1:
        return flat
    works = []
    for lo in range(0, flat.numel(), bucket_numel):
        works.append(dist.all_reduce(flat[lo: lo + bucket_numel], group=group, async_op=True))
    for w in works:
        w.wait()
    flat.mul_(1.0 / world)
    return flat


def _avg_supported(group=None) -> bool:
    """RCCL (backend "nccl") reduces with ReduceOp.AVG in the collective itself; gloo only sums"""
    try:
        return dist.get_backend(group) == "nccl"
    except Exception:
        return False


class OverlappedGradReducer:
    """Gradient averaging OVERLAPPED with the backward pass: gradients are handed over as the backward produces them (deepest
    levels last).  Each is copied ONCE into its slot of the trainer's flat gradient buffer (the copy every path needs), the filled
    slots are merged into contiguous ranges, and a range that reaches `bucket_numel` elements is all-reduced IN PLACE (a view of the
    flat buffer, async) while the backward of the next blocks runs; `finish()` reduces the remaining ranges and waits.  No bucket
    packing (`torch.cat`) and no copy-back: the only extra pass over the gradients is the 1/world scale, and over RCCL not even that
    (ReduceOp.AVG).  Few large ranges (xGMI is point-to-point: a ring all-reduce is bound per link).  Per element the summation is
    that of one all-reduce; against `allreduce_mean_` (SUM, then one multiply by 1/world) the mean can differ in the last bit when the
    collective averages itself (ReduceOp.AVG) and the world is not a power of two -- every rank still holds the same bits."""

    def __init__(self, bucket_numel: int = 1 << 25, group=None, flat: Optional[torch.Tensor] = None):
        self.bucket_numel, self.group = int(bucket_numel), group
        self.world, _ = world_info(group)
        self.flat = flat                     # the flat buffer every `dest` is a view of (found from the first dest when None)
        self._copies: List[tuple] = []       # (dest view, source gradient) not yet copied
        self._ranges: List[List[int]] = []   # filled, not yet reduced element ranges [lo, hi) of `flat`, sorted and merged
        self._inflight: List[tuple] = []     # (work, lo, hi)
        self._avg = self.world > 1 and _avg_supported(group)
        self.calls = 0                       # collectives issued (tests / reports)

    def _offset(self, dest: torch.Tensor) -> int:
        if self.flat is None:
            st = dest.untyped_storage()
            self.flat = torch.empty(0, dtype=dest.dtype, device=dest.device).set_(st, 0, (st.nbytes() // dest.element_size(),))
        off = (dest.data_ptr() - self.flat.data_ptr()) // dest.element_size()
        if off < 0 or off + dest.numel() > self.flat.numel() or not dest.is_contiguous():
            raise ValueError("OverlappedGradReducer: dest must be a contiguous view of the flat gradient buffer")
        return off

    def add(self, dest: torch.Tensor, grad: torch.Tensor) -> None:
        """dest: flat view that must finally hold the mean of `grad` over the ranks"""
        self._copies.append((dest, grad.reshape(-1)))
        if self.world == 1:
            return
        lo = self._offset(dest)
        self._insert(lo, lo + dest.numel())
        if max(hi - lo_ for lo_, hi in self._ranges) >= self.bucket_numel:
            self.flush(final=False)

    def _insert(self, lo: int, hi: int) -> None:
        import bisect
        r = self._ranges
        i = bisect.bisect_left(r, [lo, hi])
        r.insert(i, [lo, hi])
        if i + 1 < len(r) and r[i + 1][0] <= r[i][1]:
            r[i][1] = max(r[i][1], r[i + 1][1])
            del r[i + 1]
        if i > 0 and r[i][0] <= r[i - 1][1]:
            r[i - 1][1] = max(r[i - 1][1], r[i][1])
            del r[i]

    def _copy_pending(self) -> None:
        if self._copies:  # one multi-tensor copy for everything handed over since the last flush
            torch._foreach_copy_([d for d, _ in self._copies], [g for _, g in self._copies])
            self._copies = []

    def flush(self, final: bool = True) -> None:
        """all-reduce the filled ranges (all of them when `final`, else only those that reached the bucket size)"""
        self._copy_pending()
        if self.world == 1:
            return
        keep = []
        for lo, hi in self._ranges:
            if final or hi - lo >= self.bucket_numel:
                op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
                work = dist.all_reduce(self.flat[lo:hi], op=op, group=self.group, async_op=True)
                self._inflight.append((work, lo, hi))
                self.calls += 1
            else:
                keep.append([lo, hi])
        self._ranges = keep

    def finish(self) -> None:
        self.flush(final=True)
        for work, lo, hi in self._inflight:
            work.wait()
            if not self._avg:
                self.flat[lo:hi].mul_(1.0 / self.world)
        self._inflight = []


_BRANCH_GROUPS = {}


def branch_group(nfe: int, group=None):
    """The `nfe`-rank sub-group this rank exchanges History-Guidance branches in: ranks [g*nfe, (g+1)*nfe) form group g, so that with
    8 ranks and 2 branches four PAIRS each exchange 2 x v (instead of one all-gather over 8 ranks, of which 6 carried redundant
    copies).  The world must divide into such groups (the sampler runs a step unsplit on every rank otherwise: leftover ranks would
    evaluate another model batch than the grouped ones, and kernel choices -- hence bits -- follow the batch).  Collective on first
    use: EVERY rank of the world must call it with the same nfe (dist.new_group is a world-wide call); the sampler does so at the top
    of each replicated window, before its step loop."""
    world, rank = world_info(group)
    if group is not None:
        raise ValueError("branch_group: nested groups are not supported")
    if nfe < 1 or world % nfe != 0:
        raise ValueError(f"branch_group: {world} ranks do not divide into groups of {nfe}")
    key = (nfe, world)
    if key not in _BRANCH_GROUPS:
        groups = []
        for g in range(world // nfe):
            ranks = list(range(g * nfe, (g + 1) * nfe))
            groups.append(dist.new_group(ranks) if world > nfe else None)  # world == nfe: the default group is the one group
        _BRANCH_GROUPS[key] = groups
    return _BRANCH_GROUPS[key][rank // nfe]


def exchange_branches(v_local: torch.Tensor, nfe: int, group=None) -> torch.Tensor:
    """History-Guidance branch parallelism for the sequential key-frame windows (the branches of a step are
    independent forwards combined only by compose, history_guidance.py:545-568,978-982).  Rank r has evaluated branch r % nfe of
    every sample: v_local [B, T, ...].  One all-gather inside the rank's `nfe`-rank sub-group (`branch_group`) returns
    v [B * nfe, T, ...] in the sampler's (sample, branch) row order; every sub-group computes the same result."""
    world, rank = world_info(group)
    if world < nfe:
        raise ValueError(f"branch parallelism needs at least {nfe} ranks, have {world}")
    sub = branch_group(nfe, group)
    out = v_local.new_empty((nfe * v_local.shape[0], *v_local.shape[1:]))
    dist.all_gather_into_tensor(out, v_local.contiguous(), group=sub)
    b = v_local.shape[0]
    return out.view(nfe, b, *v_local.shape[1:]).transpose(0, 1).reshape(b * nfe, *v_local.shape[1:]).contiguous()
