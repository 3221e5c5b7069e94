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


class OverlappedGradReducer:
    """Gradient averaging OVERLAPPED with the backward pass: gradients are handed over as the backward produces
    them (deepest levels last), packed into buckets of `bucket_numel` elements and all-reduced asynchronously while the backward of
    the next blocks runs; `finish()` waits, scales by 1/world and scatters the means into the destination views (the trainer's flat
    gradient buffer).  Few large buckets (xGMI is point-to-point: a ring all-reduce is bound per link).  Summation order per
    element is that of one all-reduce, so the result equals `allreduce_mean_` on the whole flat buffer bit for bit."""

    def __init__(self, bucket_numel: int = 1 << 25, group=None):
        self.bucket_numel, self.group = int(bucket_numel), group
        self.world, _ = world_info(group)
        self._pending: List[tuple] = []   # (dest view, source gradient) of the bucket being filled
        self._pending_numel = 0
        self._inflight: List[tuple] = []  # (work, bucket buffer, [(dest, numel)])

    def add(self, dest: torch.Tensor, grad: torch.Tensor) -> None:
        """dest: flat view that must finally hold the mean of `grad` over the ranks"""
        if self.world == 1:
            dest.copy_(grad.reshape(-1))
            return
        self._pending.append((dest, grad))
        self._pending_numel += grad.numel()
        if self._pending_numel >= self.bucket_numel:
            self.flush()

    def flush(self) -> None:
        if not self._pending:
            return
        buf = torch.cat([g.reshape(-1).to(torch.float32) for _, g in self._pending])
        work = dist.all_reduce(buf, group=self.group, async_op=True)
        self._inflight.append((work, buf, [(d, g.numel()) for d, g in self._pending]))
        self._pending, self._pending_numel = [], 0

    def finish(self) -> None:
        self.flush()
        for work, buf, dests in self._inflight:
            work.wait()
            buf.mul_(1.0 / self.world)
            off = 0
            for d, n in dests:
                d.copy_(buf[off: off + n])
                off += n
        self._inflight = []


def exchange_branches(v_local: torch.Tensor, nfe: int, group=None) -> torch.Tensor:
    """History-Guidance branch parallelism for the sequential key-frame windows (the branches of a step are
    independent forwards combined only by compose, history_guidance.py:545-568,978-982).  Rank r has evaluated branch r % nfe of
    every sample: v_local [B, T, ...].  One all-gather returns v [B * nfe, T, ...] in the sampler's (sample, branch) row order."""
    world, rank = world_info(group)
    if world < nfe:
        raise ValueError(f"branch parallelism needs at least {nfe} ranks, have {world}")
    out = v_local.new_empty((world * v_local.shape[0], *v_local.shape[1:]))
    dist.all_gather_into_tensor(out, v_local.contiguous(), group=group)
    b = v_local.shape[0]
    per_rank = out.view(world, b, *v_local.shape[1:])[:nfe]          # ranks 0..nfe-1 hold branches 0..nfe-1
    return per_rank.transpose(0, 1).reshape(b * nfe, *v_local.shape[1:]).contiguous()
