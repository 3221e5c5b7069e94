"""world_size-2 gloo test (CPU) of the window sharding used for the 200-frame plan: round-robin shards + one
all-gather per stage reproduce the serial result exactly, including the ragged case (odd window counts) and the
case of more ranks than windows."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _window_result(ids):
    # stand-in for one sampled window: deterministic function of the window id only
    if not ids:
        return torch.zeros(0, 8, 3, 4, 4)
    return torch.stack([torch.full((8, 3, 4, 4), float(i)) + torch.arange(8).view(8, 1, 1, 1) * 0.01 for i in ids])


def _worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dfot_amd import parallel
    ok = True
    for n in counts:
        out = parallel.run_sharded(n, _window_result, max_batch=4)
        ref = _window_result(list(range(n)))
        ok = ok and torch.equal(out, ref)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_windows_equal_serial_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, [11, 35, 1, 2], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def test_shard_assignment_covers_every_window_once():
    from dfot_amd import parallel
    for n in (1, 2, 11, 35, 48):
        for world in (1, 2, 4, 8):
            ids = sorted(i for r in range(world) for i in parallel.shard_windows(n, world, r))
            assert ids == list(range(n))
            assert max(len(parallel.shard_windows(n, world, r)) for r in range(world)) == -(-n // world)
