"""Multi-process tests of the N>1 path on CPU (gloo, world_size 2 and 3): ray sharding,
the single all-gather of rendered tiles, and the bench's max/sum reductions."""
import importlib
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_rays, q):
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = importlib.import_module("tetra-nerf_amd.sharding")
        lo, hi = sh.shard_range(num_rays, rank, world)
        idx = torch.arange(lo, hi, dtype=torch.float32)
        # a deterministic stand-in for the per-ray render outputs
        local = {"rgb": torch.stack([idx, idx * 2, idx * 3], -1), "accumulation": (idx % 7)[:, None],
                 "depth": (idx * 0.5)[:, None]}
        full = sh.gather_rendered(local, num_rays)
        ref = torch.arange(num_rays, dtype=torch.float32)
        ok = (torch.equal(full["rgb"], torch.stack([ref, ref * 2, ref * 3], -1))
              and torch.equal(full["accumulation"], (ref % 7)[:, None])
              and torch.equal(full["depth"], (ref * 0.5)[:, None]))
        tmax = sh.max_over_ranks(1.0 + rank)
        tot = sh.sum_over_ranks([hi - lo, 10.0 * (rank + 1)])
        q.put((rank, ok, tmax, tot, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,num_rays", [(2, 4096), (2, 4097), (3, 1000)])
def test_shard_and_gather_gloo(world, num_rays):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    covered = 0
    for rank, ok, tmax, tot, (lo, hi) in res:
        assert ok, f"rank {rank}: gathered render differs"
        assert tmax == float(world)
        assert tot[0] == float(num_rays) and tot[1] == 10.0 * world * (world + 1) / 2
        assert lo == covered
        covered = hi
    assert covered == num_rays


def test_shard_range_properties():
    sh = importlib.import_module("tetra-nerf_amd.sharding")
    for R in (0, 1, 7, 4096, 640000):
        for W in (1, 2, 3, 8):
            spans = [sh.shard_range(R, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == R
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sh.shard_range(10, 3, 2)


def test_single_process_passthrough():
    sh = importlib.import_module("tetra-nerf_amd.sharding")
    x = {"rgb": torch.rand(5, 3)}
    assert torch.equal(sh.gather_rendered(x, 5)["rgb"], x["rgb"])
    assert sh.max_over_ranks(2.5) == 2.5
    assert sh.sum_over_ranks([1, 2]) == [1.0, 2.0]
