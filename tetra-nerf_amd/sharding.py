"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed).

Rays are independent units: mesh, acceleration structures and the feature field are replicated
per GPU exactly as the reference's DDP setup replicates the model
(/root/reference/tetranerf/nerfstudio/pipeline.py:53-58; the tracer is per device,
model.py:394-407).  No halo and no exchange on the trace path.  The only collective is the
all-gather of RENDERED tiles (rgb [R/W,3] + accumulation/depth [R/W,1]: 0.96 MB + 0.64 MB per rank
for an 800x800 frame at W = 8), latency-bound on xGMI.  Backend "nccl" is RCCL on ROCm; the same
code runs on "gloo" (CPU tensors), which is how the world_size-2/3 tests exercise it without GPUs.

Two partitions:
  * `shard_range`  -- rank k owns the contiguous slice [k*R/W, (k+1)*R/W): what the weak-scaling trace
    leg of bench.py uses (every rank has its own frame there).
  * `deal_tiles`   -- tiles of 4096 consecutive rays (a nerfstudio chunk; 5 rows of an 800-wide frame)
    are dealt ROUND-ROBIN to the ranks.  This is what `render_sharded` uses: the cost of a
    ray is its MLP samples, i.e. whether it hits the mesh at all -- on the bench frame 38 % of the rays
    miss and they are the top and bottom image rows, so contiguous slices leave the outer ranks idle
    (~60 % efficiency at W = 8) while dealt tiles give every rank the same share of every image region
    (hitting rays per rank within a few per cent: tests/test_sharding.py).  One all-gather, then the
    inverse permutation restores ray order.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def shard_range(num_rays: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one ray."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(num_rays, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


TILE_RAYS = 4096   # rays per dealt tile = the reference's eval/train chunk (registration.py:33-34)


def deal_tiles(num_rays: int, rank: int, world_size: int, tile: int = TILE_RAYS) -> torch.Tensor:
    """int64 indices of the rays rank `rank` renders: tiles t = rank, rank + W, rank + 2W, ... of `tile`
    consecutive rays each (the last tile may be short), in ascending order."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    if tile <= 0:
        raise ValueError("tile must be positive")
    ntiles = -(-num_rays // tile)
    mine = torch.arange(rank, max(ntiles, rank), world_size, dtype=torch.int64)   # (empty when rank >= ntiles)
    idx = (mine[:, None] * tile + torch.arange(tile, dtype=torch.int64)[None]).reshape(-1)
    return idx[idx < num_rays]


def dealt_capacity(num_rays: int, world_size: int, tile: int = TILE_RAYS) -> int:
    """Rows every rank contributes to the all-gather: an upper bound of the largest dealt share (less than one tile
    above it), so that ONE all_gather_into_tensor (equal blocks) carries the frame."""
    ntiles = -(-num_rays // tile)
    return min(num_rays, -(-ntiles // world_size) * tile)


def undeal_index(num_rays: int, world_size: int, tile: int = TILE_RAYS) -> torch.Tensor:
    """int64 [num_rays]: position of ray r in the gathered buffer [world_size * dealt_capacity, ...]."""
    per = dealt_capacity(num_rays, world_size, tile)
    pos = torch.empty(num_rays, dtype=torch.int64)
    for r in range(world_size):
        idx = deal_tiles(num_rays, r, world_size, tile)
        pos[idx] = r * per + torch.arange(len(idx), dtype=torch.int64)
    return pos


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    """dist.all_gather_into_tensor; device tensors on a "gloo" group (the one-GPU test mode of bench.py / the GPU tests:
    N processes sharing one device, where RCCL refuses duplicate devices) are staged through the host."""
    import torch.distributed as dist

    if inp.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _all_reduce(t: torch.Tensor, op, group=None) -> torch.Tensor:
    import torch.distributed as dist

    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        return h
    dist.all_reduce(t, op=op, group=group)
    return t


def all_gather_dealt(local: torch.Tensor, num_rows_total: int, group=None, tile: int = TILE_RAYS) -> torch.Tensor:
    """All-gather rows partitioned by `deal_tiles` into the full [num_rows_total, ...] tensor in ray order on every
    rank: ONE all_gather_into_tensor of equal (padded) blocks + one index_select with the inverse permutation."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        assert local.shape[0] == num_rows_total
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = deal_tiles(num_rows_total, rank, world, tile)
    if local.shape[0] != len(mine):
        raise ValueError(f"rank {rank}: local rows {local.shape[0]} != dealt share {len(mine)}")
    per = dealt_capacity(num_rows_total, world, tile)
    padded = local
    if local.shape[0] != per:
        padded = local.new_zeros((per,) + tuple(local.shape[1:]))
        padded[: local.shape[0]] = local
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    _all_gather_into(out, padded.contiguous(), group)
    return out.index_select(0, undeal_index(num_rows_total, world, tile).to(out.device))


def all_gather_rows(local: torch.Tensor, num_rows_total: int, group=None) -> torch.Tensor:
    """All-gather row blocks that follow `shard_range` (uneven by at most one row) into the full
    [num_rows_total, ...] tensor on every rank.  One collective: shards are padded to the
    largest shard so that all_gather_into_tensor (a single RCCL kernel) can be used."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        assert local.shape[0] == num_rows_total
        return local
    # (a one-rank group still goes through the collective: the same code path as N ranks)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(num_rows_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: local rows {local.shape[0]} != shard size {hi - lo}")
    per = -(-num_rows_total // world)
    tail = tuple(local.shape[1:])
    padded = local
    if local.shape[0] != per:
        padded = local.new_zeros((per,) + tail)
        padded[: local.shape[0]] = local
    out = local.new_empty((world * per,) + tail)
    _all_gather_into(out, padded.contiguous(), group)
    if world * per == num_rows_total:
        return out
    pieces = []
    for r in range(world):
        a, b = shard_range(num_rows_total, r, world)
        pieces.append(out[r * per: r * per + (b - a)])
    return torch.cat(pieces, 0)


def gather_rendered(local: Dict[str, torch.Tensor], num_rays_total: int, group=None, tile: int = 0) -> Dict[str, torch.Tensor]:
    """All-gather the per-ray outputs of a render ({'rgb': [r,3], 'accumulation': [r,1], ...})
    as ONE collective: the tensors are concatenated along the feature axis first.  tile = 0: the local rows are a
    contiguous `shard_range` slice; tile > 0: they are the rank's `deal_tiles` share."""
    keys = sorted(local)
    import math

    widths = [math.prod(local[k].shape[1:]) for k in keys]      # (a rank's share may be empty: no reshape(-1) on it)
    flat = torch.cat([local[k].reshape(local[k].shape[0], w).to(torch.float32) for k, w in zip(keys, widths)], 1)
    full = all_gather_dealt(flat, num_rays_total, group, tile) if tile else all_gather_rows(flat, num_rays_total, group)
    out, c = {}, 0
    for k, w in zip(keys, widths):
        out[k] = full[:, c:c + w].reshape((num_rays_total,) + tuple(local[k].shape[1:])).to(local[k].dtype)
        c += w
    return out


def render_sharded(render_fn, origins: torch.Tensor, directions: torch.Tensor, group=None, chunk: int = 65536,
                   timings: dict | None = None, tile: int = TILE_RAYS) -> Dict[str, torch.Tensor]:
    """The multi-GPU render flow as one call: every rank holds the frame's rays (and a replicated tracer / field /
    MLP behind `render_fn(origins, directions) -> {'rgb','accumulation','depth','ray_mask'}`, e.g.
    TetraRenderer.render), renders ITS share -- tiles of `tile` rays dealt round-robin (`deal_tiles`; tile = 0: one
    contiguous slice) -- in chunks of `chunk` rays, and the shares are all-gathered in one collective (RCCL over xGMI
    with backend "nccl"; reference analogue: the DDP wrapper of pipeline.py:53-58 replicates the model, each rank works
    on its own rays) and put back into ray order.  Returns the full frame on every rank.  `timings`, if given, receives
    the seconds spent in 'render' and 'all_gather' (device-synchronised) and 'hitting_rays' of this rank."""
    import time

    import torch.distributed as dist

    R = origins.shape[0]
    grouped = dist.is_available() and dist.is_initialized()
    if grouped:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    sync = (lambda: torch.cuda.synchronize(origins.device)) if origins.is_cuda else (lambda: None)
    sync()
    t0 = time.perf_counter()
    if tile and grouped:   # (a one-rank group takes the same deal -> index_select -> all-gather -> un-deal path as N ranks)
        mine = deal_tiles(R, rank, world, tile).to(origins.device)
        o_mine, d_mine = origins.index_select(0, mine), directions.index_select(0, mine)
    else:
        lo, hi = shard_range(R, rank, world)
        o_mine, d_mine = origins[lo:hi], directions[lo:hi]
        tile = 0
    n = o_mine.shape[0]
    parts = [render_fn(o_mine[s:s + chunk].contiguous(), d_mine[s:s + chunk].contiguous()) for s in range(0, n, chunk)]
    keys = ("rgb", "accumulation", "depth", "ray_mask")
    if parts:
        local = {k: torch.cat([p[k] for p in parts], 0) for k in keys}
    else:  # more ranks than tiles
        z = origins.new_zeros
        local = {"rgb": z((0, 3)), "accumulation": z((0, 1)), "depth": z((0, 1)), "ray_mask": z((0,), dtype=torch.bool)}
    sync()
    t1 = time.perf_counter()
    full = gather_rendered({k: (v[:, None] if v.dim() == 1 else v) for k, v in local.items()}, R, group, tile)
    full["ray_mask"] = full["ray_mask"][:, 0]
    sync()
    if timings is not None:
        timings["render"] = t1 - t0
        timings["all_gather"] = time.perf_counter() - t1
        timings["hitting_rays"] = int(local["ray_mask"].sum())
    return full


def gather_scalars(value: float, device=None, group=None):
    """[world_size] list of one host scalar per rank (per-rank render times / hitting rays in the bench line)."""
    import torch.distributed as dist

    # (a ONE-rank group still runs the collective: on backend "nccl" that is an RCCL kernel -- the GPU suite's only way to
    #  execute this code on the real backend; without a process group there is nothing to reduce over)
    if not dist.is_available() or not dist.is_initialized():
        return [float(value)]
    world = dist.get_world_size(group)
    out = torch.zeros(world, dtype=torch.float64, device=device)
    _all_gather_into(out, torch.tensor([value], dtype=torch.float64, device=device), group)
    return [float(x) for x in out.tolist()]


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX-reduce a host scalar (the bench's elapsed time) over all ranks."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return float(value)
    t = _all_reduce(torch.tensor([value], dtype=torch.float64, device=device), dist.ReduceOp.MAX, group)
    return float(t.item())


def sum_over_ranks(values, device=None, group=None):
    """SUM-reduce a list of host scalars over all ranks (units processed by the whole job)."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return [float(v) for v in values]
    t = _all_reduce(torch.tensor(list(values), dtype=torch.float64, device=device), dist.ReduceOp.SUM, group)
    return [float(x) for x in t.tolist()]
