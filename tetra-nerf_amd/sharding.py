"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed).

Rays are independent units: mesh, acceleration structures and the feature field are replicated
per GPU exactly as the reference's DDP setup replicates the model
(/root/reference/tetranerf/nerfstudio/pipeline.py:53-58; the tracer is per device,
model.py:394-407).  Rank k traces the contiguous slice [k*R/W, (k+1)*R/W) of the ray array --
no halo and no exchange on the trace path.  The only collective is the all-gather of RENDERED
tiles (rgb [R/W,3] + accumulation/depth [R/W,1]: 0.96 MB + 0.64 MB per rank for an 800x800 frame
at W = 8), latency-bound on xGMI.  Backend "nccl" is RCCL on ROCm; the same code runs on "gloo"
(CPU tensors), which is how the world_size-2 tests exercise it without GPUs.
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
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if world * per == num_rows_total:
        return out
    pieces = []
    for r in range(world):
        a, b = shard_range(num_rows_total, r, world)
        pieces.append(out[r * per: r * per + (b - a)])
    return torch.cat(pieces, 0)


def gather_rendered(local: Dict[str, torch.Tensor], num_rays_total: int, group=None) -> Dict[str, torch.Tensor]:
    """All-gather the per-ray outputs of a render ({'rgb': [r,3], 'accumulation': [r,1], ...})
    as ONE collective: the tensors are concatenated along the feature axis first."""
    keys = sorted(local)
    widths = [int(local[k].reshape(local[k].shape[0], -1).shape[1]) for k in keys]
    flat = torch.cat([local[k].reshape(local[k].shape[0], -1).to(torch.float32) for k in keys], 1)
    full = all_gather_rows(flat, num_rays_total, group)
    out, c = {}, 0
    for k, w in zip(keys, widths):
        out[k] = full[:, c:c + w].reshape((num_rays_total,) + tuple(local[k].shape[1:])).to(local[k].dtype)
        c += w
    return out


def render_sharded(render_fn, origins: torch.Tensor, directions: torch.Tensor, group=None, chunk: int = 65536,
                   timings: dict | None = None) -> Dict[str, torch.Tensor]:
    """The multi-GPU render flow as one call: every rank holds the frame's rays (and a replicated tracer / field /
    MLP behind `render_fn(origins, directions) -> {'rgb','accumulation','depth','ray_mask'}`, e.g.
    TetraRenderer.render), renders ITS contiguous slice in nerfstudio-sized chunks and the slices are all-gathered
    in one collective (RCCL over xGMI with backend "nccl"; reference analogue: the DDP wrapper of
    pipeline.py:53-58 replicates the model, each rank works on its own rays).  Returns the full frame on every
    rank.  `timings`, if given, receives the seconds spent in 'render' and 'all_gather' (device-synchronised)."""
    import time

    import torch.distributed as dist

    R = origins.shape[0]
    world = rank = 1
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_range(R, rank, world)
    sync = (lambda: torch.cuda.synchronize(origins.device)) if origins.is_cuda else (lambda: None)
    sync()
    t0 = time.perf_counter()
    parts = [render_fn(origins[s:min(s + chunk, hi)].contiguous(), directions[s:min(s + chunk, hi)].contiguous())
             for s in range(lo, hi, chunk)]
    keys = ("rgb", "accumulation", "depth", "ray_mask")
    if parts:
        local = {k: torch.cat([p[k] for p in parts], 0) for k in keys}
    else:  # more ranks than rays
        z = origins.new_zeros
        local = {"rgb": z((0, 3)), "accumulation": z((0, 1)), "depth": z((0, 1)), "ray_mask": z((0,), dtype=torch.bool)}
    sync()
    t1 = time.perf_counter()
    full = gather_rendered({k: (v[:, None] if v.dim() == 1 else v) for k, v in local.items()}, R, group)
    full["ray_mask"] = full["ray_mask"][:, 0]
    sync()
    if timings is not None:
        timings["render"] = t1 - t0
        timings["all_gather"] = time.perf_counter() - t1
    return full


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX-reduce a host scalar (the bench's elapsed time) over all ranks."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(values, device=None, group=None):
    """SUM-reduce a list of host scalars over all ranks (units processed by the whole job)."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [float(x) for x in t.tolist()]
