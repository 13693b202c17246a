"""In-process A/B of the stand-alone gather kernels (interpolate_values forward): round 3-5's register-carry kernel
(TETRANERF_HIP_GATHER=carry) against round 6's straight-line kernel, on the two shapes of bench.py's C4_ops leg; bit-identical
outputs; also times find_visited_cells and the gather's adjoint as they are."""
import importlib
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
cpp = tn.cpp
M = 512
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev)
tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
torch.manual_seed(0)
field = torch.randn(64, len(pts), device=dev)
fo, fd = bench.frame_rays(scenes, 0, 800, 800)
bo, bd = scenes.outside_in_rays(4096, 1)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, (o_np, d_np), S, want in (("4096x513", (bo, bd), 513, 4096), ("65536x256", (fo, fd), 256, 65536)):
    o, d = torch.from_numpy(o_np).to(dev), torch.from_numpy(d_np).to(dev)
    out = tr.trace_rays(o, d, M, compact_rows=True)
    idx = torch.nonzero(out["num_visited_cells"] > 0)[:, 0][:want]
    lists = [out[k][idx].contiguous() for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates", "hit_distances", "vertex_indices")]
    del out
    R = int(idx.numel())
    nv = lists[0].long()
    near = lists[3][:, 0, 0][:, None]
    far = torch.gather(lists[3][:, :, 1], 1, (nv[:, None] - 1).clamp_min(0))
    ts = torch.linspace(0.0, 1.0, S, device=dev)[None]
    dist = (near * (1 - ts) + far * ts).contiguous()
    n = R * S
    traced = tr.find_visited_cells(*lists, dist)
    vi, bc = traced["vertex_indices"], traced["barycentric_coordinates"]
    res = {}
    outs = {}
    for rnd in range(5):
        for mode in ("carry", "straight"):
            os.environ["TETRANERF_HIP_GATHER"] = mode
            res.setdefault(mode, []).append(timed(lambda: cpp.interpolate_values(vi, bc, field)))
            if rnd == 0:
                outs[mode] = cpp.interpolate_values(vi, bc, field).clone()
    same = torch.equal(outs["carry"].view(torch.int32), outs["straight"].view(torch.int32))
    g = torch.randn(R, S, 64, device=dev)
    ms_b = timed(lambda: cpp.interpolate_values_backward(vi, bc, field, g))
    ms_m = timed(lambda: tr.find_visited_cells(*lists, dist))
    print(f"== {name}: {n} samples, matched {float(traced['mask'].float().mean()):.3f}")
    for mode in ("carry", "straight"):
        ms = sorted(res[mode])
        med = ms[len(ms) // 2]
        print(f"  interpolate_values {mode:9s} median {med * 1e3:7.1f} us  min {ms[0] * 1e3:7.1f}  = {n * 284 / (med * 1e-3) / 1e9 / 8000:.3f} of 8 TB/s  identical: {same}")
    print(f"  backward {ms_b * 1e3:7.1f} us = {n * 284 / (ms_b * 1e-3) / 1e9 / 8000:.3f};  find_visited_cells {ms_m * 1e3:7.1f} us = {n * 37 / (ms_m * 1e-3) / 1e9 / 8000:.3f}")
    del lists, traced, vi, bc, g, dist
    torch.cuda.empty_cache()
