#!/usr/bin/env python3
"""Stand-alone find_visited_cells / interpolate_values (fwd, bwd) against their HBM rooflines (SURVEY.md 8d:
37 B per sample for the match, 284 B per sample for the gather) at the C4 training-batch size and at frame size."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
V = len(pts)
field = torch.randn(64, V, device=dev, requires_grad=True)
for R, S in ((4096, 256), (4096, 513), (65536, 256)):
    o, d = scenes.outside_in_rays(R, 1)
    out = tr.trace_rays(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 512)
    n = out["num_visited_cells"]
    near = out["hit_distances"][:, 0, 0]; far = torch.gather(out["hit_distances"][:, :, 1], 1, (n[:, None].long() - 1).clamp_min(0))[:, 0]
    ts = torch.linspace(0, 1, S, device=dev)[None]
    dist = (near[:, None] * (1 - ts) + far[:, None] * ts).contiguous()
    args = (out["num_visited_cells"], out["visited_cells"], out["barycentric_coordinates"], out["hit_distances"], out["vertex_indices"], dist)
    ms = timeit(lambda: tr.find_visited_cells(*args))
    N = R * S
    print(f"find_visited_cells {R}x{S}: {ms*1e3:.1f} us = {N*37/ms/1e9:.2f} TB/s of 37 B/sample ({N*37/ms/1e9/8*100:.0f} % of 8 TB/s)", flush=True)
    m = tr.find_visited_cells(*args)
    vi, bc = m["vertex_indices"], m["barycentric_coordinates"]
    with torch.no_grad():
        ms = timeit(lambda: tn.cpp.interpolate_values(vi, bc, field))
    print(f"interpolate_values fwd {N} samples: {ms*1e3:.1f} us = {N*284/ms/1e9:.2f} TB/s of 284 B/sample ({N*284/ms/1e9/8*100:.0f} % of 8 TB/s)", flush=True)
    g = torch.randn(R, S, 64, device=dev)
    ms = timeit(lambda: tn.cpp.interpolate_values_backward(vi, bc, field.detach(), g))
    print(f"interpolate_values bwd {N} samples: {ms*1e3:.1f} us ({N*284/ms/1e9:.2f} TB/s of the forward's bytes)", flush=True)
    del out, m, g
