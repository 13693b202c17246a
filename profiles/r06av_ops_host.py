"""Is the 4096 x 513 ops leg kernel time or host time?  Per call: HIP events around 10 and around 200 back-to-back calls, host
wall clock per call without synchronisation, and the same ops through preallocated outputs where the wrapper allows it."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
cpp = tn.cpp
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev)
tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
field = torch.randn(64, len(pts), device=dev)
for R, S, rays in ((4096, 513, scenes.outside_in_rays(4096, 1)), (65536, 256, scenes.outside_in_rays(65536, 3))):
    o, d = (torch.from_numpy(x).to(dev) for x in rays)
    out = tr.trace_rays(o, d, 512, compact_rows=True)
    lists = [out[k].contiguous() for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates", "hit_distances", "vertex_indices")]
    nv = lists[0].long()
    near = lists[3][:, 0, 0][:, None]
    far = torch.gather(lists[3][:, :, 1], 1, (nv[:, None] - 1).clamp_min(0))
    ts = torch.linspace(0.0, 1.0, S, device=dev)[None]
    dist = (near * (1 - ts) + far * ts).contiguous()
    traced = tr.find_visited_cells(*lists, dist)
    vi, bc = traced["vertex_indices"], traced["barycentric_coordinates"]
    g = torch.randn(R, S, 64, device=dev)
    for name, f in (("find_visited_cells", lambda: tr.find_visited_cells(*lists, dist)),
                    ("interpolate_values", lambda: cpp.interpolate_values(vi, bc, field)),
                    ("interpolate_values_backward", lambda: cpp.interpolate_values_backward(vi, bc, field, g))):
        f(); torch.cuda.synchronize()
        res = []
        for n in (10, 200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            res.append(f"{n} calls: {e0.elapsed_time(e1) / n * 1e3:7.1f} us per call by events, host issue {1e6 * (t1 - t0) / n:6.1f} us per call")
        print(f"{R}x{S} {name:28s}: " + " | ".join(res), flush=True)
