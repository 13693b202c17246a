#!/usr/bin/env python3
"""Small batches on the BVH path (below walk_min_rays): LDS hit arrays sized for the mesh (option small_lds, overflow rays
in a second launch) vs the full M-entry arrays.  4096-ray training batches of the C4 mesh (and the C2 / C5 meshes), M = 512;
outputs compared bit for bit."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cfg, npts, seed in (("c2", 15000, 0), ("c4", 45000, 2), ("c5", 150000, 3)):
    pts, cells = scenes.random_mesh(npts, seed)
    ref_out = {}
    for lw in (16, 32, 64):
      tr = tn.TetrahedraTracer(dev); tr.set_option("leaf_width", lw); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
      tr.set_option("gdebug", 8); tr.trace_rays(*[torch.from_numpy(x).to(dev) for x in scenes.outside_in_rays(4096, 1)], M); torch.cuda.synchronize(); tr.set_option("gdebug", 0)
      print(f"{cfg} leaf width {lw}:", flush=True)
      for name, (o, d) in (("outside-in", scenes.outside_in_rays(4096, 1)), ("inside-out", scenes.inside_out_rays(4096, 2))):
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        res, outs = {}, {}
        for sl in (1, 0, 1, 0):
            tr.set_option("small_lds", sl)
            outs[sl] = tr.trace_rays(o, d, M)
            def call():
                x = tr.trace_rays(o, d, M); del x
            res.setdefault(sl, []).append(timed(call))
        for k in KEYS: assert torch.equal(outs[0][k], outs[1][k]), (cfg, name, k)
        if lw == 16: ref_out[name] = outs[1]
        else:
            for k in KEYS: assert torch.equal(ref_out[name][k], outs[1][k]), (cfg, name, k, lw)
        n = int(outs[1]["num_visited_cells"].sum()); mx = int(outs[1]["num_visited_cells"].max())
        print(f"  {cfg} 4096 rays {name}: mesh-sized LDS arrays {min(res[1]):.3f} ms, full arrays {min(res[0]):.3f} ms  ({n} segments, max {mx} per ray) {tr.trace_stats()}", flush=True)
      del tr
