#!/usr/bin/env python3
"""4096-ray training batches of the C4 mesh (compact rows, as render_train traces them): the default small-batch BVH path
against the adjacency walk forced (option walk = 2), medians of 5 x 50 calls each, interleaved."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
def timed(o, d, compact):
    for _ in range(5): tr.trace_rays(o, d, 512, compact_rows=compact)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(50): out = tr.trace_rays(o, d, 512, compact_rows=compact)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50, out
for name, gen in (("outside-in", scenes.outside_in_rays), ("inside-out", scenes.inside_out_rays)):
    o, d = gen(4096, 1); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    for compact in (True, False):
        res = {0: [], 2: []}
        outs = {}
        for rep in range(5):
            for walk in (0, 2):
                tr.set_option("walk", walk if walk else 1)      # 1 = default rule (BVH below walk_min_rays)
                ms, outs[walk] = timed(o, d, compact)
                res[walk].append(ms)
        n = outs[0]["num_visited_cells"]
        same = torch.equal(n, outs[2]["num_visited_cells"])
        print(f"{name:11s} compact={compact!s:5s}  default {np.median(res[0]):.3f} ms   walk forced {np.median(res[2]):.3f} ms   counts equal: {same}   stats {tr.trace_stats()}")
