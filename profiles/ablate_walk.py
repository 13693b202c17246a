#!/usr/bin/env python3
"""Ablation of the walk kernel on the bench workload (not a test; outputs are invalid when a
stage is skipped).  debug bit 1 = skip the constant tails, bit 2 = skip the segment stores."""
import importlib, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
pts, cells = scenes.random_mesh(npts, 0)
o, d = bench.frame_rays(scenes, 0, 800, 800)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for walk, dbg, mode in ((1, 0, 0), (1, 0, 1), (1, 0, 2), (1, 1, 0), (1, 2, 0), (1, 3, 0)):
    tr.set_option("walk", walk); tr.set_option("debug", dbg); tr.set_option("mode", mode)
    for _ in range(2): out = tr.trace_rays(o, d, M)
    n = int(out["num_visited_cells"].sum()); del out
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(5): out = tr.trace_rays(o, d, M); del out
    e1.record(); torch.cuda.synchronize()
    print(f"tets={len(cells)} M={M} walk={walk} debug={dbg} mode={mode}: {e0.elapsed_time(e1)/5:.3f} ms/frame, segs={n}, stats={tr.trace_stats()}")
