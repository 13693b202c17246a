#!/usr/bin/env python3
"""Crossover of the two trace paths on the 300k-tet mesh after round 2b (4096 .. 32768 rays) -- one wavefront per ray
through the wide BVH (walk = 0) vs the adjacency walk + log pipeline forced onto the batch (walk = 2)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
for name, gen in (("outside-in", scenes.outside_in_rays), ("inside-out", scenes.inside_out_rays)):
    for R in (4096, 8192, 12288, 16384, 24576, 32768):
        o, d = gen(R, 1 if name == "outside-in" else 2)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        res = {}
        for walk in (0, 2):
            tr.set_option("walk", walk)
            for dense in (1, 0):
                tr.set_option("dense_tails", dense)
                res[(walk, dense)] = timed(lambda: tr.trace_rays(o, d, M))
        tr.set_option("dense_tails", 1)
        print(f"{name} R={R}: BVH wave-per-ray {res[(0,1)]*1e3:.0f} us (segments only {res[(0,0)]*1e3:.0f}), "
              f"walk+log {res[(2,1)]*1e3:.0f} us (segments only {res[(2,0)]*1e3:.0f}); stats walk=2: {tr.trace_stats()}", flush=True)
