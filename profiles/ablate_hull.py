#!/usr/bin/env python3
"""A/B of the hull-tree leaf size inside one process (walk kernel time via the whole call)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
M = 512
pts, cells = scenes.random_mesh(npts, 0)
o, d = bench.frame_rays(scenes, 0, 800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
trs = {}
for leaf in (8,):
    tr = tn.TetrahedraTracer(dev); tr.set_option("hull_leaf", leaf)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev)); trs[leaf] = tr
res = {}
for rep in range(3):
    for leaf, tr in trs.items():
        for dbg in (0, 16):
            tr.set_option("debug", dbg)
            for _ in range(2): out = tr.trace_rays(o, d, M); del out
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5): out = tr.trace_rays(o, d, M); del out
            e1.record(); torch.cuda.synchronize()
            res.setdefault((leaf, dbg), []).append(e0.elapsed_time(e1) / 5)
for k, v in res.items():
    print(f"tets={len(cells)} hull_leaf={k[0]} debug={k[1]}: " + " ".join(f"{x:.3f}" for x in v) + " ms/frame")
