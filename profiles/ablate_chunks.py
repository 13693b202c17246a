#!/usr/bin/env python3
"""Launch-structure sweep of the walk path on the bench workload (mode / chunk size)."""
import importlib, sys
from pathlib import Path
import torch
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
res = {}
for rep in range(3):
    for mode, chunk, dbg in ((1, 0, 0), (0, 0, 0)):
        tr.set_option("mode", mode); tr.set_option("debug", dbg)
        if chunk: tr.set_option("chunk_rays", chunk)
        for _ in range(2): out = tr.trace_rays(o, d, M); del out
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(5): out = tr.trace_rays(o, d, M); del out
        e1.record(); torch.cuda.synchronize()
        res.setdefault((mode, dbg), []).append(e0.elapsed_time(e1) / 5)
for k, v in res.items():
    print(f"tets={len(cells)} M={M} mode={k[0]} debug={k[1]}: " + " ".join(f"{x:.3f}" for x in v) + " ms/frame")
