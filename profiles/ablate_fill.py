#!/usr/bin/env python3
"""A/B of the tail-fill store flavour (debug 512 = nontemporal) on the bench frame."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(15000, 0)
o, d = bench.frame_rays(scenes, 0, 800, 800)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
res = {}
for rep in range(4):
    for dbg in (32, 64):
        tr.set_option("debug", dbg)
        for _ in range(2): out = tr.trace_rays(o, d, 512); del out
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(5): out = tr.trace_rays(o, d, 512); del out
        e1.record(); torch.cuda.synchronize()
        res.setdefault(dbg, []).append(e0.elapsed_time(e1) / 5)
for k, v in res.items():
    print(f"FILL debug={k}: " + " ".join(f"{x:.3f}" for x in v) + " ms/frame")
