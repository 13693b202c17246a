#!/usr/bin/env python3
"""A/B of the two walk-kernel variants (debug 32 = early prefetch, 64 = compact) for a large
frame and a 4096-ray batch, plus the reasons rays fall back to the general path."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
M = 512
for npts, seed in ((15000, 0), (45000, 2)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    sets = {"frame640k": bench.frame_rays(scenes, 0, 800, 800), "outside_in4096": scenes.outside_in_rays(4096, 1),
            "inside_out4096": scenes.inside_out_rays(4096, 2)}
    for name, (o, d) in sets.items():
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        res = {}
        for rep in range(3):
            for dbg in (32, 64, 32 + 128):
                tr.set_option("debug", dbg)
                for _ in range(2): out = tr.trace_rays(o, d, M); del out
                torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                n = 5 if len(o) > 100000 else 20
                e0.record()
                for _ in range(n): out = tr.trace_rays(o, d, M); del out
                e1.record(); torch.cuda.synchronize()
                res.setdefault(dbg, []).append(e0.elapsed_time(e1) / n * 1e3)
        print(f"tets={len(cells)} {name}: prefetch " + " ".join(f"{x:.0f}" for x in res[32]) + " us | compact " +
              " ".join(f"{x:.0f}" for x in res[64]) + " us | prefetch+uniform-hull " + " ".join(f"{x:.0f}" for x in res[160]) + f" us | stats {tr.trace_stats()} reasons {tr.flag_reasons()}")
