#!/usr/bin/env python3
"""One warm-up + two trace_rays calls on a chosen config, for a per-dispatch rocprofv3 timeline.
usage: c5_once.py <mesh points> <seed> <rays: N outside-in | frame> [option=value ...]"""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
npts, seed, rays = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
pts, cells = scenes.random_mesh(npts, seed)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = bench.frame_rays(scenes, 0, 800, 800) if rays == "frame" else scenes.outside_in_rays(int(rays), 4)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for kv in sys.argv[4:]:        # option=value pairs
    k, v = kv.split("="); tr.set_option(k, int(v)) if k != "x" else None
for _ in range(3):
    out = tr.trace_rays(o, d, 512); del out
    torch.cuda.synchronize()
print(tr.trace_stats(), tr.flag_reasons())
