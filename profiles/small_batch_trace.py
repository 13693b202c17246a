#!/usr/bin/env python3
"""C4-style 4096-ray batches on the 300k-tet stand-in, for `rocprofv3 --kernel-trace --stats`."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(4096, 1)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
walk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tr.set_option("walk", walk)
for _ in range(20):
    out = tr.trace_rays(o, d, 512)
torch.cuda.synchronize()
print("stats", tr.trace_stats())
