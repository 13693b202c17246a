#!/usr/bin/env python3
"""4096-ray batch on the 300k-tet stand-in: walk kernel variants (prefetch / no segment stores)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
tr.set_option("walk", 2)
o, d = scenes.outside_in_rays(4096, 1)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for dbg, name in ((32, "prefetch"), (64, "compact"), (32 | 2, "prefetch, no segment stores"), (32 | 2 | 1, "prefetch, no stores, no tails"), (32 | 2 | 1 | 256, "prefetch, no stores/tails, no tuv math"), (64 | 2 | 1 | 256, "compact, no stores/tails, no tuv math")):
    tr.set_option("debug", dbg)
    for _ in range(3): tr.trace_rays(o, d, 512)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20): tr.trace_rays(o, d, 512)
    e1.record(); torch.cuda.synchronize()
    print(f"SB walk {name}: {e0.elapsed_time(e1)/20*1e3:.0f} us  {tr.trace_stats()}")
