"""Serialised breakdown of the walk schedule on small batches (option "timing")."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev)
tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
tr.set_option("walk", 2)
for R in (1024, 4096, 16384, 65536):
    o, d = scenes.outside_in_rays(R, 1)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    for _ in range(3):
        bd = bench.trace_breakdown(tr, o, d, 512)
    print(f"R={R}: " + " ".join(f"{k[:9]} {v:.3f}" for k, v in bd.items()), "paths", tr.trace_stats(), flush=True)
