#!/usr/bin/env python3
"""Two coarse-pass frames of the render path (800x800, 100k tets) for a per-kernel rocprofv3 --stats breakdown.
usage: render_once.py [fp32|bf16x3] [coarse|tetra-nerf-original|tetra-nerf]"""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = sys.argv[2] if len(sys.argv) > 2 else "coarse"
S, Sf, biased = {"coarse": (256, 0, False), "tetra-nerf-original": (256, 256, False), "tetra-nerf": (128, 128, True)}[cfg]
pts, cells = scenes.random_mesh(15000, 0)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = bench.frame_rays(scenes, 0, 800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev)
field = (torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4
tn.cpp.mlp_set_mode(mode)
rd = render.TetraRenderer(tr, field, mlp, S, 512, fused=True, num_fine_samples=Sf, biased=biased)
import time
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(0, o.shape[0], 65536):
        rd.render(o[s:s + 65536], d[s:s + 65536])
    torch.cuda.synchronize(); print(f"{cfg} {mode}: frame {1e3 * (time.perf_counter() - t0):.1f} ms")
