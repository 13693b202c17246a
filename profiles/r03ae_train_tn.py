#!/usr/bin/env python3
"""A few whole training iterations of the `tetra-nerf` configuration (biased sampler, 128 + 128 samples, gradient scaling) on
the C4 batch, for a rocprofv3 --kernel-trace --stats breakdown."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(4096, 1); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
target = torch.rand(len(o), 3, device=dev)
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev)
field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4); field[1:4] = torch.rand(3, len(pts), device=dev) * 2 - 1
field.requires_grad_(True)
opt = torch.optim.SGD([field] + list(mlp.parameters()), lr=1e-3)
rd = render.TetraRenderer(tr, field, mlp, 128, 512, fused=True, num_fine_samples=128, biased=True)
def step():
    opt.zero_grad(set_to_none=True)
    out = rd.render_train(o, d, gradient_scaling=True)
    ((out["rgb"] - target) ** 2).mean().backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 10:.2f} ms per iteration")
