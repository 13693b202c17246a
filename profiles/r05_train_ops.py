#!/usr/bin/env python3
"""Which PyTorch operators of a training iteration launch the rocprim merge-sort kernels the rocprofv3 summary shows
(torch.profiler with stacks, `tetra-nerf` configuration on the C4 batch)."""
import importlib, sys
from pathlib import Path
import torch
from torch.profiler import profile, ProfilerActivity
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
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = prof.events()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA and ("merge" in e.name or "sort" in e.name.lower()):
        print("KERNEL", e.name[:90], "us", e.device_time if hasattr(e, "device_time") else "")
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60, max_src_column_width=90))
