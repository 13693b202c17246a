#!/usr/bin/env python3
"""What the field gradient without float atomics costs: a training iteration of both shipped configurations on the C4 batch with
cpp.DETERMINISTIC_FIELD_GRADIENT off / on, interleaved in one process."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
cpp = tn.cpp
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(4096, 1); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
target = torch.rand(len(o), 3, device=dev)
for name, S, Sf, biased in (("tetra-nerf-original", 256, 256, False), ("tetra-nerf", 128, 128, True)):
    torch.manual_seed(0)
    mlp = render.TetraMLP().to(dev)
    field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4); field[1:4] = torch.rand(3, len(pts), device=dev) * 2 - 1
    field.requires_grad_(True)
    opt = torch.optim.SGD([field] + list(mlp.parameters()), lr=1e-3)
    rd = render.TetraRenderer(tr, field, mlp, S, 512, fused=True, num_fine_samples=Sf, biased=biased)
    def step():
        opt.zero_grad(set_to_none=True)
        out = rd.render_train(o, d, gradient_scaling=biased)
        ((out["rgb"] - target) ** 2).mean().backward()
        opt.step()
    def timed(reps=20):
        for _ in range(4): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps): step()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    res = {False: [], True: []}
    for rnd in range(3):
        for flag in (False, True):
            cpp.DETERMINISTIC_FIELD_GRADIENT = flag
            res[flag].append(round(timed(), 3))
    cpp.DETERMINISTIC_FIELD_GRADIENT = False
    a, b = sorted(res[False])[1], sorted(res[True])[1]
    print(f"{name}: atomic field gradient {a:.3f} ms per iteration, without atomics {b:.3f} ms (+{100 * (b / a - 1):.1f} %)   {res}")
