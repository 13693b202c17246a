"""Round 4: the sync-free training forward (TetraRenderer.sync_free_train, default) against the compacting form
(torch.nonzero: one host synchronisation per iteration), interleaved in one process on the C4 4096-ray batch, both shipped
configurations: whole iterations (trace + forward + backward + SGD).  Usage: python profiles/r04j_syncfree_ab.py"""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
tn = importlib.import_module("tetra-nerf_amd")
render = importlib.import_module("tetra-nerf_amd.render")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev)
tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(4096, 1)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
target = torch.rand(len(o), 3, device=dev)
for name, (s_c, s_f, biased, scaling) in (("tetra-nerf-original", (256, 256, False, False)), ("tetra-nerf", (128, 128, True, True))):
    torch.manual_seed(0)
    mlp = render.TetraMLP().to(dev)
    field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4)
    field[1:4] = torch.rand(3, len(pts), device=dev) * 2 - 1
    field.requires_grad_(True)
    params = [field] + list(mlp.parameters())
    opt = torch.optim.SGD(params, lr=1e-3)
    rds = {flag: render.TetraRenderer(tr, field, mlp, s_c, 512, fused=True, num_fine_samples=s_f, biased=biased, sync_free_train=flag)
           for flag in (True, False)}
    times = {True: [], False: []}
    hit = 0
    for rnd in range(5):
        for flag in (False, True):
            rd = rds[flag]

            def step():
                opt.zero_grad(set_to_none=True)
                o_ = rd.render_train(o, d, gradient_scaling=scaling)
                ((o_["rgb"] - target) ** 2).mean().backward()
                opt.step()
                return o_
            hit = int(step()["ray_mask"].sum())
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                step()
            e1.record()
            torch.cuda.synchronize()
            times[flag].append(e0.elapsed_time(e1) / 10)
    flop = hit * (s_c * 82176 + 3 * (s_c + s_f + 1) * 122624)
    for flag in (False, True):
        t = sorted(times[flag])
        m = t[len(t) // 2]
        print(f"{name:20s} sync_free_train={flag!s:5s}: median {m:.3f} ms/iteration (min {t[0]:.3f}, max {t[-1]:.3f})  "
              f"{flop / (m * 1e-3) / 1e12 / 157.3 * 100:.1f} % of the fp32 MFMA peak")
