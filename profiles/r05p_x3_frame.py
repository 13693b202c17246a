#!/usr/bin/env python3
"""The bench's rendered frame (800x800, 15,000-point mesh seed 0, 65,536-ray chunks) in bf16x3 mode and as the fp32 kernel chain, with
the package of the tree named on the command line (A/B of round 4's tree against this one):
    python profiles/r05p_x3_frame.py <root containing tetra-nerf_amd/>"""
import importlib, math, sys, time
import numpy as np, torch
root = sys.argv[1]
sys.path.insert(0, root)
tn = importlib.import_module("tetra-nerf_amd"); render = importlib.import_module("tetra-nerf_amd.render")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(15000, 0)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
c = np.array([0.5, 0.5, 0.5], np.float32); eye = c + 2.0 * np.array([0.0, 1.0, 0.0], np.float32)
o, d = scenes.pinhole_rays(800, 800, eye=tuple(eye), lookat=tuple(c), up=(0.0, 0.0, 1.0), fov_y=45.0)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev)
field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4); field[1:4] = torch.rand(3, len(pts), device=dev) * 2 - 1
R, chunk = o.shape[0], 65536
def timed(rd, reps=3):
    def frame():
        for s in range(0, R, chunk): rd.render(o[s:s + chunk], d[s:s + chunk])
    frame(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): frame()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for name, (s_c, s_f, biased) in (("tetra-nerf-original", (256, 256, False)), ("tetra-nerf", (128, 128, True)), ("coarse-256", (256, 0, False))):
    x3 = timed(render.TetraRenderer(tr, field, mlp, s_c, 512, fused=True, num_fine_samples=s_f, biased=biased, mlp_mode="bf16x3"))
    ch = timed(render.TetraRenderer(tr, field, mlp, s_c, 512, fused=True, num_fine_samples=s_f, biased=biased, fused_pass=False))
    print(f"{root} {name}: bf16x3 {x3:.2f} ms   fp32 kernel chain {ch:.2f} ms")
