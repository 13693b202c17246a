#!/usr/bin/env python3
"""Render path, fp32: every pass as ONE launch (tn_render_pass) vs the separate match / gather+MLP / composite kernels,
on the bench frame (800x800, C2 mesh, 65536-ray chunks) and on one 4096-ray batch of the C4 mesh, for the three
configurations of bench.py.  ms per frame / batch, rendered rays/s."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
dev = torch.device("cuda:0"); M = 512
CONFIGS = (("coarse-256", (256, 0, False)), ("tetra-nerf-original", (256, 256, False)), ("tetra-nerf", (128, 128, True)))

def setup(npts, seed):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    torch.manual_seed(0)
    mlp = render.TetraMLP().to(dev)
    field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4)
    field[1:4] = torch.rand(3, len(pts), device=dev) * 2 - 1
    return tr, mlp, field

def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

tr, mlp, field = setup(15000, 0)
o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for name, (s_c, s_f, biased) in CONFIGS:
    res = {}
    for fp in (True, False, True, False):
        rd = render.TetraRenderer(tr, field, mlp, s_c, M, fused=True, num_fine_samples=s_f, biased=biased, fused_pass=fp)
        def frame():
            for s in range(0, o.shape[0], 65536): rd.render(o[s:s + 65536], d[s:s + 65536])
        res.setdefault(fp, []).append(timed(frame, 2))
    a, b = min(res[True]), min(res[False])
    print(f"frame 800x800 {name}: one launch per pass {a*1e3:.1f} ms ({o.shape[0]/a/1e6:.2f} M rays/s), separate kernels {b*1e3:.1f} ms ({o.shape[0]/b/1e6:.2f} M rays/s), ratio {b/a:.3f}", flush=True)
del tr
tr, mlp, field = setup(45000, 2)
o, d = scenes.outside_in_rays(4096, 1); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for name, (s_c, s_f, biased) in CONFIGS:
    res = {}
    for fp in (True, False, True, False):
        rd = render.TetraRenderer(tr, field, mlp, s_c, M, fused=True, num_fine_samples=s_f, biased=biased, fused_pass=fp)
        res.setdefault(fp, []).append(timed(lambda: rd.render(o, d), 10))
    a, b = min(res[True]), min(res[False])
    print(f"C4 batch of 4096 rays {name}: one launch per pass {a*1e3:.2f} ms, separate kernels {b*1e3:.2f} ms, ratio {b/a:.3f}", flush=True)
