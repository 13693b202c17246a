#!/usr/bin/env python3
"""trace_rays on the larger BASELINE.json configurations: C4 (300k-tet stand-in: 800x800 frame and
4096-ray batches) and C5 (1M tets, 2^20 outside-in rays), M = 512, dense reference outputs.
Prints ms, rays/s, intersections/s and the HBM roofline fraction (R*(28+52M) bytes / time / 8 TB/s)."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
M = 512

def timed(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def report(name, tr, o, d, n):
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    def run():
        out = tr.trace_rays(o, d, M); k = out["num_visited_cells"]; del out; return k
    inter = int(run().sum()); st = tr.trace_stats()
    ms = timed(run, n)
    R = o.shape[0]; gb = R * (28 + 52 * M) / 1e9
    print(f"{name}: R={R} M={M}: {ms:.3f} ms, {R/ms*1e3:.3e} rays/s, {inter/ms*1e3:.3e} intersections/s "
          f"({inter/R:.1f}/ray), {gb/ms:.2f} TB/s = {gb/ms/8*100:.1f}% of 8 TB/s; paths {st}; "
          f"walk hand-over reasons {tr.flag_reasons()}", flush=True)

for cfg, npts, seed in (("C4", 45000, 2), ("C5", 150000, 3)):
    t0 = time.time(); pts, cells = scenes.random_mesh(npts, seed); t_mesh = time.time() - t0
    tr = tn.TetrahedraTracer(dev)
    t0 = time.time(); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev)); torch.cuda.synchronize()
    print(f"{cfg}: {len(pts)} vertices, {len(cells)} tets (Delaunay {t_mesh:.1f} s on the host), load_tetrahedra {time.time()-t0:.2f} s", flush=True)
    if cfg == "C4":
        report("C4 800x800 pinhole frame", tr, *bench.frame_rays(scenes, 0, 800, 800), 5)
        report("C4 4096 outside-in rays", tr, *scenes.outside_in_rays(4096, 1), 20)
        report("C4 4096 inside-out rays", tr, *scenes.inside_out_rays(4096, 2), 20)
        report("C4 65536 outside-in rays", tr, *scenes.outside_in_rays(65536, 1), 10)
    else:
        report("C5 2^20 outside-in rays", tr, *scenes.outside_in_rays(1 << 20, 4), 3)
        report("C5 800x800 pinhole frame", tr, *bench.frame_rays(scenes, 0, 800, 800), 5)
    del tr
