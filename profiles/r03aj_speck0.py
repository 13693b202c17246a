#!/usr/bin/env python3
"""Speculative fill beside the walk: off / the shipped last quarter / larger shares of the row (spec_k0), C2 and C4 frames,
interleaved repetitions on one box, outputs compared bit for bit."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
VARIANTS = [("spec fill off", {"spec_fill": 0, "spec_k0": 0}), ("last quarter (384)", {"spec_fill": 1, "spec_k0": 0}),
            ("from 320", {"spec_fill": 1, "spec_k0": 320}), ("from 256", {"spec_fill": 1, "spec_k0": 256}), ("from 448", {"spec_fill": 1, "spec_k0": 448})]
def timed(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cfg, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tr.set_option("spec_fill", 0)
    ref = tr.trace_rays(o, d, M); torch.cuda.synchronize()
    def frame():
        out = tr.trace_rays(o, d, M); del out
    res = {n: [] for n, _ in VARIANTS}
    for rep in range(3):
        for name, opts in (VARIANTS if rep % 2 == 0 else VARIANTS[::-1]):
            for k, v in opts.items(): tr.set_option(k, v)
            if rep == 0:
                out = tr.trace_rays(o, d, M); torch.cuda.synchronize()
                bad = [k for k in KEYS if not torch.equal(out[k].view(torch.int32), ref[k].view(torch.int32))]
                del out
                if bad: print(f"{cfg} {name}: MISMATCH in {bad}", flush=True)
            res[name].append(timed(frame))
    for name, _ in VARIANTS:
        print(f"{cfg} {name:22s} {min(res[name]):.3f} ms  (runs {[round(x, 3) for x in res[name]]})", flush=True)
    del ref, tr
