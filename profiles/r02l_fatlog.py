#!/usr/bin/env python3
"""Fat hit log (option log_records: the walk also logs the tet id, combine code and vertex ids of every hit, the
segment writer reads no walk records) vs the 16-byte log: ms per trace_rays call, interleaved repetitions, outputs
compared bit for bit."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
def timed(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cfg, npts, seed, rays in (("c2", 15000, 0, "frame"), ("c4", 45000, 2, "frame"), ("c5", 150000, 3, "frame"), ("c5rays", 150000, 3, 1 << 20)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800) if rays == "frame" else scenes.outside_in_rays(rays, 4)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    def frame():
        out = tr.trace_rays(o, d, M); del out
    gb = o.shape[0] * (28 + 52 * M) / 1e9
    res, outs = {}, {}
    for rep in range(2):
        for lr in ((1, 0) if rep == 0 else (0, 1)):
            tr.set_option("log_records", lr)
            if rep == 0 and rays == "frame":
                outs[lr] = tr.trace_rays(o, d, M); torch.cuda.synchronize()
            res.setdefault(lr, []).append(timed(frame, 8 if rays == "frame" else 3))
    if outs:
        for k in KEYS: assert torch.equal(outs[0][k], outs[1][k]), (cfg, k)
    print(f"{cfg}: fat log {min(res[1]):.3f} ms ({gb/min(res[1])/8*100:.1f} %), 16-byte log {min(res[0]):.3f} ms ({gb/min(res[0])/8*100:.1f} %)", flush=True)
    del tr, outs
