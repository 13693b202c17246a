#!/usr/bin/env python3
"""trace_rays on the bench frame (C2) and the C4 frame: ms per call vs the grid of k_write_rows (`fill_blocks`),
plus the segments-only launch (dense_tails = 0) and the pure write ceiling of the box (torch zero_ of 15 GB).
Usage: python profiles/r02_walk_sweep.py [c2|c4|c5 ...]"""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512

def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

buf = torch.empty(15 * 1000**3 // 4, dtype=torch.float32, device=dev)
ms = timed(lambda: buf.zero_()); print(f"write ceiling: zero_ 15 GB {ms:.3f} ms = {15/ms:.2f} TB/s", flush=True)
del buf
which = sys.argv[1:] or ["c2", "c4"]
for cfg, npts, seed in (("c2", 15000, 0), ("c4", 45000, 2), ("c5", 150000, 3)):
    if cfg not in which: continue
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    def run():
        out = tr.trace_rays(o, d, M); k = out["num_visited_cells"]; del out; return k
    inter = int(run().sum()); R = o.shape[0]; gb = R * (28 + 52 * M) / 1e9
    print(f"{cfg}: {len(cells)} tets, {R} rays, {inter} intersections, paths {tr.trace_stats()} reasons {tr.flag_reasons()}", flush=True)
    for opts in ({}, {"seg_unroll": 2}, {"seg_blocks": 512}, {"seg_blocks": 1024}, {"seg_unroll": 2, "seg_blocks": 768},
                 {"seg_unroll": 2, "seg_blocks": 1792}, {"fill_blocks": 1024}):
        for k, v in opts.items(): tr.set_option(k, v)
        ms = timed(run)
        print(f"{cfg} {opts}: {ms:.3f} ms/frame = {gb/ms:.2f} TB/s = {gb/ms/8*100:.1f} % of 8 TB/s", flush=True)
        for k in opts: tr.set_option(k, 4 if k == "seg_unroll" else 0)
    tr.set_option("dense_tails", 0)
    ms = timed(run); print(f"{cfg} segments only (dense_tails=0): {ms:.3f} ms/frame", flush=True)
    tr.set_option("dense_tails", 1)
    del tr
