#!/usr/bin/env python3
"""Tail fill (needs the walk counts only) on a side stream BESIDE the segment writer, or after it (the shipped schedule):
same box, interleaved repetitions, bit-identical outputs; C2 / C4 frames and the C5 stress set.
Needs the `fill_early` option of the experiment (not in the tree: 5-9 % slower on the frames, r03ag_fill_ab.txt)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")

def timed(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for cfg, npts, seed, rays in (("C2 frame", 15000, 0, None), ("C4 frame", 45000, 2, None), ("C5 2^20 rays", 150000, 3, 1 << 20)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800) if rays is None else scenes.outside_in_rays(rays, 4)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tr.set_option("fill_early", 0)
    ref = tr.trace_rays(o, d, M); torch.cuda.synchronize()
    def frame():
        out = tr.trace_rays(o, d, M); del out
    res = {0: [], 1: []}
    for rep in range(4):
        for v in ((0, 1) if rep % 2 == 0 else (1, 0)):
            tr.set_option("fill_early", v)
            if rep == 0 and v == 1:
                out = tr.trace_rays(o, d, M); torch.cuda.synchronize()
                bad = [k for k in KEYS if not torch.equal(out[k].view(torch.int32), ref[k].view(torch.int32))]
                del out
                print(f"{cfg}: outputs {'MISMATCH in ' + str(bad) if bad else 'bit-identical'}", flush=True)
            res[v].append(timed(frame))
    a, b = min(res[0]), min(res[1])
    print(f"{cfg}: writer then fill (serial) {a:.3f} ms (runs {[round(x, 3) for x in res[0]]}), fill beside the writer {b:.3f} ms ({[round(x, 3) for x in res[1]]}): {100 * (a / b - 1):+.1f} %", flush=True)
    del ref, tr
