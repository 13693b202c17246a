#!/usr/bin/env python3
"""Scheduling of the literal-pairing / BVH-fallback kernels relative to the main chain (walk -> writer -> fill):
options side_late, aux_general, seg_variant; ms per trace_rays call on the C2 / C4 / C5 frames (two interleaved
repetitions), outputs compared bit for bit with the default schedule."""
import importlib, itertools, sys
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

for cfg, npts, seed in (("c2", 15000, 0), ("c4", 45000, 2), ("c5", 150000, 3)):
    if cfg not in (sys.argv[1:] or ["c2", "c4"]): continue
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    ref = tr.trace_rays(o, d, M); torch.cuda.synchronize()
    gb = o.shape[0] * (28 + 52 * M) / 1e9
    def frame():
        out = tr.trace_rays(o, d, M); del out
    settings = list(itertools.product((0, 1), (0, 1), (0, 1)))   # side_late, aux_general, seg_variant
    res = {s: [] for s in settings}
    for rep in range(2):
        for s in (settings if rep == 0 else settings[::-1]):
            for k, v in zip(("side_late", "aux_general", "seg_variant"), s): tr.set_option(k, v)
            if rep == 0:
                out = tr.trace_rays(o, d, M); torch.cuda.synchronize()
                for k in KEYS: assert torch.equal(out[k], ref[k]), (cfg, s, k)
                del out
            res[s].append(timed(frame))
    for s in settings:
        ms = min(res[s])
        print(f"{cfg} side_late {s[0]} aux_general {s[1]} writer variant {s[2]}: {res[s][0]:.3f} / {res[s][1]:.3f} ms -> {gb/ms/8*100:.1f} % of 8 TB/s", flush=True)
    del tr, ref
