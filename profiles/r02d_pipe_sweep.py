#!/usr/bin/env python3
"""Walk -> writer pipelining over ray chunks (option `pipe`) x segment-writer variant: ms per trace_rays call on the
C2 / C4 / C5 frames, each setting measured twice in interleaved order; outputs of every setting compared bit for bit with
pipe = 1.  Usage: python profiles/r02d_pipe_sweep.py [c2|c4|c5 ...]"""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")

def timed(fn, n=10):
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
    settings = [(p, v) for v in (0, 1) for p in (1, 2, 3, 4, 6, 8)]
    res = {s: [] for s in settings}
    for rep in range(2):
        for s in (settings if rep == 0 else settings[::-1]):
            tr.set_option("pipe", s[0]); tr.set_option("seg_variant", s[1])
            if rep == 0:
                out = tr.trace_rays(o, d, M); torch.cuda.synchronize()
                for k in KEYS: assert torch.equal(out[k], ref[k]), (cfg, s, k)
                del out
            res[s].append(timed(frame))
    for s in settings:
        ms = min(res[s])
        print(f"{cfg} pipe {s[0]} writer variant {s[1]}: {res[s][0]:.3f} / {res[s][1]:.3f} ms -> {gb/ms/8*100:.1f} % of 8 TB/s", flush=True)
    del tr, ref
