#!/usr/bin/env python3
"""Speculative tail fill (option spec_fill: slots [K0, M) of every row streamed beside the walk, K0 from the mesh size) vs
the serial walk -> writer -> fill, for several grids of that fill (spec_blocks): ms per trace_rays call, interleaved
repetitions, outputs compared bit for bit."""
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
SETTINGS = ((0, 0, 0), (1, 1024, 0), (1, 2048, 0), (1, 4096, 0), (1, 1024, 320), (1, 1024, 384), (1, 2048, 384))
for cfg, npts, seed, rays in (("c2", 15000, 0, "frame"), ("c4", 45000, 2, "frame"), ("c5", 150000, 3, "frame"), ("c5rays", 150000, 3, 1 << 20)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800) if rays == "frame" else scenes.outside_in_rays(rays, 4)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    def frame():
        out = tr.trace_rays(o, d, M); del out
    gb = o.shape[0] * (28 + 52 * M) / 1e9
    res, ref = {}, None
    for rep in range(2):
        for st in (SETTINGS if rep == 0 else SETTINGS[::-1]):
            tr.set_option("spec_fill", st[0]); tr.set_option("spec_blocks", st[1]); tr.set_option("spec_k0", st[2])
            if rep == 0 and rays == "frame":
                out = tr.trace_rays(o, d, M); torch.cuda.synchronize()
                if ref is None: ref = out
                else:
                    for k in KEYS: assert torch.equal(out[k], ref[k]), (cfg, st, k)
                    del out
            res.setdefault(st, []).append(timed(frame, 8 if rays == "frame" else 3))
    print(f"{cfg}: " + ", ".join(f"{'serial' if not s[0] else 'spec/' + str(s[1]) + '/K0=' + (str(s[2]) if s[2] else 'auto')} {min(res[s]):.3f} ms ({gb/min(res[s])/8*100:.1f} %)" for s in SETTINGS), flush=True)
    del tr, ref
