#!/usr/bin/env python3
"""Segment writer: groups handed out by an atomic counter (draw issued one group ahead) vs the static round-robin deal,
same box, interleaved repetitions, bit-identical outputs; C2 / C4 frames."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
VARIANTS = [("static deal", {"seg_dynamic": 0}), ("dynamic", {"seg_dynamic": 1}), ("dynamic, 1024 blocks", {"seg_dynamic": 1, "seg_blocks": 1024}),
            ("static, no spec fill", {"seg_dynamic": 0, "spec_fill": 0}), ("dynamic, no spec fill", {"seg_dynamic": 1, "spec_fill": 0})]
DEFAULTS = {"seg_dynamic": 1, "seg_blocks": 0, "spec_fill": 1}

def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for cfg, npts, seed in (("c2", 15000, 0), ("c4", 45000, 2)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    ref = tr.trace_rays(o, d, M); torch.cuda.synchronize()
    gb = o.shape[0] * (28 + 52 * M) / 1e9
    def frame():
        out = tr.trace_rays(o, d, M); del out
    res = {n: [] for n, _ in VARIANTS}
    for rep in range(3):
        for name, opts in (VARIANTS if rep % 2 == 0 else VARIANTS[::-1]):
            for k, v in {**DEFAULTS, **opts}.items(): tr.set_option(k, v)
            if rep == 0:
                out = tr.trace_rays(o, d, M); torch.cuda.synchronize()
                bad = [k for k in KEYS if not torch.equal(out[k].view(torch.int32), ref[k].view(torch.int32))]
                del out
                if bad: print(f"{cfg} {name}: MISMATCH in {bad}", flush=True)
            res[name].append(timed(frame))
    for name, _ in VARIANTS:
        r = res[name]
        print(f"{cfg} {name:<28} " + " / ".join(f"{x:.3f}" for x in r) + f" ms -> {gb/min(r)/8*100:.1f} % of 8 TB/s", flush=True)
    del tr, ref
