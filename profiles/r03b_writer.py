#!/usr/bin/env python3
"""Round-3 segment-writer shapes on the C2 / C4 frames (bit-identical, interleaved repetitions): hits per ray per
iteration (seg_unroll 4 = 2 waves per SIMD, 2 = 4 waves per SIMD) x grid cap (seg_blocks)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
VARIANTS = [("U4 default grid", {"seg_unroll": 4, "seg_blocks": 0}), ("U4 1024", {"seg_unroll": 4, "seg_blocks": 1024}),
            ("U2 default grid (1024)", {"seg_unroll": 2, "seg_blocks": 0}), ("U2 512", {"seg_unroll": 2, "seg_blocks": 512}),
            ("U2 768", {"seg_unroll": 2, "seg_blocks": 768}), ("U2 2048", {"seg_unroll": 2, "seg_blocks": 2048}),
            ("U2 1024 no spec fill", {"seg_unroll": 2, "seg_blocks": 0, "spec_fill": 0}),
            ("U4 no spec fill", {"seg_unroll": 4, "seg_blocks": 0, "spec_fill": 0})]
DEFAULTS = {"seg_unroll": 4, "seg_blocks": 0, "spec_fill": 1}

def timed(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for cfg, npts, seed in (("c2", 15000, 0), ("c4", 45000, 2)):
    if cfg not in (sys.argv[1:] or ["c2", "c4"]): continue
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    ref = tr.trace_rays(o, d, M); torch.cuda.synchronize()
    gb = o.shape[0] * (28 + 52 * M) / 1e9
    def frame():
        out = tr.trace_rays(o, d, M); del out
    res = {n: [] for n, _ in VARIANTS}
    for rep in range(2):
        for name, opts in (VARIANTS if rep == 0 else VARIANTS[::-1]):
            for k, v in {**DEFAULTS, **opts}.items(): tr.set_option(k, v)
            if rep == 0:
                out = tr.trace_rays(o, d, M); torch.cuda.synchronize()
                bad = [k for k in KEYS if not torch.equal(out[k].view(torch.int32), ref[k].view(torch.int32))]
                del out
                if bad: print(f"{cfg} {name}: MISMATCH in {bad}", flush=True)
            res[name].append(timed(frame))
    for name, _ in VARIANTS:
        a, b = res[name]
        print(f"{cfg} {name:<28} {a:.3f} / {b:.3f} ms -> {gb/min(a,b)/8*100:.1f} % of 8 TB/s", flush=True)
    del tr, ref
