#!/usr/bin/env python3
"""Walk/fill overlap sweep on the bench frame: mode 2 (chunked walk on the main stream, tail fill of the
previous chunk on the side stream) with the fill grid capped (`fill_blocks`) so that the fabric is not
saturated while the latency-bound walk runs; mode 1 (walk, then fill) is the baseline.  Also the pure
write ceiling of this box (torch fill_ of a 15 GB buffer)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
M = 512
pts, cells = scenes.random_mesh(npts, 0)
o, d = bench.frame_rays(scenes, 0, 800, 800)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)

def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

buf = torch.empty(15 * 10**9 // 4, dtype=torch.float32, device=dev)
ms = timed(lambda: buf.fill_(0.0)); print(f"write ceiling: fill_ 15 GB {ms:.3f} ms = {15e9 / ms / 1e9:.2f} TB/s")
ms = timed(lambda: buf.zero_()); print(f"write ceiling: zero_ 15 GB {ms:.3f} ms = {15e9 / ms / 1e9:.2f} TB/s")
del buf

def run():
    out = tr.trace_rays(o, d, M); del out

configs = [(1, 0, 0)] + [(2, c, fb) for c in (40000, 80000, 160000, 320000) for fb in (0, 1024, 512, 256, 128)]
res = {}
for rep in range(2):
    for mode, chunk, fb in configs:
        tr.set_option("mode", mode); tr.set_option("fill_blocks", fb)
        if chunk: tr.set_option("chunk_rays", chunk)
        res.setdefault((mode, chunk, fb), []).append(timed(run, 4))
for k, v in res.items():
    print(f"tets={len(cells)} mode={k[0]} chunk={k[1]} fill_blocks={k[2]}: " + " ".join(f"{x:.3f}" for x in v) + " ms/frame")
