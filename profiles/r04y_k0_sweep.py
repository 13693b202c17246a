#!/usr/bin/env python3
"""First slot K0 of the speculative tail fill on the two frame meshes, finer than profiles/r04f_overlap_sweep.txt (option
spec_k0; interleaved in one process, outputs compared by checksum)."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
for name, npts, seed, k0s in (("C4", 45000, 2, (0, 192, 224, 288)), ("C2", 15000, 0, (0, 320, 352, 416))):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    def run(k0):
        tr.set_option("spec_k0", k0)
        out = tr.trace_rays(o, d, M); chk = 0
        for k in sorted(out): chk = (chk * 1000003 + int(out[k].view(torch.int32).to(torch.int64).sum())) & ((1 << 62) - 1)
        del out
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10): tr.trace_rays(o, d, M)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10, chk
    res = {}
    for rep in range(4):
        for k0 in k0s:
            res.setdefault(k0, []).append(run(k0))
    base = np.median([m for m, _ in res[0]])
    for k0, v in res.items():
        ms = np.median([m for m, _ in v])
        print(f"{name} spec_k0 {k0:3d}{' (default rule)' if k0 == 0 else ''}: median {ms:.3f} ms (min {min(m for m, _ in v):.3f})  {100 * (ms / base - 1):+.1f} %  outputs identical: {len({c for _, c in v} | {res[0][0][1]}) == 1}")
    del tr; torch.cuda.empty_cache()
