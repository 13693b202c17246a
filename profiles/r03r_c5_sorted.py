#!/usr/bin/env python3
"""C5 (1.01M tets, 2^20 outside-in rays): how much of the time is the INCOHERENCE of the ray order?  The same rays traced in
the given (random) order, sorted by a Morton key of the point where they enter the scene box, and by a 5-D key (entry
point + direction).  Results are identical up to the permutation; the sort itself is not timed (it would be ~0.2 ms)."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")

def part(x, bits):
    x = x.astype(np.uint64); out = np.zeros_like(x)
    return x
def morton(cols, bits):
    key = np.zeros(len(cols[0]), dtype=np.uint64)
    for b in range(bits - 1, -1, -1):
        for c in cols:
            key = (key << np.uint64(1)) | ((c >> np.uint64(b)) & np.uint64(1))
    return key
def quant(x, lo, hi, bits):
    return np.clip(((x - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1).astype(np.uint64)

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

npts, seed = (150000, 3) if len(sys.argv) < 2 else (int(sys.argv[1]), 3)
pts, cells = scenes.random_mesh(npts, seed)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(1 << 20, 4)
# entry point into the unit cube (slab test)
inv = 1.0 / d
t0 = (0.0 - o) * inv; t1 = (1.0 - o) * inv
tin = np.minimum(t0, t1).max(axis=1); p = o + tin[:, None] * d
orders = {"as given": np.arange(len(o))}
orders["entry point, 3x7-bit Morton"] = np.argsort(morton([quant(p[:, i], 0, 1, 7) for i in range(3)], 7), kind="stable")
orders["entry point 3x5 + direction 3x5 bits"] = np.argsort(morton([quant(p[:, i], 0, 1, 5) for i in range(3)] + [quant(d[:, i], -1, 1, 5) for i in range(3)], 5), kind="stable")
orders["direction 3x5 major, entry 3x5 minor"] = np.argsort((morton([quant(d[:, i], -1, 1, 5) for i in range(3)], 5) << np.uint64(15)) | morton([quant(p[:, i], 0, 1, 5) for i in range(3)], 5), kind="stable")
ref = None
for name, perm in orders.items():
    oo, dd = torch.from_numpy(o[perm]).to(dev), torch.from_numpy(d[perm]).to(dev)
    ms = timeit(lambda: tr.trace_rays(oo, dd, 512))
    out = tr.trace_rays(oo, dd, 512)
    nv = out["num_visited_cells"].cpu().numpy()
    inv_perm = np.empty_like(perm); inv_perm[perm] = np.arange(len(perm))
    if ref is None: ref = nv
    same = bool((nv[inv_perm] == ref).all())
    st = tr.trace_stats() if hasattr(tr, "trace_stats") else None
    print(f"{name:42s} {ms:7.2f} ms   counts identical to the given order: {same}   {st}", flush=True)
    del out
