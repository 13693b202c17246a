"""How much of the C5 time (2^20 incoherent rays, 1M tets) is the incoherence?  The same rays in three orders: as given,
sorted by the Morton code of the origin, sorted by (origin cell, target cell) -- i.e. what an in-library ray sort could buy."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(150000, 3)
tr = tn.TetrahedraTracer(dev)
tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(1 << 20, 4)


def morton(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.uint64), 0, (1 << bits) - 1)
    code = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for k in range(3):
            code |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + k)
    return code


# where the ray leaves the unit cube region (a proxy for its target): origin + direction * t_mid
mid = o + d * (np.linalg.norm(o - 0.5, axis=1, keepdims=True) / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9))
orders = {"as given": np.arange(len(o)), "by origin": np.argsort(morton(o), kind="stable"),
          "by origin cell (4 bits), then target": np.lexsort((morton(mid), morton(o, 4)))}
for name, perm in orders.items():
    oo = torch.from_numpy(np.ascontiguousarray(o[perm])).to(dev)
    dd = torch.from_numpy(np.ascontiguousarray(d[perm])).to(dev)
    for _ in range(2):
        tr.trace_rays(oo, dd, 512)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        tr.trace_rays(oo, dd, 512)
    e1.record(); torch.cuda.synchronize()
    bd = bench.trace_breakdown(tr, oo, dd, 512)
    print(f"{name:40s}: {e0.elapsed_time(e1) / 5:7.3f} ms per call | " + " ".join(f"{k[:9]} {v:.3f}" for k, v in bd.items()), flush=True)
