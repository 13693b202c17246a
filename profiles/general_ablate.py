#!/usr/bin/env python3
"""Where the general (wave-per-ray) kernel spends its time at 4096 rays (300k-tet stand-in)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
tr.set_option("walk", 0)
o, d = scenes.outside_in_rays(4096, 1)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for M in (512, 256):
    for g, name in ((0, "full"), (2, "no sort"), (1, "traversal only (no sort / pairing of 0 hits, tails)"), (3, "traversal only, no sort")):
        tr.set_option("gdebug", g)
        for _ in range(3): tr.trace_rays(o, d, M)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): tr.trace_rays(o, d, M)
        e1.record(); torch.cuda.synchronize()
        print(f"GA M={M} {name}: {e0.elapsed_time(e1)/20*1e3:.0f} us")
tr.set_option("gdebug", 0)
tr.trace_rays(o, d, 512)
import ctypes as C
arr = (C.c_uint64 * 16)(); tr._lib.tn_trace_flag_reasons(tr._h, C.byref(arr))
print(f"GA visits per ray: internal {arr[14]/4096:.1f}, leaves {arr[15]/4096:.1f}")
