#!/usr/bin/env python3
"""Where does k_write_segments spend its time?  Re-runs the segment writer alone on the hit log of a traced frame
(tn_probe_write_segments) with parts switched off: 1 = no walk-record loads, 2 = only the cell-id store, 4 = no stores,
for several grid sizes.  Usage: python profiles/r02b_writer_ablate.py [c2|c4|c5 ...]"""
import ctypes as C, importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
lib = importlib.import_module("tetra-nerf_amd._lib").load()
lib.tn_probe_write_segments.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0"); M = 512

def timed(fn, n=5):
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
    out = tr.trace_rays(o, d, M); torch.cuda.synchronize()
    hits = int(out["num_visited_cells"].sum())
    h = tr._handle if hasattr(tr, "_handle") else tr._h
    st = torch.cuda.current_stream().cuda_stream
    def run(ab, blocks):
        rc = lib.tn_probe_write_segments(h, M, out["visited_cells"].data_ptr(), out["barycentric_coordinates"].data_ptr(),
                                         out["hit_distances"].data_ptr(), out["vertex_indices"].data_ptr(), ab, blocks, st)
        assert rc == 0, lib.tn_last_error()
    print(f"{cfg}: paths {tr.trace_stats()} reasons {tr.flag_reasons()}", flush=True)
    for variant in (1, 0):
        tr.set_option("seg_variant", variant)
        for blocks in (0, 1024):
            for ab in ((0, 1, 2, 3, 4, 5) if variant == 0 else (0,)):
                ms = timed(lambda: run(ab, blocks))
                print(f"{cfg} ({hits} segments) writer variant {variant} blocks {blocks or 512} ablate {ab}: {ms:.3f} ms", flush=True)
    tr.set_option("seg_variant", 1)
    def frame():
        o2 = tr.trace_rays(o, d, M); del o2
    for variant in (1, 0):
        tr.set_option("seg_variant", variant)
        ms = timed(frame); gb = o.shape[0] * (28 + 52 * M) / 1e9
        print(f"{cfg} whole trace_rays, writer variant {variant}: {ms:.3f} ms = {gb/ms/8*100:.1f} % of 8 TB/s", flush=True)
    del tr, out
