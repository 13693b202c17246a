#!/usr/bin/env python3
"""Why do latency-bound kernels crawl beside a saturating write stream, and can it be avoided?
trace_rays with dense_tails = 0 (walk + segment writer + literal pairing: all latency-bound) alone, a probe fill of
`gb` GB alone, and both at once, for store flavours (plain / nt / sc1 / sc0 sc1: the last two drop the line from the
XCD's L2) and for CU-masked streams (fill on N CUs, trace on the others).  Perfect overlap: both ~= max; none: sum."""
import ctypes as C, importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
lib = importlib.import_module("tetra-nerf_amd._lib").load()
lib.tn_probe_stream_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
lib.tn_probe_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0"); M = 512; gb = 11.0
pts, cells = scenes.random_mesh(15000, 0)
o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
buf = torch.empty(int(gb * 1e9) // 4, dtype=torch.float32, device=dev)

def mk_stream(first, n):
    if n <= 0: return torch.cuda.Stream()
    h = C.c_void_p(); assert lib.tn_probe_stream_create(first, n, C.byref(h)) == 0, lib.tn_last_error()
    return torch.cuda.ExternalStream(h.value)

def measure(tr, s_trace, s_fill, flavour, blocks, what):
    def once():
        if "t" in what:
            with torch.cuda.stream(s_trace): tr.trace_rays(o, d, M)
        if "f" in what:
            lib.tn_probe_fill(buf.data_ptr(), buf.numel() * 4, flavour, blocks, s_fill.cuda_stream)
    for _ in range(2): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    ms = []
    for _ in range(4):
        torch.cuda.synchronize(); e0.record()
        s_trace.wait_event(e0); s_fill.wait_event(e0)
        once()
        e1a = torch.cuda.Event(); e1b = torch.cuda.Event(); e1a.record(s_trace); e1b.record(s_fill)
        torch.cuda.current_stream().wait_event(e1a); torch.cuda.current_stream().wait_event(e1b)
        e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return min(ms)

for ncu_fill in (0, 64, 96, 128):
    s_fill = mk_stream(0, ncu_fill); s_trace = mk_stream(ncu_fill, 256 - ncu_fill) if ncu_fill else torch.cuda.Stream()
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    tr.set_option("dense_tails", 0)
    t_alone = measure(tr, s_trace, s_fill, 0, 512, "t")
    for flavour, name in ((0, "plain"), (1, "nt"), (2, "sc1"), (3, "sc0 sc1")):
        if ncu_fill and flavour == 3: continue
        for blocks in ((512,) if ncu_fill == 0 else (ncu_fill * 2, ncu_fill * 4)):
            f_alone = measure(tr, s_trace, s_fill, flavour, blocks, "f")
            both = measure(tr, s_trace, s_fill, flavour, blocks, "tf")
            print(f"fill CUs {ncu_fill or 'all'} blocks {blocks} store {name}: trace alone {t_alone:.3f} ms, fill alone {f_alone:.3f} ms "
                  f"({gb/f_alone:.2f} TB/s), both {both:.3f} ms (sum {t_alone+f_alone:.3f}, max {max(t_alone,f_alone):.3f})", flush=True)
    del tr
