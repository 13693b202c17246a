"""Fill order vs physical placement (needs the TN_FILL_EXPERIMENT variant library: tn_debug_set_fill_chunk).
For `trials` fresh allocations of the dense rows (same virtual addresses, new physical pages): tn_fill_rows alone with every
row order; then the C2 frame's trace_rays per order, again over fresh allocations."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd")
cpp = importlib.import_module("tetra-nerf_amd.tetranerf_cpp_extension")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
lib = importlib.import_module("tetra-nerf_amd._lib").load()
dev = torch.device("cuda:0")
trials = int(sys.argv[1])
ORDERS = [int(x, 0) for x in sys.argv[2].split(",")]
R, M = 640000, 512


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]


for t in range(trials):
    torch.cuda.empty_cache()
    vc = torch.empty(R, M, dtype=torch.int32, device=dev)
    bc = torch.empty(R, M, 2, 3, dtype=torch.float32, device=dev)
    hd = torch.empty(R, M, 2, dtype=torch.float32, device=dev)
    vi = torch.empty(R, M, 4, dtype=torch.int32, device=dev)
    nbytes = sum(x.numel() * 4 for x in (vc, bc, hd, vi))
    line = []
    for c in ORDERS:
        lib.tn_debug_set_fill_chunk(c & 0xFFFFF); lib.tn_debug_set_fill_blocks(c >> 20)
        ms = timed(lambda: cpp.fill_rows(vc, bc, hd, vi, 0))
        line.append(f"{c:#x}: {(nbytes if not c >> 16 else sum(x.numel() * 4 for x, b in ((vc, 1), (hd, 2), (bc, 4), (vi, 8)) if (c >> 16) & b)) / ms / 1e6:5.0f}")
    zr = bc.numel() * 4 / timed(lambda: bc.zero_()) / 1e6
    print(f"alloc {t}: fill_rows GB/s  " + "  ".join(line) + f"   torch zero_ {zr:5.0f}", flush=True)
    del vc, bc, hd, vi

pts, cells = scenes.random_mesh(15000, 0)
tr = tn.TetrahedraTracer(dev)
tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = bench.frame_rays(scenes, 0, 800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for t in range(trials):
    torch.cuda.empty_cache()
    line = []
    for c in ORDERS:
        lib.tn_debug_set_fill_chunk(c & 0xFFFFF); lib.tn_debug_set_fill_blocks(c >> 20)
        ms = timed(lambda: tr.trace_rays(o, d, M), reps=7)
        line.append(f"{c:#x}: {ms:.3f}")
    print(f"alloc {t}: C2 trace_rays ms  " + "  ".join(line), flush=True)
