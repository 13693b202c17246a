"""Where the walk kernel's time goes (diagnostic build -DTN_WALK_DIAG=1: 100 MHz wall clock around the hull search and the
walk loop of every wave)."""
import ctypes as C, importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
lib = importlib.import_module("tetra-nerf_amd._lib").load()
dev = torch.device("cuda:0")
for name, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2), ("C5", 150000, 3)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    tr.set_option("walk", 2)
    sets = [("frame", bench.frame_rays(scenes, 0, 800, 800) if name != "C5" else scenes.outside_in_rays(1 << 20, 4)),
            ("4096 outside-in", scenes.outside_in_rays(4096, 1)), ("4096 inside-out", scenes.inside_out_rays(4096, 2))]
    for sname, (o, d) in sets:
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        arr = (C.c_ulonglong * 8)()
        for _ in range(2):
            tr.trace_rays(o, d, 512)
        torch.cuda.synchronize(); lib.tn_debug_walk_time(arr, 1)
        tr.trace_rays(o, d, 512)
        torch.cuda.synchronize(); lib.tn_debug_walk_time(arr, 1)
        bd = bench.trace_breakdown(tr, o, d, 512)
        w = max(arr[0], 1)
        print(f"{name} {sname:16s}: walk kernel {bd['walk']:.3f} ms | waves {arr[0]}  hull search mean {arr[1] / w / 100:.1f} us max {arr[3] / 100:.1f} us | "
              f"loop mean {arr[2] / w / 100:.1f} us max {arr[4] / 100:.1f} us | steps of the longest lane: mean {arr[6] / w:.0f} max {arr[5]} | "
              f"per step {arr[2] / max(arr[6], 1) * 10:.0f} ns", flush=True)
