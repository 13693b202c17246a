"""Where a ray's time goes on the BVH path (one wavefront per ray, 4096-ray batches): diagnostic build, 100 MHz clock around
traversal + leaf tests, the sort, pairing + row write."""
import ctypes as C, importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
lib = importlib.import_module("tetra-nerf_amd._lib").load()
dev = torch.device("cuda:0")
for name, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2), ("C5", 150000, 3)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    for kind in ("outside_in", "inside_out"):
        o, d = getattr(scenes, kind + "_rays")(4096, 1)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        arr = (C.c_ulonglong * 8)()
        for _ in range(2):
            tr.trace_rays(o, d, 512)
        torch.cuda.synchronize(); lib.tn_debug_general_time(arr, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); tr.trace_rays(o, d, 512); e1.record()
        torch.cuda.synchronize(); lib.tn_debug_general_time(arr, 1)
        n = max(arr[0], 1)
        st = tr.trace_stats()
        print(f"{name} {kind:10s}: call {e0.elapsed_time(e1):.3f} ms | rays {arr[0]} | per ray: traversal + leaves {arr[1] / n / 100:.1f} us, sort {arr[2] / n / 100:.1f} us, "
              f"pairing + rows {arr[3] / n / 100:.1f} us, hits {arr[6] / n:.0f} | {st}", flush=True)
