"""Does carving the four row arrays from ONE allocation remove the placement spread of the in-step writers?
C2 / C4 frame per fresh allocation, separate allocations vs one slab (TETRANERF_HIP_SLAB read per call)."""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
for name, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    print(f"== {name}")
    for a in range(8):
        line = []
        for mode in ("", "1", "2", "5"):
            os.environ["TETRANERF_HIP_SLAB"] = mode
            torch.cuda.empty_cache()
            ms = []
            for rep in range(3):
                tr.trace_rays(o, d, 512); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    tr.trace_rays(o, d, 512)
                e1.record(); torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1) / 3)
            line.append(f"{'separate' if not mode else 'slab' + mode} {sorted(ms)[1]:.3f}")
        print(f"alloc {a}: " + "  ".join(line), flush=True)
