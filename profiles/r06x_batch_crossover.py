"""Small batches: BVH path (one wavefront per ray) vs the walk schedule, per batch size (the threshold walk_min_rays was
measured in round 2).  ms per trace_rays call, M = 512, outside-in and inside-out batches."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
for name, npts, seed in (("C2 100k tets", 15000, 0), ("C4 300k tets", 45000, 2), ("C5 1M tets", 150000, 3)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    for kind in ("outside_in", "inside_out"):
        for R in (1024, 2048, 4096, 8192, 16384):
            o, d = getattr(scenes, kind + "_rays")(R, 1)
            o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
            res = {}
            for mode, opts in (("bvh", {"walk": 0}), ("walk", {"walk": 2}), ("default", {"walk": 1})):
                for k, v in opts.items():
                    tr.set_option(k, v)
                ms = []
                for rep in range(5):
                    tr.trace_rays(o, d, 512); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        tr.trace_rays(o, d, 512)
                    e1.record(); torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1) / 10)
                res[mode] = sorted(ms)[2]
            print(f"{name} {kind:10s} R={R:6d}: " + "  ".join(f"{m} {v:.3f}" for m, v in res.items()), flush=True)
