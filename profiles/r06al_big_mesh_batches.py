"""Small batches on BIG meshes: BVH path (one wavefront per ray; its LDS hit arrays grow with the mesh and the batch then needs
several rounds of waves) vs the walk schedule, per batch size.  Meshes: 400k points (2.7 M tets) and 1M points (6.7 M tets,
C6), M = 1024."""
import importlib, sys, time
from pathlib import Path
import numpy as np, torch
from scipy.spatial import Delaunay
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
for npts in [int(x) for x in sys.argv[1].split(",")]:
    t0 = time.time()
    pts = np.random.default_rng(7).random((npts, 3)).astype(np.float32)
    cells = np.ascontiguousarray(Delaunay(pts.astype(np.float64)).simplices.astype(np.int32))
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    print(f"== {npts} points, {len(cells)} tets (Delaunay {time.time() - t0:.0f} s)", flush=True)
    for kind in ("outside_in", "inside_out"):
        for R in (1024, 4096, 8192, 16384):
            o, d = getattr(scenes, kind + "_rays")(R, 31)
            o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
            res = {}
            for mode, opts in (("bvh", {"walk": 0}), ("walk", {"walk": 2})):
                for k, v in opts.items():
                    tr.set_option(k, v)
                ms = []
                for rep in range(3):
                    tr.trace_rays(o, d, 1024); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        tr.trace_rays(o, d, 1024)
                    e1.record(); torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1) / 5)
                res[mode] = sorted(ms)[1]
            print(f"   {kind:10s} R={R:6d}: " + "  ".join(f"{m} {v:.3f}" for m, v in res.items()), flush=True)
    del tr
    torch.cuda.empty_cache()
