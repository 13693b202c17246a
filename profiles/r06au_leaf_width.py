"""BVH path, 4096-ray batches: faces per leaf block 16 (default) / 8 / 4 (tighter leaf boxes, more leaves per wave instruction)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
for name, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2), ("C5", 150000, 3)):
    pts, cells = scenes.random_mesh(npts, seed)
    ref = {}
    for lw in (16, 8, 4, 32):
        tr = tn.TetrahedraTracer(dev)
        tr.set_option("leaf_width", lw)
        try:
            tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
        except Exception as e:
            print(name, lw, "load failed:", str(e)[:200]); continue
        line = []
        for kind in ("outside_in", "inside_out"):
            o, d = getattr(scenes, kind + "_rays")(4096, 1)
            o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
            out = tr.trace_rays(o, d, 512)
            chk = int(sum(int(v.view(torch.int32).to(torch.int64).sum()) for v in out.values()))
            same = ref.setdefault(kind, chk) == chk
            ms = []
            for rep in range(3):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    tr.trace_rays(o, d, 512)
                e1.record(); torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1) / 10)
            line.append(f"{kind} {sorted(ms)[1]:.3f} ms (identical: {same})")
        print(f"{name} leaf_width {lw:2d}: " + "  ".join(line), flush=True)
        del tr
