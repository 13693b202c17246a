"""Child of profiles/r06c_lib_ab.py: trace_rays on the C2 / C4 frames and the C5 rays with whatever library
TETRANERF_HIP_LIB names: median ms of the call as scheduled, the serialised per-kernel breakdown of extra launches (option
"timing"), a checksum of all five outputs (builds must agree bit for bit).  One JSON line."""
import importlib
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
M = 512
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["C2", "C4", "C5"]
opts = [kv.split("=") for kv in sys.argv[2:]]
res = {}
for name, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2), ("C5", 150000, 3)):
    if name not in which:
        continue
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    for k, v in opts:
        tr.set_option(k, int(v))
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    sets = [(name + "_frame", bench.frame_rays(scenes, 0, 800, 800))] if name != "C5" else [("C5_rays", scenes.outside_in_rays(1 << 20, 4))]
    for key, (o, d) in sets:
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        out = tr.trace_rays(o, d, M)
        chk = 0
        for k in sorted(out):
            chk = (chk * 1000003 + int(out[k].view(torch.int32).to(torch.int64).sum())) & ((1 << 62) - 1)
        del out
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                tr.trace_rays(o, d, M)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        ts.sort()
        bds = [bench.trace_breakdown(tr, o, d, M) for _ in range(3)]
        bd = {k: sorted(b[k] for b in bds)[1] for k in bds[0]}
        res[key] = {"ms": ts[len(ts) // 2], "min": ts[0], "checksum": chk, "breakdown": bd}
    del tr
    torch.cuda.empty_cache()
print("RESULT " + json.dumps(res))
