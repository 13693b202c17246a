#!/usr/bin/env python3
"""Debug aid: for an adversarial mesh, list the rays on which the HIP trace differs from the oracle (per path option)."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
tn = importlib.import_module("tetra-nerf_amd"); sc = importlib.import_module("tetra-nerf_amd.scenes")
from oracle import tn_oracle
import test_parity_configs_gpu as T
dev = torch.device("cuda:0"); M = 512
names = sys.argv[1:] or sorted(T.ADVERSARIAL)
for name in names:
    pts, cells = T.ADVERSARIAL[name](sc)
    ot = tn_oracle.OracleTracer(use_bvh=True); ot.load_tetrahedra(pts, cells)
    for sname, (o, d) in T._ray_sets(sc, pts, 20000, 40, pts.min(0), pts.max(0)).items():
        want = ot.trace_rays(o, d, M)
        for walk, extra in ((2, {}), (2, {"prefill": 1}), (0, {})):
            tr = T._tracer(tn, dev, pts, cells, walk=walk, **extra)
            out = T._trace(tr, dev, o, d, M)
            got = {k: v.cpu().numpy() for k, v in out.items()}
            badrows = np.zeros(len(o), bool)
            for k in T.KEYS:
                badrows |= (got[k].view(np.uint32) != np.ascontiguousarray(want[k]).view(np.uint32)).reshape(len(o), -1).any(1)
            idx = np.nonzero(badrows)[0]
            print(f"{name}/{sname} walk={walk} {extra}: {len(idx)} bad rays of {len(o)}; stats {tr.trace_stats()} reasons {tr.flag_reasons()}", flush=True)
            for r in idx[:4]:
                gn, wn = int(got["num_visited_cells"][r]), int(want["num_visited_cells"][r])
                print(f"   ray {r}: o={o[r].tolist()} d={d[r].tolist()} got n={gn} want n={wn}")
                n = max(gn, wn, 1)
                print("     got  cells", got["visited_cells"][r, :min(n, 12)].tolist(), "t", np.round(got["hit_distances"][r, :min(n, 6)].ravel(), 7).tolist())
                print("     want cells", want["visited_cells"][r, :min(n, 12)].tolist(), "t", np.round(want["hit_distances"][r, :min(n, 6)].ravel(), 7).tolist())
                raw = ot.trace_rays(o[r:r+1], d[r:r+1], M, with_raw=True)["raw"]
                c = int(raw["count"][0]); print("     oracle raw hits", c, "ids", raw["ids"][0, :min(c, 12)].tolist(), "t", raw["t"][0, :min(c, 12)].tolist())
