#!/usr/bin/env python3
"""Which path is wrong when the walk and the BVH path disagree, and why rule 8 misses the rest: for aimed rays on the
meshes where profiles/r03_hole_fuzz.py still finds count mismatches, compares (a) the walk path WITHOUT the count
cross-check, (b) the BVH all-hits path, (c) the CPU oracle by BRUTE FORCE (every face tested, no culling) and, for the
rays where (a) != (c), measures the chain's own geometry (only the tets the walk visited): nearest edge / vertex in
units of delta and the thin-neighbourhood exponent rule 8 compares."""
import importlib, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))   # (round 4: the generators moved to tests/)
import hole_fuzz_lib as lib   # noqa: E402
from oracle import tn_oracle      # noqa: E402
from r03_hole_analyse import tet_stats  # noqa: E402  (imports run its module-level code? no: guarded below)
dev = torch.device("cuda:0"); M = 256; B = 400_000
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
rng = lib.rng


def same(a, b, rows):
    bad = np.zeros(len(rows), bool)
    for k in KEYS:
        x = a[k][rows] if isinstance(a[k], np.ndarray) else a[k].index_select(0, torch.from_numpy(rows).to(dev)).cpu().numpy()
        y = b[k] if isinstance(b[k], np.ndarray) else b[k]
        bad |= (np.ascontiguousarray(x).view(np.uint32).reshape(len(rows), -1) != np.ascontiguousarray(y).view(np.uint32).reshape(len(rows), -1)).any(1)
    return ~bad


tot = {"rays": 0, "walk!=bvh": 0, "walk==bf": 0, "bvh==bf": 0, "neither": 0}
geom = []
t0 = time.time()
for name, make in [m for m in lib.MESHES if m[0] in ("twins_1e-07", "twins_1e-06", "lattice_1e-7", "lattice_exact")]:
    pts, cells = make()
    x, c = torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev)
    tw = tn.TetrahedraTracer(dev); tw.set_option("walk", 2); tw.load_tetrahedra(x, c)
    tb = tn.TetrahedraTracer(dev); tb.set_option("walk", 0); tb.load_tetrahedra(x, c)
    bf = tn_oracle.OracleTracer(use_bvh=False); bf.load_tetrahedra(pts, cells)
    smax = float(np.abs(pts).max())
    vars_raw = tw.build_table(2).numpy().view(np.uint32).reshape(-1, 16)            # WalkVar records: dword 10 = code_hi, dword 8 = orig
    thin_of_tet = np.zeros(len(cells), np.uint32)
    thin_of_tet[vars_raw[::4, 8]] = (vars_raw[::4, 10] >> 8) & 0xFF
    for kind in ("edge", "face", "vertex"):
        for k_ulp, origin in ((0, "far"), (1, "near"), (0, "inside")):
            o, d = lib.aimed_rays(pts, cells, B, kind, k_ulp, origin)
            to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
            a = tw.trace_rays(to, td, M); b = tb.trace_rays(to, td, M)
            bad = torch.zeros(len(o), dtype=torch.bool, device=dev)
            for k in KEYS:
                bad |= (a[k].view(torch.int32).reshape(len(o), -1) != b[k].view(torch.int32).reshape(len(o), -1)).any(1)
            rows = torch.nonzero(bad)[:, 0].cpu().numpy()
            tot["rays"] += len(o); tot["walk!=bvh"] += len(rows)
            if len(rows) == 0: continue
            rows = rows[:300]
            truth = bf.trace_rays(o[rows], d[rows], M)
            wa, wb = same(a, truth, rows), same(b, truth, rows)
            tot["walk==bf"] += int(wa.sum()); tot["bvh==bf"] += int(wb.sum()); tot["neither"] += int((~wa & ~wb).sum())
            nva = a["num_visited_cells"].cpu().numpy(); ca = a["visited_cells"].index_select(0, torch.from_numpy(rows).to(dev)).cpu().numpy()
            for j, r in enumerate(rows):
                if wa[j]: continue                       # the walk is right: a BVH-path problem, no geometry needed
                chain = ca[j, :nva[r]]; chain = chain[chain >= 0]
                if len(chain) == 0: continue
                delta = 7.0 * 2.0 ** -24 * (float(np.abs(o[r]).max()) + smax)
                vd, ed, hm = tet_stats(pts, cells, chain, o[r], d[r])
                texp = (np.float32(32.0 * delta).view(np.uint32) >> 23) & 0xFF
                geom.append((name, kind, origin, vd / delta, ed / delta, hm / delta, int(thin_of_tet[chain].min()), int(texp),
                             int(truth["num_visited_cells"][j]) - int(nva[r])))
            del a, b
    print(f"{name}: {tot}, {time.time() - t0:.0f} s", flush=True)
    del tw, tb
print("\nTOTAL", tot)
if geom:
    g = np.array([x[3:] for x in geom], dtype=np.float64)
    print(f"{len(geom)} rays where the walk differs from brute force; of the CHAIN's own tets:")
    print("  nearest vertex / delta  min/50/max", np.percentile(g[:, 0], [0, 50, 100]))
    print("  nearest edge / delta    min/50/90/max", np.percentile(g[:, 1], [0, 50, 90, 100]), " rays with no edge within 8 delta:", int((g[:, 1] > 8).sum()))
    print("  min tet height / delta  min/50/max", np.percentile(g[:, 2], [0, 50, 100]))
    print("  rays whose chain has no thin-flagged tet (min exponent > threshold):", int((g[:, 3] > g[:, 4]).sum()))
    print("  truth - walk segment counts:", np.unique(g[:, 5], return_counts=True))
    for x in geom[:12]: print("   ", x)
