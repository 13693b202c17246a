#!/usr/bin/env python3
"""Differential fuzz aimed at the walk's certification hole: the adjacency walk sees only the
connected component of crossed faces that contains the two hull faces; a second component -- faces the rounded
projection accepts although the exact ray misses them -- is invisible to it and excluded only by the vertex-proximity
rule (reason 4).  Such components arise at vertices / edges that lie within rounding distance of the ray, so this
generator aims there:

  meshes   near-duplicate clouds (every 4th vertex has a twin at 1e-8 .. 1e-5), exact and jittered lattices
           (cospherical points: slivers, coplanar faces), thin shells and flat hulls (hull slivers)
  rays     through a mesh vertex, an edge midpoint or a face centroid, the target perturbed by 0 / 1 / 2 / 8 ulps per
           coordinate; origins far outside (large |o| = large shear error), near the mesh, and INSIDE it; plus
           vertex-to-vertex lines

Every batch is traced three ways on the GPU and compared bit for bit (all five outputs, every tail byte):
  walk path (+ count-only BVH cross-check of every certified ray, option verify_stride = 1)  vs  BVH all-hits path;
a slice of every batch is also compared with the CPU oracle.   python profiles/r03_hole_fuzz.py [million rays] [verify_stride]"""
import importlib, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
from oracle import tn_oracle
dev = torch.device("cuda:0")
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
target_rays = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 50_000_000
verify = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # 0: the walk path as shipped (no count cross-check)


sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))   # (round 4: the generators moved to tests/)
import hole_fuzz_lib as lib   # noqa: E402
rng, MESHES, aimed_rays = lib.rng, lib.MESHES, lib.aimed_rays

t0 = time.time(); total = 0; mism_rays = 0; oracle_rays = 0; oracle_bad = 0; reasons = {}; batches = 0
B = 400_000; M = 256
if lib.FAMILY == "big": B, M = 200_000, 1024   # rays of the 1M-point mesh cross 500-650 faces
while total < target_rays:
    for name, make in MESHES:
        pts, cells = make()
        x, c = torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev)
        tw = tn.TetrahedraTracer(dev); tw.set_option("walk", 2); tw.set_option("verify_stride", verify); tw.load_tetrahedra(x, c)
        tb = tn.TetrahedraTracer(dev); tb.set_option("walk", 0); tb.load_tetrahedra(x, c)
        ot = tn_oracle.OracleTracer(use_bvh=True); ot.load_tetrahedra(pts, cells)
        for kind in ("vertex", "edge", "face", "v2v"):
            for k_ulp, origin in ((0, "far"), (1, "near"), (2, "inside"), (8, "near"), (0, "inside"), (1, "far")):
                if kind == "v2v":
                    o, d = scenes.vertex_to_vertex_rays(pts, B, int(rng.integers(1 << 30)), extend=float(rng.choice([0.3, 1.5, 30.0])))
                else:
                    o, d = aimed_rays(pts, cells, B, kind, k_ulp, origin)
                to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
                a = tw.trace_rays(to, td, M); why = tw.flag_reasons(); st = tw.trace_stats()
                b = tb.trace_rays(to, td, M)
                bad = torch.zeros(len(o), dtype=torch.bool, device=dev)
                for k in KEYS:
                    bad |= (a[k].view(torch.int32).reshape(len(o), -1) != b[k].view(torch.int32).reshape(len(o), -1)).any(1)
                nb = int(bad.sum())
                if nb:
                    mism_rays += nb
                    print(f"MISMATCH {name} {kind} ulp={k_ulp} origin={origin}: {nb} rays, first {int(torch.nonzero(bad)[0])}", flush=True)
                for kk, vv in why.items(): reasons[kk] = reasons.get(kk, 0) + vv
                ns = 4096   # oracle slice
                want = ot.trace_rays(o[:ns], d[:ns], M)
                ob = sum(not np.array_equal(a[k][:ns].cpu().numpy().view(np.uint32), np.ascontiguousarray(want[k]).view(np.uint32)) for k in KEYS)
                oracle_rays += ns; oracle_bad += int(ob > 0)
                total += len(o); batches += 1
                del a, b
                if total >= target_rays: break
            if total >= target_rays: break
        print(f"{name}: tets={len(cells)} total rays {total/1e6:.1f} M, {time.time()-t0:.0f} s, reasons so far {dict(sorted(reasons.items()))}", flush=True)
        del tw, tb
        if total >= target_rays: break
print(f"hole fuzz (verify_stride={verify}): {batches} batches, {total} rays (walk path vs BVH path, bitwise), {mism_rays} mismatching rays; "
      f"count-only cross-check: {reasons.get(15, 0)} certified rays verified, {reasons.get(14, 0)} count mismatches; "
      f"oracle slices: {oracle_rays} rays, {oracle_bad} mismatching batches; walk hand-over reasons {dict(sorted(reasons.items()))}; {time.time()-t0:.0f} s")
sys.exit(1 if (mism_rays or oracle_bad) else 0)
