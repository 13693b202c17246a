#!/usr/bin/env python3
"""Randomised bit-exact parity stress of trace_rays (walk + re-walk + BVH paths) against the CPU oracle:
many meshes / ray sets / M, results compared as raw bits.  Run on a GPU box: python profiles/stress_parity.py [n_cases]"""
import importlib, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
from oracle import tn_oracle
dev = torch.device("cuda:0")
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2024)
bad = 0; total_rays = 0; t0 = time.time(); rewalked = 0; bvh = 0
for case in range(n_cases):
    npts = int(rng.choice([300, 1500, 5000, 20000, 60000]))
    seed = int(rng.integers(0, 10_000))
    M = int(rng.choice([64, 256, 512]))
    kind = int(rng.integers(0, 3))
    R = int(rng.choice([20000, 70000]))
    pts, cells = scenes.random_mesh(npts, seed)
    if kind == 0: o, d = scenes.outside_in_rays(R, seed + 1)
    elif kind == 1: o, d = scenes.inside_out_rays(R, seed + 2)
    else:
        w = int(np.sqrt(R)); o, d = scenes.pinhole_rays(w, w, eye=(0.5 + 1.7 * np.cos(seed), 0.5 + 1.7 * np.sin(seed), 0.6), lookat=(0.5, 0.5, 0.5))
    ot = tn_oracle.OracleTracer(use_bvh=True); ot.load_tetrahedra(pts, cells)
    want = ot.trace_rays(o, d, M)
    tr = tn.TetrahedraTracer(dev)
    # round 2b: every path variant is drawn at random (all must be bit-identical): walk or BVH path for every ray, device
    # or host structure build, literal pairing as rows or as an emit mask, 16-byte or fat hit log, LDS-staged or direct
    # segment stores, mesh-sized LDS hit arrays with a forced tiny capacity (overflow relaunch)
    opts = {"walk": int(rng.choice([2, 2, 0])), "gpu_build": int(rng.integers(0, 2)), "literal": int(rng.integers(0, 2)),
            "spec_k0": int(rng.choice([0, 32, 96])), "lds_cap": int(rng.choice([0, 0, 16, 64]))}
    for k, v in opts.items(): tr.set_option(k, v)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    got = tr.trace_rays(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), M)
    st, fr = tr.trace_stats(), tr.flag_reasons()
    rewalked += fr.get(13, 0); bvh += st["general"] - fr.get(13, 0)
    ok = all(np.array_equal(got[k].cpu().numpy().view(np.uint32), np.ascontiguousarray(want[k]).view(np.uint32)) for k in KEYS)
    total_rays += len(o)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: npts={npts} seed={seed} M={M} kind={kind} R={len(o)}", flush=True)
    print(f"case {case}: tets={len(cells)} M={M} kind={kind} rays={len(o)} segs/ray={want['num_visited_cells'].mean():.1f} opts={opts} paths={st} {'ok' if ok else 'FAIL'}", flush=True)
print(f"stress: {n_cases} cases, {total_rays} rays, {rewalked} re-walked chains, {bvh} BVH re-traces, {bad} mismatching cases, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
