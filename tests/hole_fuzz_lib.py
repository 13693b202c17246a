"""Meshes and aimed-ray generators of the certification-hole fuzzer: used by tests/test_parity_configs_gpu.py (seeded
sample) and by profiles/r03_hole_fuzz.py / r03_hole_analyse.py / r03_hole_classify.py (the long runs)."""
import importlib, os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
scenes = importlib.import_module("tetra-nerf_amd.scenes")
# TETRANERF_FUZZ_SEED (default 0 = the runs of rounds 3-4 and the seeded sample of the GPU suite): other values give other
# meshes of the same families and other rays
SEED = int(os.environ.get("TETRANERF_FUZZ_SEED", "0"))
rng = np.random.default_rng(77 + SEED)


def flat_hull_mesh(n=3000, eps=1e-6, seed=9):
    r = np.random.default_rng(seed + 1000 * SEED)
    inner = r.random((n, 3))
    faces = []
    for ax in range(3):
        for side in (0.0, 1.0):
            p = r.random((n // 8, 3)); p[:, ax] = side + (eps * r.normal(size=len(p)) if eps else 0.0)
            faces.append(p)
    return scenes._mesh_of(np.clip(np.concatenate([inner] + faces, 0), -0.001, 1.001))


MESHES = [(f"twins_{s:g}", (lambda s=s: scenes.near_duplicates_mesh(3000, s, seed=int(-np.log10(s)) + 1000 * SEED))) for s in (1e-8, 1e-7, 1e-6, 1e-5)] + [
    ("lattice_exact", lambda: scenes.grid_mesh(10, 0.0)), ("lattice_1e-7", lambda: scenes.grid_mesh(12, 1e-7, seed=5 + 1000 * SEED)),
    ("shells", lambda: scenes.shells_mesh(4000, 1e-4, seed=6 + 1000 * SEED)), ("flat_hull_1e-6", lambda: flat_hull_mesh(3000, 1e-6)),
    ("flat_hull_exact", lambda: flat_hull_mesh(3000, 0.0))]


def ulp_perturb(p, k):
    if k == 0: return p
    p = p.astype(np.float32).copy()
    steps = rng.integers(-k, k + 1, size=p.shape)
    for _ in range(k):
        up, dn = steps > 0, steps < 0
        p[up] = np.nextafter(p[up], np.float32(np.inf)); p[dn] = np.nextafter(p[dn], np.float32(-np.inf))
        steps = steps - np.sign(steps)
    return p


def aimed_rays(pts, cells, n, kind, k_ulp, origin):
    c = cells[rng.integers(0, len(cells), n)]
    v = pts[c]                                            # [n,4,3]
    if kind == "vertex": tgt = v[:, 0]
    elif kind == "edge": tgt = (0.5 * (v[:, 0].astype(np.float64) + v[:, 1])).astype(np.float32)
    else: tgt = (v[:, :3].astype(np.float64).mean(1)).astype(np.float32)
    tgt = ulp_perturb(tgt, k_ulp)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=-1, keepdims=True)
    if origin == "far": o = 0.5 + 40.0 * u
    elif origin == "near": o = 0.5 + 1.2 * u
    else: o = 0.2 + 0.6 * rng.random((n, 3))              # inside the mesh
    o = o.astype(np.float32)
    d = tgt.astype(np.float64) - o
    d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-30)
    return np.ascontiguousarray(o), np.ascontiguousarray(d.astype(np.float32))


