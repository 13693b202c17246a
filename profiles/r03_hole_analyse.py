#!/usr/bin/env python3
"""What do the rays that fall into the walk's certification hole have in common?  Re-runs the aimed generator of
profiles/r03_hole_fuzz.py on the meshes where it found mismatches (+ two well-shaped meshes as controls), and for every
ray whose walk result differs from the BVH result -- and for a sample of agreeing rays -- measures, in float64 on the
host: the distance of the ray to the nearest VERTEX and the nearest EDGE of the tets on its chain, and the smallest
HEIGHT of those tets, in units of the rounding distance delta = 6 * 2^-24 * (|o|_inf + scene_max) of the sheared
projection (tn_device.h).  Candidate exclusion rules are then scored: caught mismatches vs flagged agreeing rays."""
import importlib, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))   # (round 4: the generators moved to tests/)
import hole_fuzz_lib as lib   # noqa: E402  (generators shared with the fuzzer)
dev = torch.device("cuda:0"); M = 256; B = 400_000
KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")
rng = lib.rng


def tet_stats(pts, cells, ids, o, d):
    """for the tets `ids` of one ray: (min vertex distance, min edge distance, min height) in float64"""
    v = pts[cells[ids]].astype(np.float64)                 # [k,4,3]
    o = o.astype(np.float64); d = d.astype(np.float64) / np.linalg.norm(d.astype(np.float64))
    rel = v - o
    perp = rel - (rel @ d)[..., None] * d
    vdist = np.linalg.norm(perp, axis=-1).min()
    # edge distance: segment-to-line distance in the plane perpendicular to d
    e = np.array([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)])
    a, b = perp[:, e[:, 0]], perp[:, e[:, 1]]              # [k,6,3] projected endpoints (origin = the ray)
    ab = b - a
    t = np.clip(-(a * ab).sum(-1) / np.maximum((ab * ab).sum(-1), 1e-300), 0.0, 1.0)
    edist = np.linalg.norm(a + t[..., None] * ab, axis=-1).min()
    # heights: 3 * volume / face area, for the 4 faces
    vol = np.abs(np.einsum("ki,ki->k", np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), v[:, 3] - v[:, 0])) / 6.0
    hmin = np.inf
    for f in ((1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)):
        area = 0.5 * np.linalg.norm(np.cross(v[:, f[1]] - v[:, f[0]], v[:, f[2]] - v[:, f[0]]), axis=-1)
        hmin = min(hmin, float((3.0 * vol / np.maximum(area, 1e-300)).min()))
    return vdist, edist, hmin


def main():
    MESHES = [m for m in lib.MESHES if m[0] in ("twins_1e-08", "twins_1e-07", "twins_1e-06", "lattice_1e-7")] + [
        ("random_3000", lambda: scenes.random_mesh(3000, 3)), ("colmap_like", lambda: scenes.colmap_like_mesh(6000))]
    rows = []   # (mesh, mismatch?, vdist/delta, edist/delta, hmin/delta, walk class)
    t0 = time.time()
    for name, make in MESHES:
        pts, cells = make()
        x, c = torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev)
        tw = tn.TetrahedraTracer(dev); tw.set_option("walk", 2); tw.load_tetrahedra(x, c)
        tb = tn.TetrahedraTracer(dev); tb.set_option("walk", 0); tb.load_tetrahedra(x, c)
        smax = float(np.abs(pts).max())
        nbad = 0
        for kind in ("edge", "face", "vertex"):
            for k_ulp, origin in ((0, "far"), (1, "near"), (0, "inside"), (1, "far")):
                o, d = lib.aimed_rays(pts, cells, B, kind, k_ulp, origin)
                to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
                a = tw.trace_rays(to, td, M); b = tb.trace_rays(to, td, M)
                bad = torch.zeros(len(o), dtype=torch.bool, device=dev)
                for k in KEYS:
                    bad |= (a[k].view(torch.int32).reshape(len(o), -1) != b[k].view(torch.int32).reshape(len(o), -1)).any(1)
                bad_idx = torch.nonzero(bad)[:, 0].cpu().numpy()
                good_idx = rng.choice(len(o), 300, replace=False)
                nv_a = a["num_visited_cells"].cpu().numpy(); nv_b = b["num_visited_cells"].cpu().numpy()
                for is_bad, idxs in ((True, bad_idx[:200]), (False, good_idx)):
                    if len(idxs) == 0: continue
                    ti = torch.from_numpy(np.asarray(idxs)).to(dev)
                    ca = a["visited_cells"].index_select(0, ti).cpu().numpy(); cb = b["visited_cells"].index_select(0, ti).cpu().numpy()
                    for j, r in enumerate(idxs):
                        chain = ca[j, :nv_a[r]]; chain = chain[chain >= 0]
                        truth = cb[j, :nv_b[r]]; truth = truth[truth >= 0]
                        ids = np.unique(np.concatenate([chain, truth]))
                        if len(ids) == 0: continue
                        delta = 6.0 * 2.0 ** -24 * (float(np.abs(o[r]).max()) + smax)
                        vd, ed, hm = tet_stats(pts, cells, ids, o[r], d[r])
                        rows.append((name, kind, origin, is_bad, vd / delta, ed / delta, hm / delta, len(truth) - len(chain)))
                nbad += len(bad_idx)
                del a, b
        print(f"{name}: {len(cells)} tets, {nbad} mismatching rays, {time.time() - t0:.0f} s", flush=True)
        del tw, tb

    arr = np.array([(r[3], r[4], r[5], r[6], r[7]) for r in rows], dtype=np.float64)
    names = np.array([r[0] for r in rows])
    bad = arr[:, 0] > 0
    print(f"\n{int(bad.sum())} mismatching rays analysed, {int((~bad).sum())} agreeing rays sampled")
    for label, col in (("nearest vertex", 1), ("nearest edge", 2), ("smallest tet height", 3)):
        qb = np.percentile(arr[bad, col], [0, 50, 90, 99, 100]) if bad.any() else []
        print(f"{label:>20} / delta, mismatching rays: min/50/90/99/max = {np.array2string(np.asarray(qb), precision=3)}")
        for m in sorted(set(names)):
            sel = (~bad) & (names == m)
            if sel.any(): print(f"{'':>20}   agreeing, {m:<14}: 1/10/50 % = {np.array2string(np.percentile(arr[sel, col], [1, 10, 50]), precision=3)}")
    print("extra faces in the BVH result of mismatching rays (truth - walk segments): ", np.unique(arr[bad, 4], return_counts=True))
    for rule, fn in (("vertex < 45 delta (= the kernel's pad)", lambda a: a[:, 1] < 45), ("edge < 4 delta", lambda a: a[:, 2] < 4), ("edge < 16 delta", lambda a: a[:, 2] < 16),
                     ("height < 16 delta", lambda a: a[:, 3] < 16), ("height < 64 delta", lambda a: a[:, 3] < 64), ("height < 256 delta", lambda a: a[:, 3] < 256),
                     ("edge < 4 delta or height < 64 delta", lambda a: (a[:, 2] < 4) | (a[:, 3] < 64))):
        f = fn(arr)
        line = f"rule {rule:<40}: catches {int((f & bad).sum())}/{int(bad.sum())} mismatches; flags of agreeing rays:"
        for m in sorted(set(names)):
            sel = (~bad) & (names == m)
            if sel.any(): line += f" {m} {100.0 * (f & sel).sum() / sel.sum():.1f}%"
        print(line)


if __name__ == "__main__":
    main()
