"""CPU tests of the oracle (oracle/tn_oracle.c) against everything the reference's own tests
pin for this path, plus the committed golden fixtures.  No GPU needed."""
import hashlib
import random
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"


def _checksum(res):
    h = hashlib.sha256()
    for k in ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances",
              "barycentric_coordinates"):
        h.update(np.ascontiguousarray(res[k]).tobytes())
    return h.digest()


def test_face_table_cube(oracle, scenes):
    pts, cells = scenes.cube_mesh()
    faces, ft = oracle.build_faces(cells)
    # 12 tets around a centre vertex: 12 hull faces + 18 interior faces
    assert len(faces) == 30
    assert int((ft[:, 1] == 0xFFFFFFFF).sum()) == 12
    # first-seen order and unsorted first-seen triple (tetrahedra_tracer.cpp:45-71):
    # tet 0 = (0,1,2,8); face j = vertices (j+1)%4,(j+2)%4,(j+3)%4
    assert faces[0].tolist() == [1, 2, 8]
    assert faces[1].tolist() == [2, 8, 0]
    assert faces[2].tolist() == [8, 0, 1]
    assert faces[3].tolist() == [0, 1, 2]
    assert ft[0].tolist()[0] == 0


def test_face_table_rejects_triple_face(oracle):
    cells = np.array([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 5]], np.int32)
    with pytest.raises(RuntimeError, match="shared by more than two"):
        oracle.build_faces(cells)


def test_power_of_two(oracle, scenes):
    pts, cells = scenes.cube_mesh()
    tr = oracle.OracleTracer()
    tr.load_tetrahedra(pts, cells)
    with pytest.raises(RuntimeError, match="power of 2"):
        tr.trace_rays(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32), 24)


def test_cube_ray_known_answer(oracle, scenes):
    """The cube of test_trace_rays_simple (tests/test_tetrahedra_tracer.py:228-267; the
    reference asserts nothing there: "TODO: check results").  Its axis-parallel ray passes
    exactly through mesh edges (a tie case, kept as a GPU-vs-oracle parity vector); the
    geometric known answers are checked on a generic ray: it enters the unit cube at x=0 and
    leaves at x=1."""
    pts, cells = scenes.cube_mesh()
    tr = oracle.OracleTracer()
    tr.load_tetrahedra(pts, cells)
    o = np.array([[-0.05, 0.07, 0.13]], np.float32)
    d = np.array([[1.0, 0.11, 0.23]], np.float32)
    d /= np.linalg.norm(d)
    res = tr.trace_rays(o, d, 16)
    n = int(res["num_visited_cells"][0])
    assert n >= 2
    hd = res["hit_distances"][0, :n]
    np.testing.assert_allclose(hd[0, 0], 0.05 / d[0, 0], rtol=1e-5)
    np.testing.assert_allclose(hd[-1, 1], 1.05 / d[0, 0], rtol=1e-5)
    # contiguous, ascending segments
    np.testing.assert_array_equal(hd[1:, 0], hd[:-1, 1])
    assert np.all(hd[:, 1] > hd[:, 0])
    # every segment's entry/exit points reconstructed from barycentrics lie on the ray
    vi = res["vertex_indices"][0, :n]
    bc = res["barycentric_coordinates"][0, :n]
    for k in range(n):
        for side in (0, 1):
            w = bc[k, side]
            w4 = np.array([1 - w.sum(), *w])
            p = (pts[vi[k]] * w4[:, None]).sum(0)
            np.testing.assert_allclose(p, o[0] + hd[k, side] * d[0], atol=2e-6)
    # tail: ids -1, floats 0
    assert np.all(res["visited_cells"][0, n:] == -1)
    assert np.all(res["vertex_indices"][0, n:] == -1)
    assert np.all(res["hit_distances"][0, n:] == 0) and np.all(res["barycentric_coordinates"][0, n:] == 0)
    # the visited cells are tets that contain the segment midpoints
    for k in range(n):
        c = res["visited_cells"][0, k]
        mid = o[0] + 0.5 * (hd[k, 0] + hd[k, 1]) * d[0]
        T = pts[cells[c]]
        lam = np.linalg.solve((T[1:] - T[0]).T.astype(np.float64), mid - T[0])
        assert lam.min() > -1e-5 and lam.sum() < 1 + 1e-5
        assert sorted(vi[k].tolist()) == sorted(cells[c].tolist())
    # the reference's own (degenerate, edge-grazing) ray still yields ordered segments
    res = tr.trace_rays(np.array([[-0.05, 0.05, 0.05]], np.float32), np.array([[1, 0, 0]], np.float32), 16)
    n = int(res["num_visited_cells"][0])
    hd = res["hit_distances"][0, :n]
    assert n >= 1 and np.all(hd[:, 1] > hd[:, 0]) and abs(hd[0, 0] - 0.05) < 1e-6


def test_bottle_golden(oracle, scenes, bottle):
    """Config C1: the reference's test asset, 64x64 rays, M=256 -- regression-pinned."""
    g = np.load(GOLDEN / "bottle_oracle.npz")
    tr = oracle.OracleTracer()
    tr.load_tetrahedra(bottle["vertices"], bottle["cells"])
    assert len(tr.faces) == int(g["num_faces"]) == 28299
    o, d = scenes.pinhole_rays(64, 64)
    res = tr.trace_rays(o, d, 256)
    np.testing.assert_array_equal(res["num_visited_cells"], g["num_visited_cells"])
    np.testing.assert_array_equal(res["visited_cells"][:, :96], g["visited_cells"])
    assert _checksum(res) == g["sha256"].tobytes()
    assert int((res["num_visited_cells"] > 0).sum()) == 146


def test_bottle_on_ray_property(oracle, scenes, bottle):
    """The reference's only check of the tracer (tests/test_tetrahedra_tracer.py:153-207):
    points reconstructed from find_visited_cells barycentrics lie on the ray."""
    tr = oracle.OracleTracer()
    tr.load_tetrahedra(bottle["vertices"], bottle["cells"])
    o, d = scenes.pinhole_rays(64, 64)
    out = tr.trace_rays(o, d, 256)
    samples = np.ascontiguousarray(np.broadcast_to(np.linspace(0.90, 1.1, 300, dtype=np.float32), (len(o), 300)))
    inter = tr.find_visited_cells(out["num_visited_cells"], out["visited_cells"], out["barycentric_coordinates"],
                                  out["hit_distances"], out["vertex_indices"], samples)
    nonempty = np.nonzero(out["num_visited_cells"] > 0)[0].tolist()
    random.seed(42)
    checked = 0
    for r in random.sample(nonempty, min(100, len(nonempty))):
        m = inter["mask"][r]
        if not m.any():
            continue
        vi = inter["vertex_indices"][r][m]
        b = inter["barycentric_coordinates"][r][m]
        w = np.concatenate([1 - b.sum(-1, keepdims=True), b], -1)
        p = (bottle["vertices"][vi] * w[..., None]).sum(-2)
        dirs = p - o[r]
        dots = (dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)) @ d[r]
        assert np.all(np.abs(dots - 1.0) <= 0.05)
        # much tighter than the reference's 0.05: the points are the samples themselves
        np.testing.assert_allclose(p, o[r] + samples[r][m][:, None] * d[r], atol=5e-5)
        checked += 1
    assert checked > 50


def test_bvh_equals_bruteforce(oracle, scenes):
    pts, cells = scenes.random_mesh(3000, 11)
    a = oracle.OracleTracer()
    a.load_tetrahedra(pts, cells)
    b = oracle.OracleTracer(use_bvh=True)
    b.load_tetrahedra(pts, cells)
    for (o, d) in (scenes.outside_in_rays(1500, 1), scenes.inside_out_rays(1500, 2)):
        ra = a.trace_rays(o, d, 256, with_raw=True)
        rb = b.trace_rays(o, d, 256, with_raw=True)
        for k in ("count", "ids", "t", "uv"):
            np.testing.assert_array_equal(ra["raw"][k], rb["raw"][k])
        for k in ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates"):
            np.testing.assert_array_equal(ra[k], rb[k])


def test_segments_cover_ray_inside_hull(oracle, scenes):
    """Structural property on a random Delaunay mesh: segments ascend, never overlap, gaps are
    only dropped slivers (< ~eps), and each cell contains its segment midpoint."""
    pts, cells = scenes.random_mesh(2000, 3)
    tr = oracle.OracleTracer(use_bvh=True)
    tr.load_tetrahedra(pts, cells)
    o, d = scenes.outside_in_rays(512, 4)
    res = tr.trace_rays(o, d, 512)
    n = res["num_visited_cells"]
    assert n.max() < 511
    bad_mid = 0
    for r in range(len(o)):
        k = int(n[r])
        if k == 0:
            continue
        hd = res["hit_distances"][r, :k]
        assert np.all(hd[:, 1] - hd[:, 0] >= 1e-6 * 0.999)
        assert np.all(hd[1:, 0] >= hd[:-1, 1])
        assert np.all(hd[1:, 0] - hd[:-1, 1] < 1e-4)
        # check a few midpoints
        for s in range(0, k, 7):
            c = res["visited_cells"][r, s]
            mid = o[r].astype(np.float64) + 0.5 * (float(hd[s, 0]) + float(hd[s, 1])) * d[r]
            T = pts[cells[c]].astype(np.float64)
            lam = np.linalg.solve((T[1:] - T[0]).T, mid - T[0])
            if not (lam.min() > -1e-3 and lam.sum() < 1 + 1e-3):
                bad_mid += 1
    assert bad_mid == 0


def test_overflow_keeps_nearest(oracle, scenes):
    pts, cells = scenes.random_mesh(3000, 5)
    tr = oracle.OracleTracer(use_bvh=True)
    tr.load_tetrahedra(pts, cells)
    o, d = scenes.outside_in_rays(64, 6)
    big = tr.trace_rays(o, d, 512, with_raw=True)
    small = tr.trace_rays(o, d, 32, with_raw=True)
    assert big["raw"]["count"].max() > 40
    for r in range(len(o)):
        c = min(int(big["raw"]["count"][r]), 31)
        assert int(small["raw"]["count"][r]) == c
        np.testing.assert_array_equal(small["raw"]["ids"][r, :c], big["raw"]["ids"][r, :c])


def test_find_visited_cells_defaults_and_lerp(oracle):
    nv = np.array([2, 0], np.int32)
    vc = np.array([[7, 9, -1, -1], [-1, -1, -1, -1]], np.int32)
    hd = np.zeros((2, 4, 2), np.float32)
    hd[0, 0] = (1.0, 2.0)
    hd[0, 1] = (2.5, 3.0)
    bc = np.zeros((2, 4, 2, 3), np.float32)
    bc[0, 0, 0] = (1, 0, 0)
    bc[0, 0, 1] = (0, 1, 0)
    bc[0, 1, 0] = (0, 0, 1)
    bc[0, 1, 1] = (0.5, 0.5, 0)
    vi = np.full((2, 4, 4), -1, np.int32)
    vi[0, 0] = (1, 2, 3, 4)
    vi[0, 1] = (5, 6, 7, 8)
    d = np.array([[0.5, 1.0, 1.5, 2.25, 2.75, 3.5], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6]], np.float32)
    out = oracle.find_visited_cells(nv, vc, bc, hd, vi, d)
    assert out["mask"].tolist() == [[False, True, True, False, True, False], [False] * 6]
    assert out["cell_indices"][0].tolist() == [-1, 7, 7, -1, 9, -1]
    np.testing.assert_allclose(out["barycentric_coordinates"][0, 2], (0.5, 0.5, 0))
    np.testing.assert_allclose(out["barycentric_coordinates"][0, 4], (0.25, 0.25, 0.5))
    assert out["vertex_indices"][0, 4].tolist() == [5, 6, 7, 8]
    assert np.all(out["vertex_indices"][1] == -1) and np.all(out["barycentric_coordinates"][1] == 0)


def test_interpolate_values_einsum(oracle):
    """The reference's definition of gather parity: forward == einsum('jrbi,rbi->rbj') and
    backward == autograd of it (tests/test_tetrahedra_tracer.py:400-416,436-453)."""
    import torch

    rng = np.random.default_rng(0)
    V, R, S = 500, 16, 24
    vi = rng.integers(0, V, (R, S, 4)).astype(np.int32)
    vi[rng.random((R, S)) < 0.2] = -1
    bc = rng.random((R, S, 3)).astype(np.float32) * 0.3
    bc[vi[..., 0] < 0] = 0
    field = rng.standard_normal((64, V)).astype(np.float32)
    val = oracle.interpolate_values(vi, bc, field)
    assert val.shape == (R, S, 64)
    tvi = torch.from_numpy(vi)
    tf = torch.from_numpy(field).requires_grad_(True)
    safe = tvi.long().clamp_min(0)
    gathered = torch.where(tvi >= 0, tf[:, safe], torch.zeros(()))
    w = torch.from_numpy(np.concatenate([1 - bc.sum(-1, keepdims=True), bc], -1))
    gt = torch.einsum("jrbi,rbi->rbj", gathered, w)
    np.testing.assert_allclose(val, gt.detach().numpy(), rtol=1e-5, atol=1e-5)
    gt.sum().backward()
    grad = oracle.interpolate_values_backward(vi, bc, field, np.ones((R, S, 64), np.float32))
    np.testing.assert_allclose(grad, tf.grad.numpy(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError, match="Unsupported interpolation dimension"):
        oracle.interpolate_values(np.zeros((4, 5), np.int32), np.zeros((4, 4), np.float32), field)
