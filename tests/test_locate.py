"""find_tetrahedra / trace_rays_triangles (exported by the reference's TetrahedraTracer but not
called by the model).  The oracle is pinned by the reference's own
known-answer test (tests/test_tetrahedra_tracer.py:270-344) and the on-ray property of its
triangles test (tests/test_tetrahedra_tracer_triangles.py:124-127); the HIP path is then
compared bit for bit with the oracle."""
import numpy as np
import pytest


def _mix(*args):
    out = 0
    for p, w in zip(args[::2], args[1::2]):
        out = out + p * np.float32(w)
    return out


def _cube_points(pts):
    return np.stack([
        _mix(pts[0], 0.23, pts[1], 0.27, pts[2], 0.21, pts[8], 0.29),
        _mix(pts[2], 0.23, pts[4], 0.24, pts[6], 0.26, pts[8], 0.27),
        _mix(pts[3], 0.39, pts[5], 0.41, pts[7], 0.09, pts[8], 0.11)], 0).astype(np.float32)


GT_COORDS = np.array([[0.23, 0.27, 0.21, 0.29], [0.23, 0.24, 0.26, 0.27], [0.39, 0.41, 0.09, 0.11]], np.float32)
GT_INDICES = np.array([[0, 1, 2, 8], [2, 4, 6, 8], [3, 5, 7, 8]], np.int32)


def _check_known_answer(out):
    assert out["tetrahedra"].tolist() == [0, 5, 11]
    b = out["barycentric_coordinates"]
    full = np.concatenate([1 - b.sum(-1, keepdims=True), b], -1)
    for i in range(3):
        order = np.argsort(out["vertex_indices"][i], kind="stable")
        assert out["vertex_indices"][i][order].tolist() == GT_INDICES[i].tolist()
        # torch.testing.assert_allclose defaults for float32: rtol 1.3e-6, atol 1e-5
        np.testing.assert_allclose(full[i][order], GT_COORDS[i], rtol=1.3e-6, atol=1e-5)
    assert out["valid_mask"].all()


def test_find_tetrahedra_known_answer_oracle(oracle, scenes):
    pts, cells = scenes.cube_mesh()
    tr = oracle.OracleTracer()
    tr.load_tetrahedra(pts, cells)
    _check_known_answer(tr.find_tetrahedra(_cube_points(pts)))
    outside = tr.find_tetrahedra(np.array([[2.0, 0.5, 0.5], [0.5, 0.5, -3.0]], np.float32))
    assert outside["tetrahedra"].tolist() == [-1, -1] and not outside["valid_mask"].any()
    assert np.all(outside["barycentric_coordinates"] == 0) and np.all(outside["vertex_indices"] == 0)


def test_triangles_on_ray_oracle(oracle, scenes, bottle):
    """The reference's check of trace_rays_triangles: hit points reconstructed from (u,v) lie on the ray."""
    tr = oracle.OracleTracer()
    tr.load_tetrahedra(bottle["vertices"], bottle["cells"])
    o, d = scenes.pinhole_rays(64, 64)
    out = tr.trace_rays_triangles(o, d, 256)
    n = out["num_visited_triangles"]
    assert int((n > 0).sum()) == 146
    V = bottle["vertices"]
    for r in np.nonzero(n > 0)[0][:60]:
        k = n[r]
        vi = out["vertex_indices"][r, :k]
        uv = out["barycentric_coordinates"][r, :k]
        w = np.concatenate([1 - uv.sum(-1, keepdims=True), uv], -1)
        p = (V[vi] * w[..., None]).sum(-2)
        t = out["hit_distances"][r, :k]
        assert np.all(np.diff(t) >= 0)
        np.testing.assert_allclose(p, o[r] + t[:, None] * d[r], atol=5e-5)
    assert np.all(out["visited_triangles"][np.arange(256)[None] >= n[:, None]] == 0)


@pytest.mark.gpu
def test_find_tetrahedra_gpu(tn, device, oracle, scenes):
    import torch

    pts, cells = scenes.cube_mesh()
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    out = {k: v.cpu().numpy() for k, v in tr.find_tetrahedra(torch.from_numpy(_cube_points(pts)).to(device)).items()}
    _check_known_answer(out)
    # random points in a random mesh: bit-equal to the oracle, and inside points are found in their tet
    pts, cells = scenes.random_mesh(1500, 4)
    rng = np.random.default_rng(0)
    q = (rng.random((4000, 3)) * 1.2 - 0.1).astype(np.float32)
    ot = oracle.OracleTracer()
    ot.load_tetrahedra(pts, cells)
    want = ot.find_tetrahedra(q)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    got = {k: v.cpu().numpy() for k, v in tr.find_tetrahedra(torch.from_numpy(q).to(device)).items()}
    np.testing.assert_array_equal(got["tetrahedra"], want["tetrahedra"])
    np.testing.assert_array_equal(got["vertex_indices"], want["vertex_indices"])
    np.testing.assert_array_equal(got["barycentric_coordinates"].view(np.uint32), want["barycentric_coordinates"].view(np.uint32))
    np.testing.assert_array_equal(got["valid_mask"], want["valid_mask"])
    found = got["tetrahedra"] >= 0
    assert found.mean() > 0.5
    b = got["barycentric_coordinates"][found]
    full = np.concatenate([1 - b.sum(-1, keepdims=True), b], -1)
    rec = (pts[got["vertex_indices"][found]] * full[..., None]).sum(-2)
    np.testing.assert_allclose(rec, q[found], atol=2e-5)
    # shape handling like the reference (positions [..., 3])
    out2 = tr.find_tetrahedra(torch.from_numpy(q[:12].reshape(3, 4, 3)).to(device))
    assert tuple(out2["tetrahedra"].shape) == (3, 4) and tuple(out2["vertex_indices"].shape) == (3, 4, 4)


@pytest.mark.gpu
def test_trace_rays_triangles_gpu(tn, device, oracle, scenes, bottle):
    import torch

    for pts, cells, (o, d), M in ((bottle["vertices"], bottle["cells"], scenes.pinhole_rays(64, 64), 256),
                                  (*scenes.random_mesh(2000, 6), scenes.outside_in_rays(2048, 7), 32)):
        ot = oracle.OracleTracer(use_bvh=True)
        ot.load_tetrahedra(pts, cells)
        want = ot.trace_rays_triangles(o, d, M)
        tr = tn.TetrahedraTracer(device)
        tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
        got = {k: v.cpu().numpy() for k, v in tr.trace_rays_triangles(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device), M).items()}
        for k in ("num_visited_triangles", "visited_triangles", "vertex_indices"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        for k in ("barycentric_coordinates", "hit_distances"):
            np.testing.assert_array_equal(got[k].view(np.uint32), want[k].view(np.uint32), err_msg=k)
