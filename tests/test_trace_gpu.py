"""GPU parity tests (run on a real MI355X through the C-ABI): trace_rays / find_visited_cells
of libtetranerf_hip.so against the CPU oracle on the same seeded inputs.

Bar: bit-exact on counts, hit tet indices, vertex ids and segment order; the float outputs
(hit distances, barycentrics) are compared bit-exactly too, because both sides evaluate the
same fp32 expression tree without contraction (the documented tolerance of 1e-5 is the
fallback bar stated in BASELINE.json, asserted as well)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS_INT = ("num_visited_cells", "visited_cells", "vertex_indices")
KEYS_FLT = ("hit_distances", "barycentric_coordinates")


def _to_np(out):
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def _gpu_tracer(tn, device, pts, cells, walk=1):
    import torch

    tr = tn.TetrahedraTracer(device)
    tr.set_option("walk", 2 if walk else 0)  # 2 = walk for any batch size (default: >= 16384 rays)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    return tr


def _trace(tr, device, o, d, M):
    import torch

    return _to_np(tr.trace_rays(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device), M))


def _assert_same(got, want, ctx=""):
    for k in KEYS_INT:
        np.testing.assert_array_equal(got[k], want[k], err_msg=f"{ctx} {k}")
    for k in KEYS_FLT:
        np.testing.assert_allclose(got[k], want[k], rtol=0, atol=1e-5, err_msg=f"{ctx} {k}")
        bad = np.nonzero(got[k].view(np.uint32) != want[k].view(np.uint32))
        assert len(bad[0]) == 0, f"{ctx} {k}: {len(bad[0])} values differ in bits (first at {[b[0] for b in bad]})"


def test_face_tables_match_oracle(tn, device, oracle, scenes, bottle):
    for pts, cells in (scenes.cube_mesh(), (bottle["vertices"], bottle["cells"]), scenes.random_mesh(5000, 7)):
        tr = _gpu_tracer(tn, device, pts, cells)
        faces, ft = tr.face_tables()
        ofaces, oft = oracle.build_faces(cells)
        np.testing.assert_array_equal(faces.numpy().view(np.uint32), ofaces)
        np.testing.assert_array_equal(ft.numpy().view(np.uint32), oft)


@pytest.mark.parametrize("walk", [0, 1])
def test_cube_rays(tn, device, oracle, scenes, walk):
    pts, cells = scenes.cube_mesh()
    o = np.array([[-0.05, 0.05, 0.05], [-0.05, 0.07, 0.13], [0.5, 0.5, 0.5], [2, 2, 2], [0.3, 0.2, -1]], np.float32)
    d = np.array([[1, 0, 0], [1.0, 0.11, 0.23], [0.3, -0.2, 0.9], [1, 0, 0], [0, 0, 1]], np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ot = oracle.OracleTracer()
    ot.load_tetrahedra(pts, cells)
    tr = _gpu_tracer(tn, device, pts, cells, walk)
    for M in (16, 64):
        _assert_same(_trace(tr, device, o, d, M), ot.trace_rays(o, d, M), f"cube M={M}")


@pytest.mark.parametrize("walk", [0, 1])
def test_bottle_c1(tn, device, oracle, scenes, bottle, walk):
    """Config C1 of BASELINE.json: bottle mesh, 64x64 rays, M=256, then 300 samples."""
    import torch

    pts, cells = bottle["vertices"], bottle["cells"]
    o, d = scenes.pinhole_rays(64, 64)
    ot = oracle.OracleTracer()
    ot.load_tetrahedra(pts, cells)
    want = ot.trace_rays(o, d, 256)
    tr = _gpu_tracer(tn, device, pts, cells, walk)
    out = tr.trace_rays(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device), 256)
    got = _to_np(out)
    _assert_same(got, want, "bottle")
    assert int((got["num_visited_cells"] > 0).sum()) == 146

    samples = np.ascontiguousarray(np.broadcast_to(np.linspace(0.90, 1.1, 300, dtype=np.float32), (len(o), 300)))
    inter = tr.find_visited_cells(out["num_visited_cells"], out["visited_cells"], out["barycentric_coordinates"],
                                  out["hit_distances"], out["vertex_indices"], torch.from_numpy(samples).to(device))
    winter = ot.find_visited_cells(want["num_visited_cells"], want["visited_cells"], want["barycentric_coordinates"],
                                   want["hit_distances"], want["vertex_indices"], samples)
    ginter = _to_np(inter)
    assert ginter["mask"].dtype == np.bool_
    for k in ("mask", "cell_indices", "vertex_indices"):
        np.testing.assert_array_equal(ginter[k], winter[k], err_msg=k)
    np.testing.assert_array_equal(ginter["barycentric_coordinates"].view(np.uint32),
                                  winter["barycentric_coordinates"].view(np.uint32))


@pytest.mark.parametrize("walk", [0, 1])
@pytest.mark.parametrize("npts,seed,M", [(2000, 3, 256), (15000, 0, 512)])
def test_random_mesh_parity(tn, device, oracle, scenes, npts, seed, M, walk):
    pts, cells = scenes.random_mesh(npts, seed)
    ot = oracle.OracleTracer(use_bvh=True)
    ot.load_tetrahedra(pts, cells)
    tr = _gpu_tracer(tn, device, pts, cells, walk)
    for name, (o, d) in (("outside_in", scenes.outside_in_rays(4096, seed + 10)),
                         ("inside_out", scenes.inside_out_rays(4096, seed + 20))):
        want = ot.trace_rays(o, d, M)
        got = _trace(tr, device, o, d, M)
        _assert_same(got, want, f"{name} V={npts}")
        assert want["num_visited_cells"].max() < M - 1
        st = tr.trace_stats()
        assert st["overflow"] == 0
        assert st["walk"] + st["general"] == len(o)
        if walk:
            # the walk certifies the bulk of the rays; the rest is re-traced by the general path
            assert st["walk"] > 0.8 * len(o), st
        else:
            assert st["walk"] == 0


@pytest.mark.parametrize("walk", [0, 1])
def test_overflow_keeps_nearest(tn, device, oracle, scenes, walk):
    """More than M-1 faces on a ray: both sides keep the M-1 nearest hits."""
    pts, cells = scenes.random_mesh(3000, 5)
    ot = oracle.OracleTracer(use_bvh=True)
    ot.load_tetrahedra(pts, cells)
    tr = _gpu_tracer(tn, device, pts, cells, walk)
    o, d = scenes.outside_in_rays(512, 6)
    for M in (32, 64):
        want = ot.trace_rays(o, d, M)
        got = _trace(tr, device, o, d, M)
        _assert_same(got, want, f"overflow M={M}")
    assert tr.trace_stats()["overflow"] > 0


@pytest.mark.parametrize("walk", [0, 1])
def test_degenerate_rays(tn, device, oracle, scenes, walk):
    """Rays through vertices / along edges and faces of the mesh, axis-parallel rays, zero and
    non-normalised directions."""
    pts, cells = scenes.random_mesh(800, 9)
    rng = np.random.default_rng(1)
    o_list, d_list = [], []
    for _ in range(300):
        a, b = rng.integers(0, len(pts), 2)
        if a == b:
            continue
        o = pts[a] + (pts[a] - pts[b]) * 2.0          # line through two mesh vertices
        o_list.append(o); d_list.append(pts[b] - pts[a])
    for c in cells[:200]:
        # through a vertex and the centroid of the opposite face (grazes edges of neighbours)
        tgt = pts[c[1:]].mean(0)
        o_list.append(pts[c[0]] + (pts[c[0]] - tgt) * 3); d_list.append(tgt - pts[c[0]])
        # in the plane of a face
        o_list.append(pts[c[0]] * 2 - pts[c[1]]); d_list.append(pts[c[2]] - pts[c[0]])
    for ax in range(3):
        for _ in range(50):
            o = rng.random(3).astype(np.float32); o[ax] = -1
            dd = np.zeros(3, np.float32); dd[ax] = 1
            o_list.append(o); d_list.append(dd)
    o_list.append(np.array([0.5, 0.5, 0.5], np.float32)); d_list.append(np.zeros(3, np.float32))
    o = np.ascontiguousarray(np.stack(o_list).astype(np.float32))
    d = np.ascontiguousarray(np.stack(d_list).astype(np.float32))
    ot = oracle.OracleTracer()
    ot.load_tetrahedra(pts, cells)
    tr = _gpu_tracer(tn, device, pts, cells, walk)
    want = ot.trace_rays(o, d, 256)
    got = _trace(tr, device, o, d, 256)
    _assert_same(got, want, "degenerate")
    assert tr.trace_stats()["serial"] > 0  # the literal tie-handling branch was exercised


def test_postprocess_crafted_ties(tn, device, oracle, scenes):
    """The dedupe/pairing stage alone on crafted sorted hit rows: duplicates of a face (OptiX may
    report a primitive more than once, cf. tests/test_sort.py:101-112), ties within eps,
    triple ties, unmatched faces, empty rows."""
    import torch

    pts, cells = scenes.random_mesh(300, 2)
    ot = oracle.OracleTracer()
    ot.load_tetrahedra(pts, cells)
    o, d = scenes.outside_in_rays(256, 3)
    base = ot.trace_rays(o, d, 128, with_raw=True)["raw"]
    rng = np.random.default_rng(4)
    M = 128
    cnt = base["count"].copy()
    ids = base["ids"].copy()
    t = base["t"].copy()
    uv = base["uv"].copy()
    for r in range(len(cnt)):
        c = int(cnt[r])
        if c < 4:
            continue
        mode = r % 6
        rows = list(zip(t[r, :c].tolist(), ids[r, :c].tolist(), uv[r, :c].tolist()))
        if mode == 0:      # duplicate random hits
            for k in rng.integers(0, c, 3):
                rows.insert(int(k), rows[int(k)])
        elif mode == 1:    # squash neighbouring distances into ties
            for k in rng.integers(1, c, 4):
                rows[int(k)] = (rows[int(k) - 1][0] + np.float32(3e-7), rows[int(k)][1], rows[int(k)][2])
        elif mode == 2:    # drop a face (unmatched neighbour)
            del rows[int(rng.integers(1, c - 1))]
        elif mode == 3:    # triple tie
            k = int(rng.integers(2, c))
            for q in (k - 1, k):
                rows[q] = (rows[k - 2][0], rows[q][1], rows[q][2])
        elif mode == 4:    # duplicate + drop
            k = int(rng.integers(1, c - 1))
            rows.insert(k, rows[k]); del rows[k + 2 if k + 2 < len(rows) else 0]
        rows = rows[:M - 1]
        rows.sort(key=lambda x: (np.float32(x[0]), x[1]))
        cnt[r] = len(rows)
        ids[r] = 0xFFFFFFFF; t[r] = 0; uv[r] = 0
        for k, (tt, ii, uu) in enumerate(rows):
            t[r, k] = tt; ids[r, k] = ii; uv[r, k] = uu
    want = oracle.postprocess(ot.faces, ot.face_tets, cnt, ids, t, uv)
    tr = _gpu_tracer(tn, device, pts, cells)
    got = _to_np(tr.postprocess_hits(torch.from_numpy(cnt.view(np.int32)).to(device),
                                     torch.from_numpy(ids.view(np.int32)).to(device),
                                     torch.from_numpy(t).to(device), torch.from_numpy(uv).to(device)))
    _assert_same(got, want, "crafted")


def test_find_visited_cells_edge_cases(tn, device, oracle, scenes):
    import torch

    pts, cells = scenes.random_mesh(1500, 12)
    ot = oracle.OracleTracer(use_bvh=True)
    ot.load_tetrahedra(pts, cells)
    o, d = scenes.outside_in_rays(700, 13)
    want = ot.trace_rays(o, d, 128)
    tr = _gpu_tracer(tn, device, pts, cells)
    out = tr.trace_rays(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device), 128)
    rng = np.random.default_rng(5)
    for S in (1, 63, 64, 257):
        s = np.sort(rng.random((len(o), S)).astype(np.float32) * 3.0, axis=-1)
        s[5] = s[5, ::-1]                   # a descending row: literal serial branch
        if S > 2:
            s[7, S // 2] = s[7, 0]          # one inversion
        s[9] = want["hit_distances"][9, 0, 0]   # all samples exactly on a boundary
        s = np.ascontiguousarray(s)
        g = _to_np(tr.find_visited_cells(out["num_visited_cells"], out["visited_cells"],
                                         out["barycentric_coordinates"], out["hit_distances"],
                                         out["vertex_indices"], torch.from_numpy(s).to(device)))
        w = ot.find_visited_cells(want["num_visited_cells"], want["visited_cells"], want["barycentric_coordinates"],
                                  want["hit_distances"], want["vertex_indices"], s)
        for k in ("mask", "cell_indices", "vertex_indices"):
            np.testing.assert_array_equal(g[k], w[k], err_msg=f"S={S} {k}")
        np.testing.assert_array_equal(g["barycentric_coordinates"].view(np.uint32),
                                      w["barycentric_coordinates"].view(np.uint32), err_msg=f"S={S}")


def test_errors(tn, device, scenes):
    import torch

    pts, cells = scenes.cube_mesh()
    tr = tn.TetrahedraTracer(device)
    o = torch.zeros((4, 3), device=device)
    with pytest.raises(RuntimeError, match="load_tetrahedra must be called first"):
        tr.trace_rays(o, o, 16)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    assert tr.device == device
    with pytest.raises(RuntimeError, match="power of 2"):
        tr.trace_rays(o, o, 24)
    with pytest.raises(RuntimeError, match="float32"):
        tr.trace_rays(o.double(), o, 16)
    with pytest.raises(RuntimeError, match="last dimension with size 3"):
        tr.trace_rays(torch.zeros((4, 2), device=device), o, 16)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        tr.trace_rays(o.cpu(), o, 16)
    with pytest.raises(RuntimeError, match="contiguous"):
        tr.trace_rays(torch.zeros((3, 4), device=device).t(), o, 16)
    bad = torch.tensor([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 5]], dtype=torch.int32, device=device)
    with pytest.raises(RuntimeError, match="shared by more than two"):
        tr.load_tetrahedra(torch.rand((6, 3), device=device), bad)
    # empty ray batch
    out = tn.TetrahedraTracer(device)
    out.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    res = out.trace_rays(torch.zeros((0, 3), device=device), torch.zeros((0, 3), device=device), 16)
    assert res["visited_cells"].shape == (0, 16)


def test_small_batch_overflow_relaunch(tn, device, oracle, scenes):
    """Small batches run the BVH path with LDS hit arrays sized for the mesh; a ray with more hits than they hold stops
    its traversal and is re-traced by a second launch with the full M-entry arrays.  Forcing tiny arrays (option
    lds_cap) sends most rays through that second launch: same bits as the one-launch path and as the oracle."""
    import torch

    pts, cells = scenes.random_mesh(6000, 21)
    o, d = scenes.outside_in_rays(3000, 22)
    tr = _gpu_tracer(tn, device, pts, cells, walk=0)
    ot = oracle.OracleTracer(use_bvh=True)
    ot.load_tetrahedra(pts, cells)
    want = ot.trace_rays(o, d, 256)
    tr.set_option("small_lds", 0)
    ref = _trace(tr, device, o, d, 256)
    _assert_same(ref, want, "full arrays")
    tr.set_option("small_lds", 1)
    for cap in (0, 128, 32, 8):
        tr.set_option("lds_cap", cap)
        got = _trace(tr, device, o, d, 256)
        _assert_same(got, want, f"lds_cap {cap}")
    assert int(want["num_visited_cells"].max()) > 40     # caps 32 and 8 really overflow
    tr.set_option("lds_cap", 0)
