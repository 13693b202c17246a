"""GPU tests of the DEVICE-side structure build of load_tetrahedra (csrc/tn_build.hip; reference:
src/tetrahedra_tracer.cpp:21-71 face table in first-seen order, :244-340 acceleration structure build).

Bar: the face table, face->tets, the 64-byte walk records and the hull tree are BIT-IDENTICAL to the single-threaded
host build (csrc/tn_mesh.cpp, itself checked against the oracle on the CPU by tests/test_host_build.py); the face BVH
is the same median-split tree up to ties, so it is checked through its invariants (every face in exactly one leaf, every
child box inside its parent's slot, references valid) and through the traces: the BVH all-hits path must return
bit-identical rows with either build."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")


def _tracer(tn, device, pts, cells, gpu_build, leaf_width=16):
    import torch

    tr = tn.TetrahedraTracer(device)
    tr.set_option("gpu_build", gpu_build)
    tr.set_option("leaf_width", leaf_width)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    return tr


def _meshes(scenes, bottle):
    yield "cube", scenes.cube_mesh()
    yield "bottle", (bottle["vertices"], bottle["cells"])       # 375 zero-volume tets
    yield "random_1500", scenes.random_mesh(1500, 1)
    yield "c2_15000", scenes.random_mesh(15000, 0)
    yield "lattice", scenes.grid_mesh(12)                       # cospherical points: ties, slivers
    yield "near_duplicates", scenes.near_duplicates_mesh(3000)


def _check_bvh(tr, F, leaf_width=16):
    child = tr.build_table(5).numpy().view(np.uint32).reshape(-1, 64)
    boxes = tr.build_table(6).numpy().view(np.float32).reshape(-1, 6, 64)
    leaf_id = tr.build_table(7).numpy().view(np.uint32).reshape(-1, leaf_width)
    leaf_tri = tr.build_table(8).numpy().view(np.float32).reshape(-1, 9, leaf_width)
    n_nodes, n_leaves = child.shape[0], leaf_id.shape[0]
    ids = leaf_id[leaf_id != 0xFFFFFFFF]
    assert len(ids) == F and len(np.unique(ids)) == F, "every face must sit in exactly one leaf"
    seen_nodes, seen_leaves = np.zeros(n_nodes, bool), np.zeros(n_leaves, bool)
    seen_nodes[0] = True
    for w in range(n_nodes):
        for i in range(64):
            ch = int(child[w, i])
            if ch == 0xFFFFFFFF:
                assert boxes[w, 0, i] == np.inf and boxes[w, 3, i] == -np.inf
                continue
            lo, hi = boxes[w, :3, i], boxes[w, 3:, i]
            if ch >> 31:
                l = ch & 0x7FFFFFFF
                assert l < n_leaves and not seen_leaves[l]
                seen_leaves[l] = True
                m = leaf_id[l] != 0xFFFFFFFF
                tri = leaf_tri[l][:, m].reshape(3, 3, -1)      # [vertex, axis, face]
                assert np.all(tri.min(axis=0).min(axis=1) >= lo) and np.all(tri.max(axis=0).max(axis=1) <= hi)
                assert np.all(tri.min(axis=0).min(axis=1) == lo) and np.all(tri.max(axis=0).max(axis=1) == hi)
            else:
                assert w < ch < n_nodes and not seen_nodes[ch], "children must have larger indices than their parent"
                seen_nodes[ch] = True
                m = child[ch] != 0xFFFFFFFF
                assert np.all(boxes[ch, :3][:, m].min(axis=1) == lo) and np.all(boxes[ch, 3:][:, m].max(axis=1) == hi)
    assert seen_nodes.all() and seen_leaves.all()


def test_device_build_matches_host_build(tn, device, scenes, bottle):
    import torch

    for name, (pts, cells) in _meshes(scenes, bottle):
        host = _tracer(tn, device, pts, cells, 0)
        dev = _tracer(tn, device, pts, cells, 1)
        for which, what in ((0, "faces"), (1, "face_tets"), (2, "walk records"), (3, "hull nodes"), (4, "hull triangles")):
            a, b = host.build_table(which).numpy(), dev.build_table(which).numpy()
            assert a.shape == b.shape, f"{name}: {what} size {a.shape} vs {b.shape}"
            diff = np.nonzero(a != b)[0]
            assert len(diff) == 0, f"{name}: {what} differ at byte {diff[:4]} ({len(diff)} bytes)"
        F = host.build_table(1).numel() // 8
        _check_bvh(dev, F)
        _check_bvh(host, F)
        # the other leaf widths (64 = one leaf per wave instruction, the layout of rounds 1 / 2a)
        for lw in (32, 64):
            other = _tracer(tn, device, pts, cells, 1, lw)
            _check_bvh(other, F, lw)
            other.set_option("walk", 0)
            o2, d2 = scenes.outside_in_rays(1000, 6)
            o2, d2 = torch.from_numpy(o2).to(device), torch.from_numpy(d2).to(device)
            host.set_option("walk", 0)
            a2, b2 = host.trace_rays(o2, d2, 256), other.trace_rays(o2, d2, 256)
            for k in KEYS:
                assert torch.equal(a2[k], b2[k]), f"{name}: {k} differs with {lw}-face leaves"
        # the BVH all-hits path on either structure: identical rows
        o, d = scenes.outside_in_rays(3000, 5)
        o, d = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
        host.set_option("walk", 0); dev.set_option("walk", 0)
        a, b = host.trace_rays(o, d, 256), dev.trace_rays(o, d, 256)
        for k in KEYS:
            assert torch.equal(a[k], b[k]), f"{name}: {k} differs between the host-built and the device-built BVH"
        # and the walk on the device-built records vs the BVH path
        dev.set_option("walk", 2)
        c = dev.trace_rays(o, d, 256)
        for k in KEYS:
            assert torch.equal(a[k], c[k]), f"{name}: {k} walk (device-built records) vs BVH path"


def test_device_build_errors(tn, device, scenes):
    import torch

    pts, cells = scenes.cube_mesh()
    tr = tn.TetrahedraTracer(device)
    bad = cells.copy(); bad[3, 2] = len(pts) + 5
    with pytest.raises(RuntimeError, match="out of bounds"):
        tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(bad).to(device))
    # the same face three times (reference: tetrahedra_tracer.cpp:63-65)
    extra = np.concatenate([cells, np.array([[cells[0, 1], cells[0, 2], cells[0, 3], len(pts) - 1]], dtype=cells.dtype),
                            np.array([[cells[0, 1], cells[0, 2], cells[0, 3], cells[5, 0]]], dtype=cells.dtype)])
    keys = {}
    for t in extra:
        for j in range(4):
            k = tuple(sorted((int(t[(j + 1) % 4]), int(t[(j + 2) % 4]), int(t[(j + 3) % 4]))))
            keys[k] = keys.get(k, 0) + 1
    if max(keys.values()) > 2:
        with pytest.raises(RuntimeError, match="more than two tetrahedra"):
            tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(extra).to(device))
    # a good mesh loads afterwards
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    out = tr.trace_rays(torch.tensor([[0.5, -2.0, 0.5]], device=device), torch.tensor([[0.0, 1.0, 0.0]], device=device), 64)
    assert int(out["num_visited_cells"][0]) > 0


def test_device_build_is_faster_at_1m_tets(tn, device, scenes):
    """load_tetrahedra at the C5 size: the device build must take well under the 0.2 s the round-1 verdict asked for."""
    import time
    import torch

    pts, cells = scenes.random_mesh(150000, 3)
    x, c = torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device)
    times = {}
    for gpu_build in (1, 0):
        tr = tn.TetrahedraTracer(device)
        tr.set_option("gpu_build", gpu_build)
        tr.load_tetrahedra(x, c)                       # warm-up (allocator, kernels)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.load_tetrahedra(x, c)
        torch.cuda.synchronize(); times[gpu_build] = time.perf_counter() - t0
    print(f"load_tetrahedra at {len(cells)} tets: device build {times[1]*1e3:.1f} ms, host build {times[0]*1e3:.1f} ms")
    assert times[1] < 0.2, times
