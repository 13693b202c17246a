"""GPU tests specific to the adjacency-walk fast path: large ray sets against the oracle
(every ray the walk certifies must equal sort + post_process_tetrahedra of the all-hits list),
the fraction it certifies, size-independent properties at the benchmark size, and the
equality of the two GPU paths on a full 800x800 frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances", "barycentric_coordinates")


def _tracer(tn, device, pts, cells, walk):
    import torch

    tr = tn.TetrahedraTracer(device)
    tr.set_option("walk", 2 if walk else 0)  # 2 = walk for any batch size (default: >= 16384 rays)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    return tr


def _trace(tr, device, o, d, M):
    import torch

    out = tr.trace_rays(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device), M)
    return {k: v.cpu().numpy() for k, v in out.items()}


def _bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_walk_vs_oracle_many_rays(tn, device, oracle, scenes):
    pts, cells = scenes.random_mesh(5000, 21)
    ot = oracle.OracleTracer(use_bvh=True)
    ot.load_tetrahedra(pts, cells)
    tr = _tracer(tn, device, pts, cells, 1)
    c = (0.5, 0.5, 0.5)
    sets = {
        "camera_outside": scenes.pinhole_rays(256, 256, eye=(0.5, 2.3, 0.6), lookat=c),
        "camera_inside": scenes.pinhole_rays(256, 256, eye=(0.45, 0.55, 0.5), lookat=(0.9, 0.1, 0.4), fov_y=90.0),
        "outside_in": scenes.outside_in_rays(65536, 31),
        "inside_out": scenes.inside_out_rays(65536, 32),
    }
    for name, (o, d) in sets.items():
        want = ot.trace_rays(o, d, 256)
        got = _trace(tr, device, o, d, 256)
        for k in KEYS:
            assert _bits_equal(got[k], want[k]), f"{name}: {k} differs from the oracle"
        st = tr.trace_stats()
        assert st["walk"] + st["general"] == len(o)
        assert st["walk"] >= 0.85 * len(o), (name, st)


def test_walk_equals_general_full_frame(tn, device, scenes):
    """BASELINE configs[1] size (100k tets, 800x800, M=512 is 17 GB per path -- compared in
    row blocks at M=256): the two GPU paths must agree bit for bit; plus size-independent
    properties of the result."""
    import torch

    pts, cells = scenes.random_mesh(15000, 0)
    c = np.array([0.5, 0.5, 0.5], np.float32)
    o, d = scenes.pinhole_rays(800, 800, eye=tuple(c + np.array([0, 2.0, 0], np.float32)), lookat=tuple(c))
    tw = _tracer(tn, device, pts, cells, 1)
    tg = _tracer(tn, device, pts, cells, 0)
    M = 256
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    total = 0
    for s in range(0, len(o), 160000):
        a = tw.trace_rays(to[s:s + 160000].contiguous(), td[s:s + 160000].contiguous(), M)
        b = tg.trace_rays(to[s:s + 160000].contiguous(), td[s:s + 160000].contiguous(), M)
        for k in KEYS:
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), k
        n = a["num_visited_cells"]
        total += int(n.sum())
        hd = a["hit_distances"]
        idx = torch.arange(M, device=device)[None, :]
        live = idx < n[:, None]
        # A grazed hull face without a partner is carried along by the reference's swap step and
        # finally "pairs" with the exit hull face through their common EMPTY tet
        # (get_common_tetrahedra, optix_trace_rays.cu:22-37): such a closing segment has cell -1.
        # Everything else: positive length >= eps, ascending, non-overlapping; tails constant.
        real = live & (a["visited_cells"] != -1)
        assert int((live & ~real).sum()) < 1e-4 * int(live.sum())
        assert bool(((hd[..., 1] - hd[..., 0] >= 1e-6 * 0.999) | ~real).all())
        # (tie handling may also re-order a few neighbours: the swap step of the reference moves a
        #  matched face forward, optix_trace_rays.cu:244-250 -- allowed for < 1e-4 of the segments)
        overlap = (hd[:, 1:, 0] < hd[:, :-1, 1]) & real[:, 1:] & real[:, :-1]
        assert int(overlap.sum()) < 1e-4 * int(real.sum())
        assert bool((a["visited_cells"][~live] == -1).all()) and bool((a["vertex_indices"][~live] == -1).all())
        assert bool((hd[~live] == 0).all()) and bool((a["barycentric_coordinates"][~live] == 0).all())
        # live segments name real tets and their own vertices
        vc = a["visited_cells"][real].long()
        assert bool((vc >= 0).all()) and bool((vc < len(cells)).all())
        tc = torch.from_numpy(cells).to(device)[vc].sort(-1).values
        assert torch.equal(tc, a["vertex_indices"][real].sort(-1).values)
        # entry/exit barycentrics are convex weights (within rounding)
        bc = a["barycentric_coordinates"][real]
        assert bool((bc > -1e-4).all()) and bool((bc.sum(-1) < 1 + 1e-4).all())
        del a, b
    assert total > 10_000_000
    st = tw.trace_stats()
    assert st["walk"] > 0.9 * (len(o) % 160000 or 160000)


def test_walk_handles_small_M_and_misses(tn, device, oracle, scenes):
    pts, cells = scenes.random_mesh(1200, 8)
    ot = oracle.OracleTracer(use_bvh=True)
    ot.load_tetrahedra(pts, cells)
    tr = _tracer(tn, device, pts, cells, 1)
    o, d = scenes.outside_in_rays(3000, 14)
    d[::3] = -d[::3]  # a third of the rays point away from the mesh
    for M in (2, 4, 8, 128):
        want = ot.trace_rays(o, d, M)
        got = _trace(tr, device, o, d, M)
        for k in KEYS:
            assert _bits_equal(got[k], want[k]), f"M={M}: {k}"


def test_dense_tails_off_keeps_the_valid_prefix(tn, device, scenes):
    """Non-reference option dense_tails=0: slots < num_visited are bit-identical to the dense result,
    slots beyond stay untouched on walked rows (here: the sentinel the test pre-fills... torch.empty
    is not controllable, so only the prefix is compared), and find_visited_cells -- which reads rows
    through num_visited only -- returns identical matches."""
    import torch

    pts, cells = scenes.random_mesh(4000, 5)
    tr = _tracer(tn, device, pts, cells, 1)
    o, d = scenes.outside_in_rays(40000, 3)
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    M = 128
    dense = tr.trace_rays(to, td, M)
    tr.set_option("dense_tails", 0)
    lean = tr.trace_rays(to, td, M)
    tr.set_option("dense_tails", 1)
    n = dense["num_visited_cells"]
    assert torch.equal(n, lean["num_visited_cells"])
    valid = torch.arange(M, device=device)[None] < n[:, None]
    for k in KEYS[1:]:
        a, b = dense[k], lean[k]
        m = valid.reshape(valid.shape + (1,) * (a.dim() - 2)).expand_as(a)
        assert torch.equal(a[m].view(torch.int32), b[m].view(torch.int32)), k
    hit = n > 0
    near = dense["hit_distances"][:, 0, 0]
    far = torch.gather(dense["hit_distances"][:, :, 1], 1, (n[:, None].long() - 1).clamp_min(0))[:, 0]
    ts = torch.linspace(0, 1, 64, device=device)[None]
    dist = torch.where(hit[:, None], near[:, None] * (1 - ts) + far[:, None] * ts, ts).contiguous()
    args = lambda r: (r["num_visited_cells"], r["visited_cells"], r["barycentric_coordinates"], r["hit_distances"],
                      r["vertex_indices"], dist)
    m1, m2 = tr.find_visited_cells(*args(dense)), tr.find_visited_cells(*args(lean))
    for k in m1:
        assert torch.equal(m1[k], m2[k]), k


def test_literal_pairing_of_the_log_equals_bvh_fallback(tn, device, oracle, scenes):
    """Chains whose ORDER the walk cannot certify (a gap below eps, a tie, an inversion) are not re-traced through the
    BVH: their logged hits go through the literal sort + pairing.  That route, the BVH route (option literal = 0),
    the variant with the speculative fill forced low and the oracle must agree bit for bit -- on a mesh / ray set dense enough
    that hundreds of rays take it."""
    import torch

    pts, cells = scenes.random_mesh(20000, 11)
    o, d = scenes.outside_in_rays(120000, 12)
    tr = _tracer(tn, device, pts, cells, 1)
    a = _trace(tr, device, o, d, 512)
    reasons = tr.flag_reasons()
    # 7 = chains whose order the walk does not certify; 13 = paired literally from the log
    assert reasons.get(13, 0) > 100 and reasons.get(13, 0) == reasons.get(7, 0), reasons
    st = tr.trace_stats()
    assert st["walk"] + st["general"] == len(o)
    tr.set_option("literal", 0)
    b = _trace(tr, device, o, d, 512)
    assert 13 not in tr.flag_reasons()
    tr.set_option("literal", 1)
    tr.set_option("spec_fill", 1)
    tr.set_option("spec_k0", 32)     # speculative fill from slot 32 on: the literal rows overwrite it
    c = _trace(tr, device, o, d, 512)
    for k in KEYS:
        assert _bits_equal(a[k], b[k]), k
        assert _bits_equal(a[k], c[k]), k
    # and against the oracle on a slice that contains such rays
    ot = oracle.OracleTracer(use_bvh=True)
    ot.load_tetrahedra(pts, cells)
    want = ot.trace_rays(o[:30000], d[:30000], 512)
    for k in KEYS:
        assert _bits_equal(a[k][:30000], want[k]), k


def test_end_of_chain_rules_agree_with_the_literal_kernel(tn, device, scenes, bottle):
    """Round 6's rules A-D of the walk's order test (tn_trace_walk.hip: struct OrderR6; tests/cert_model.py) certify rays that round 5
    handed to the literal pairing kernel: an inverted last pair, a run of short gaps at the entry face, an inverted first pair
    (whose second segment the reference loses: the writer's drop flag).  Option cert_ends = 0 restores round 5's rules, so the
    same rays come once from k_write_segments and once from k_postprocess_log -- two independent implementations -- and the
    rows must be identical, every tail byte, for both writer tables, dense and compact rows, on meshes where such rays are
    common (the bottle's zero-volume tets, a lattice, vertex twins), ray counts that are not multiples of 8 / 64 / 256 and a
    poisoned allocator."""
    import torch

    meshes = [("random", scenes.random_mesh(5000, 21)), ("bottle", (bottle["vertices"], bottle["cells"])),
              ("lattice", scenes.grid_mesh(9, 0.0)), ("near-duplicates", scenes.near_duplicates_mesh(2000, 1e-7)),
              ("jittered lattice", scenes.grid_mesh(16, 1e-6))]
    moved = 0
    for name, (pts, cells) in meshes:
        lo, hi = pts.min(0), pts.max(0)
        o, d = scenes.outside_in_rays(30011, 23)
        t = o + d
        o, t = lo + (hi - lo) * o, lo + (hi - lo) * t
        d = t - o
        o, d = o.astype(np.float32), (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        o[5000:5600] += 100.0                               # a stretch of rays that miss
        to, td = torch.from_numpy(np.ascontiguousarray(o)).to(device), torch.from_numpy(np.ascontiguousarray(d)).to(device)
        for table in (1, 2):
            res = {}
            for ends in (0, 3, 1):                          # round 5's pairwise test | + rules A-C | round 6's cluster test (A-D)
                tr = tn.TetrahedraTracer(device)
                tr.set_option("walk", 2)
                tr.set_option("writer_table", table)
                tr.set_option("cert_ends", ends)
                tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
                junk = torch.full((64 << 20,), 0x7FC12345, dtype=torch.int32, device=device)   # poison what the allocator hands out next
                del junk
                res[ends] = (tr.trace_rays(to, td, 256), tr.trace_rays(to, td, 256, compact_rows=True), tr.flag_reasons().get(13, 0))
            want, want_c, lit0 = res[0]
            n = want["num_visited_cells"]
            valid = torch.arange(256, device=device)[None] < n[:, None]
            for ends in (3, 1):
                got, got_c, lit1 = res[ends]
                assert lit1 <= lit0, (name, ends, lit0, lit1)
                for k in KEYS:
                    assert torch.equal(got[k].view(torch.int32), want[k].view(torch.int32)), (name, table, ends, k)
                assert torch.equal(got_c["num_visited_cells"], n)
                for k in KEYS[1:]:
                    a, b = got_c[k], want_c[k]
                    m = valid.reshape(valid.shape + (1,) * (a.dim() - 2)).expand_as(a)
                    assert torch.equal(a[m].view(torch.int32), b[m].view(torch.int32)), (name, table, ends, k, "compact")
            assert res[1][2] <= res[3][2], (name, res[3][2], res[1][2])      # the cluster test certifies at least what A-C do
            moved += (lit0 - res[1][2]) if table == 1 else 0
    assert moved > 200, moved                               # the rules really move rays from the literal kernel to the writer


def test_writer_tables_agree(tn, device, scenes, bottle):
    """The segment writer reads one record per (tet, entry face) on meshes whose table the L2s hold and one per TET on
    larger ones (tn_common.h: WalkCold / WalkTet; option writer_table forces either): the same rows bit for bit -- also
    with compact rows and on the reference's bottle (375 zero-volume tets) -- and the same re-assembled build records."""
    import torch

    meshes = [("random", scenes.random_mesh(5000, 21)), ("bottle", (bottle["vertices"], bottle["cells"])),
              ("lattice", scenes.grid_mesh(9, 0.0)), ("near-duplicates", scenes.near_duplicates_mesh(2000, 1e-7))]
    for name, (pts, cells) in meshes:
        lo, hi = pts.min(0), pts.max(0)
        o, d = scenes.outside_in_rays(30000, 17)            # for the unit cube: mapped onto the mesh's bounding box
        t = o + d
        o, t = lo + (hi - lo) * o, lo + (hi - lo) * t
        d = t - o
        o, d = o.astype(np.float32), (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        to, td = torch.from_numpy(np.ascontiguousarray(o)).to(device), torch.from_numpy(np.ascontiguousarray(d)).to(device)
        outs, tables = [], []
        for table in (1, 2):
            tr = tn.TetrahedraTracer(device)
            tr.set_option("walk", 2)
            tr.set_option("writer_table", table)
            tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
            outs.append((tr.trace_rays(to, td, 256), tr.trace_rays(to, td, 256, compact_rows=True)))
            tables.append(tr.build_table(2))
            assert tr.trace_stats()["walk"] > 0.2 * len(o), (name, tr.trace_stats())     # (the bottle: 375 zero-volume tets, two thirds literal)
        assert torch.equal(tables[0], tables[1]), name
        for k in KEYS:
            assert torch.equal(outs[0][0][k].view(torch.int32), outs[1][0][k].view(torch.int32)), (name, k)
        n = outs[0][1]["num_visited_cells"]
        assert torch.equal(n, outs[1][1]["num_visited_cells"]) and torch.equal(n, outs[0][0]["num_visited_cells"])
        valid = torch.arange(256, device=device)[None] < n[:, None]
        for k in KEYS[1:]:
            a, b, c = outs[0][1][k], outs[1][1][k], outs[0][0][k]
            m = valid.reshape(valid.shape + (1,) * (a.dim() - 2)).expand_as(a)
            assert torch.equal(a[m].view(torch.int32), b[m].view(torch.int32)) and torch.equal(a[m].view(torch.int32), c[m].view(torch.int32)), (name, k)


def test_literal_sort_branches_agree(tn, device, scenes, bottle):
    """The hits the walk logged for a ray whose order it does not certify are nearly sorted: the pairing kernel runs odd-even
    transposition passes until one swaps nothing and hands chains that are still unsorted after "literal_sort_passes" (8) to
    the bitonic network.  Natural rays never need the network, so both branches are forced: 0 passes (always the network, the
    code of rounds 1-3), 1 pass (the network takes over half-sorted lists) and the default must write the same rows bit for
    bit -- on meshes where most rays are literal (the bottle's zero-volume tets, a lattice, vertex twins) and against the BVH
    path, which sorts unordered hits with the network alone."""
    import torch

    meshes = [("bottle", (bottle["vertices"], bottle["cells"])), ("lattice", scenes.grid_mesh(9, 0.0)),
              ("near-duplicates", scenes.near_duplicates_mesh(2000, 1e-7)), ("random", scenes.random_mesh(5000, 22))]
    literal_total = 0
    for name, (pts, cells) in meshes:
        lo, hi = pts.min(0), pts.max(0)
        o, d = scenes.outside_in_rays(20000, 19)
        t = o + d
        o, t = lo + (hi - lo) * o, lo + (hi - lo) * t
        d = t - o
        o, d = o.astype(np.float32), (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        to, td = torch.from_numpy(np.ascontiguousarray(o)).to(device), torch.from_numpy(np.ascontiguousarray(d)).to(device)
        tr = tn.TetrahedraTracer(device)
        tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
        tr.set_option("walk", 0)
        want = tr.trace_rays(to, td, 256)                    # BVH path
        tr.set_option("walk", 2)
        for passes in (8, 0, 1):
            tr.set_option("literal_sort_passes", passes)
            got = tr.trace_rays(to, td, 256)
            reasons = tr.flag_reasons()
            for k in KEYS:
                assert torch.equal(got[k].view(torch.int32), want[k].view(torch.int32)), (name, passes, k)
        literal_total += reasons.get(13, 0)
    assert literal_total > 5000, literal_total               # rays that went through the literal pairing of the log


def test_cross_check_hand_over_paths(tn, device, scenes):
    """The always-on count cross-check never finds a mismatch on its own, so its hand-over is exercised by injection
    (option verify_inject: every checked ray counts as a mismatch): in the late form (one-chunk call: the check runs beside the
    writer and the fill, the rays are re-traced through the BVH AFTER their rows were written as certified) and in the serial
    form (chunked call: the rows are skipped by the writer and the fill) the rows must come out bit-identical, dense and
    compact, and every checked ray must be counted as handed over."""
    import torch

    pts, cells = scenes.random_mesh(6000, 13)
    o, d = scenes.outside_in_rays(60000, 14)
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    tr = _tracer(tn, device, pts, cells, 1)
    tr.set_option("verify_stride", 0)
    ref = tr.trace_rays(to, td, 256)
    for chunked in (False, True):
        tr.set_option("log_cap_mb", 48 if chunked else 0)
        for stride in (1, 7):
            tr.set_option("verify_stride", stride)
            tr.set_option("verify_inject", 1)
            got = tr.trace_rays(to, td, 256)
            why = tr.flag_reasons()
            assert why.get(15, 0) > 0.8 * len(o) / stride and why.get(14, 0) == why.get(15, 0), (chunked, stride, why)
            for k in KEYS:
                assert torch.equal(ref[k].view(torch.int32), got[k].view(torch.int32)), (chunked, stride, k)
            lean = tr.trace_rays(to, td, 256, compact_rows=True)
            n = ref["num_visited_cells"]
            assert torch.equal(n, lean["num_visited_cells"])
            valid = torch.arange(256, device=device)[None] < n[:, None]
            for k in KEYS[1:]:
                a, b = ref[k], lean[k]
                m = valid.reshape(valid.shape + (1,) * (a.dim() - 2)).expand_as(a)
                assert torch.equal(a[m].view(torch.int32), b[m].view(torch.int32)), (chunked, stride, k)
            tr.set_option("verify_inject", 0)
            clean = tr.trace_rays(to, td, 256)
            assert tr.flag_reasons().get(14, 0) == 0
            for k in KEYS:
                assert torch.equal(ref[k].view(torch.int32), clean[k].view(torch.int32)), (chunked, stride, k)


def test_risk_classes_of_the_certification_are_cross_checked(tn, device, scenes):
    """Round 5: a certified ray inside the WIDE band (16 rounding distances) of a guard that hands rays over at 8 -- a hull edge,
    an edge of a thin-neighbourhood tet -- is re-counted by the BVH cross-check, every one of them (the blind sample takes one
    ray in 1024).  On meshes with thin tets and with rays aimed at vertices the classes are populated; the check runs on all
    their rays, finds no mismatch, changes no output; option verify_risk = 0 switches it off; the chunked schedule does the same."""
    import torch

    for name, (pts, cells) in (("near_duplicates", scenes.near_duplicates_mesh()), ("shells", scenes.shells_mesh()),
                               ("random", scenes.random_mesh(5000, 3))):
        lo, hi = pts.min(0), pts.max(0)
        ext = float((hi - lo).max())
        vo, vd = scenes.vertex_to_vertex_rays(pts, 20000, 5, extend=1.2 * ext)
        oo, od = scenes.outside_in_rays(20000, 6)
        oo = ((oo - 0.5) * ext + 0.5 * (lo + hi)).astype(np.float32)
        o, d = np.concatenate([vo, oo]), np.concatenate([vd, od])
        to, td = torch.from_numpy(np.ascontiguousarray(o)).to(device), torch.from_numpy(np.ascontiguousarray(d)).to(device)
        tr = _tracer(tn, device, pts, cells, 2)
        tr.set_option("verify_risk", 0)
        ref = tr.trace_rays(to, td, 512)
        off = tr.cross_check()
        assert off["risk"] == {"hull_near_miss_rays": 0, "thin_neighbourhood_rays": 0, "checked": 0, "mismatches": 0}, off
        tr.set_option("verify_risk", 1)
        for chunked in (False, True):
            tr.set_option("log_cap_mb", 96 if chunked else 0)
            got = tr.trace_rays(to, td, 512)
            xc = tr.cross_check()
            risk = xc["risk"]
            listed = risk["hull_near_miss_rays"] + risk["thin_neighbourhood_rays"]
            print(name, "chunked" if chunked else "one chunk", xc, tr.trace_stats())
            assert xc["mismatches"] == 0 and risk["mismatches"] == 0, (name, xc)
            assert listed - listed // 128 - 8 <= risk["checked"] <= listed, (name, xc)   # all but those the blind sample holds
            if name != "random":
                assert listed > 20, (name, xc)
            for k in KEYS:
                assert torch.equal(ref[k].view(torch.int32), got[k].view(torch.int32)), (name, chunked, k)


def _trace_into_sentinels(tn, tr, to, td, M, flags):
    """tn_trace_rays_ex into buffers pre-set to sentinels: shows which bytes a call writes."""
    import ctypes as C

    import torch

    dev = to.device
    R = to.shape[0]
    out = {"num_visited_cells": torch.full((R,), 0x5a5a5a5a, dtype=torch.int32, device=dev),
           "visited_cells": torch.full((R, M), 0x5a5a5a5a, dtype=torch.int32, device=dev),
           "barycentric_coordinates": torch.full((R, M, 2, 3), 12345.0, device=dev),
           "hit_distances": torch.full((R, M, 2), 12345.0, device=dev),
           "vertex_indices": torch.full((R, M, 4), 0x5a5a5a5a, dtype=torch.int32, device=dev)}
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    lib = tn.cpp._lib.load()
    tn.cpp._lib.check(lib.tn_trace_rays_ex(tr._h, R, M, p(to), p(td), p(out["num_visited_cells"]), p(out["visited_cells"]),
                                            p(out["barycentric_coordinates"]), p(out["hit_distances"]), p(out["vertex_indices"]),
                                            flags, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out


def test_which_bytes_a_call_writes(tn, device, scenes):
    """Into sentinel-filled buffers: a default call writes EVERY slot of every row (ids -1, floats 0 beyond num_visited, as the
    reference's torch::zeros + tail fill), on the walk path and on the BVH path; the per-call flag TN_TRACE_COMPACT_ROWS and
    the tracer option dense_tails = 0 leave every slot beyond num_visited untouched.  (Round 4 shipped, for a few commits, a
    flag that was initialised from itself: whether tails were written depended on the build.  This test pins it.)"""
    import torch

    pts, cells = scenes.random_mesh(6000, 13)
    tr = _tracer(tn, device, pts, cells, 1)
    M = 128
    for n_rays in (40000, 3000):                      # walk path / wave-per-ray BVH path
        o, d = scenes.outside_in_rays(n_rays, 3)
        d[::7] = -d[::7]
        to, td = torch.from_numpy(o).to(device), torch.from_numpy(np.ascontiguousarray(d)).to(device)
        dense = _trace_into_sentinels(tn, tr, to, td, M, 0)
        n = dense["num_visited_cells"]
        assert int((n == 0).sum()) > n_rays // 10 and int(n.max()) > 40 and int(n.max()) < M
        tail = torch.arange(M, device=device)[None] >= n[:, None]
        assert bool((dense["visited_cells"][tail] == -1).all()) and bool((dense["vertex_indices"][tail] == -1).all())
        assert bool((dense["hit_distances"][tail] == 0).all()) and bool((dense["barycentric_coordinates"][tail] == 0).all())
        for how in ("flag", "option"):
            if how == "option":
                tr.set_option("dense_tails", 0)
            lean = _trace_into_sentinels(tn, tr, to, td, M, 1 if how == "flag" else 0)
            tr.set_option("dense_tails", 1)
            assert torch.equal(lean["num_visited_cells"], n), how
            for k in KEYS[1:]:
                a, b = dense[k], lean[k]
                t = tail.reshape(tail.shape + (1,) * (a.dim() - 2)).expand_as(a)
                assert torch.equal(a[~t].view(torch.int32), b[~t].view(torch.int32)), (n_rays, how, k)
                sent = 0x5a5a5a5a if b.dtype == torch.int32 else 12345.0
                if n_rays == 40000 and how == "flag":
                    # walked rows: the writer may pad a row's segments up to the next multiple of 32 slots -- it does not with
                    # compact rows; literal / fallback rows are written by the BVH kernels: compact as well
                    pass
                assert bool((b[t] == sent).all()), (n_rays, how, k)


def test_compact_rows_on_the_small_batch_path(tn, device, scenes):
    """TN_TRACE_COMPACT_ROWS on the BVH path (batches below walk_min_rays: nerfstudio's 4096-ray training batches): the
    valid prefix of every row is bit-identical to the dense call, num_visited too, and the slots beyond are left alone
    (a 4096-ray batch otherwise writes 109 MB of constants nobody reads)."""
    import torch

    pts, cells = scenes.random_mesh(6000, 13)
    o, d = scenes.outside_in_rays(4096, 3)
    d[::5] = -d[::5]                                  # some rays miss
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(np.ascontiguousarray(d)).to(device)
    tr = _tracer(tn, device, pts, cells, 1)           # 4096 rays < walk_min_rays: the wave-per-ray BVH path
    M = 256
    dense = tr.trace_rays(to, td, M)
    lean = tr.trace_rays(to, td, M, compact_rows=True)
    n = dense["num_visited_cells"]
    assert torch.equal(n, lean["num_visited_cells"]) and int((n == 0).sum()) > 500 and int(n.max()) > 50
    valid = torch.arange(M, device=device)[None] < n[:, None]
    for k in KEYS[1:]:
        a, b = dense[k], lean[k]
        m = valid.reshape(valid.shape + (1,) * (a.dim() - 2)).expand_as(a)
        assert torch.equal(a[m].view(torch.int32), b[m].view(torch.int32)), k
    # the tails of the compact call were not written: pre-set rows keep their sentinel
    import ctypes as C
    sent = {k: torch.full_like(dense[k], 0x5a5a5a5a if dense[k].dtype == torch.int32 else 12345.0) for k in KEYS}
    lib = tn.cpp._lib.load()
    tn.cpp._lib.check(lib.tn_trace_rays_ex(tr._h, len(o), M, C.c_void_p(to.data_ptr()), C.c_void_p(td.data_ptr()),
                                            C.c_void_p(sent["num_visited_cells"].data_ptr()), C.c_void_p(sent["visited_cells"].data_ptr()),
                                            C.c_void_p(sent["barycentric_coordinates"].data_ptr()), C.c_void_p(sent["hit_distances"].data_ptr()),
                                            C.c_void_p(sent["vertex_indices"].data_ptr()), 1, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    assert bool((sent["visited_cells"][~valid] == 0x5a5a5a5a).all()) and bool((sent["hit_distances"][~valid] == 12345.0).all())
    assert torch.equal(sent["visited_cells"][valid], dense["visited_cells"][valid])


def test_chunked_log_equals_single_launch(tn, device, scenes):
    """Calls whose hit log would exceed the cap are walked and written in ray chunks: same bits."""
    pts, cells = scenes.random_mesh(6000, 13)
    o, d = scenes.outside_in_rays(50000, 14)
    tr = _tracer(tn, device, pts, cells, 1)
    a = _trace(tr, device, o, d, 256)
    tr.set_option("log_cap_mb", 48)   # 4096-ray chunks at M = 256 need 16 MB each: 12288 rays per chunk
    b = _trace(tr, device, o, d, 256)
    for k in KEYS:
        assert _bits_equal(a[k], b[k]), k
    st = tr.trace_stats()
    assert st["walk"] + st["general"] == len(o) and st["walk"] > 0.8 * len(o), st


def test_find_visited_cells_ray_index(tn, device, scenes):
    """Addition to the reference surface: matching a subset of the traced rays in place (ray_index) equals
    matching their compacted rows."""
    import torch

    pts, cells = scenes.random_mesh(3000, 8)
    tr = _tracer(tn, device, pts, cells, 1)
    o, d = scenes.outside_in_rays(20000, 9)
    out = tr.trace_rays(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device), 128)
    n = out["num_visited_cells"]
    idx = torch.nonzero(n > 0)[:, 0][::3].contiguous()
    near = out["hit_distances"][idx, 0, 0]
    far = torch.gather(out["hit_distances"][idx][:, :, 1], 1, (n[idx, None].long() - 1).clamp_min(0))[:, 0]
    ts = torch.linspace(0, 1, 96, device=device)[None]
    dist = (near[:, None] * (1 - ts) + far[:, None] * ts).contiguous()
    keys = ("num_visited_cells", "visited_cells", "barycentric_coordinates", "hit_distances", "vertex_indices")
    a = tr.find_visited_cells(*[out[k] for k in keys], dist, ray_index=idx.to(torch.int32))
    b = tr.find_visited_cells(*[out[k][idx].contiguous() for k in keys], dist)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert bool(a["mask"].any())



def _sphere_hull_mesh(scenes, n_hull_points, interior=3000, seed=11):
    """`n_hull_points` points on a sphere (all of them hull vertices: 2 n - 4 hull faces) + interior points."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n_hull_points, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    pts = np.concatenate([0.5 + 0.49 * v, 0.5 + 0.5 * (rng.random((interior, 3)) - 0.5)], 0).astype(np.float32)
    return np.ascontiguousarray(pts), scenes.delaunay_cells(np.ascontiguousarray(pts))


def _hull_faces(cells):
    f = np.concatenate([cells[:, [1, 2, 3]], cells[:, [0, 2, 3]], cells[:, [0, 1, 3]], cells[:, [0, 1, 2]]], 0)
    f.sort(1)
    return int((np.unique(f, axis=0, return_counts=True)[1] == 1).sum())


@pytest.mark.parametrize("mesh", ["cube", "random", "colmap_like", "sphere_1020", "sphere_1024", "sphere_1028", "sphere_notched"])
def test_flat_hull_search_equals_the_threaded_tree(tn, device, oracle, scenes, mesh):
    """Round 6: k_hull_entry finds the two crossed hull faces through a flat two-level box table staged in LDS (hulls of at
    most 1024 faces: 66 KB of dynamic LDS at the limit, above 64 KB only by opt-in) instead of the threaded tree; both must
    give the same rows, bit for bit, and the oracle's on a sample.  Hull sizes around the limit (1020 / 1024 flat, 1028
    tree), a notched (non-convex) hull and the 12-face cube (one group, six leaves)."""
    if mesh == "cube":
        pts, cells = scenes.cube_mesh()
    elif mesh == "random":
        pts, cells = scenes.random_mesh(6000, 3)
    elif mesh == "colmap_like":
        pts, cells = scenes.colmap_like_mesh()
    elif mesh == "sphere_notched":
        # a convex hull with tets removed at the boundary: lines through a notch cross the hull four times (-> BVH path)
        pts, cells = _sphere_hull_mesh(scenes, 300)
        f = np.concatenate([cells[:, [1, 2, 3]], cells[:, [0, 2, 3]], cells[:, [0, 1, 3]], cells[:, [0, 1, 2]]], 0)
        f.sort(1)
        _, inv, cnt = np.unique(f, axis=0, return_inverse=True, return_counts=True)
        on_hull = (cnt[inv.reshape(-1)] == 1).reshape(4, -1).sum(0)
        cells = np.ascontiguousarray(np.delete(cells, np.nonzero(on_hull == 1)[0][::7], axis=0))
        assert _hull_faces(cells) > 596
    else:
        n = {"sphere_1020": 512, "sphere_1024": 514, "sphere_1028": 516}[mesh]
        pts, cells = _sphere_hull_mesh(scenes, n)
        assert _hull_faces(cells) == 2 * n - 4
    rays = [scenes.outside_in_rays(30000, 5), scenes.inside_out_rays(20000, 6)]
    o = np.ascontiguousarray(np.concatenate([r[0] for r in rays], 0))
    d = np.ascontiguousarray(np.concatenate([r[1] for r in rays], 0))
    M = 256
    outs = {}
    for flat in (1, 0):
        tr = _tracer(tn, device, pts, cells, walk=True)
        tr.set_option("hull_flat", flat)
        outs[flat] = _trace(tr, device, o, d, M)
        st = tr.trace_stats()
        assert st["walk"] > 0.5 * len(o) or mesh in ("cube", "sphere_notched"), st
    for k in KEYS:
        assert _bits_equal(outs[1][k], outs[0][k]), k
    ot = oracle.OracleTracer(use_bvh=True)
    ot.load_tetrahedra(pts, cells)
    sl = slice(0, len(o), 25)
    want = ot.trace_rays(np.ascontiguousarray(o[sl]), np.ascontiguousarray(d[sl]), M)
    for k in KEYS:
        assert _bits_equal(outs[1][k][sl], want[k]), k


@pytest.mark.parametrize("M", [4, 16, 32, 64, 128, 256, 512, 2048])
def test_fill_rows_writes_the_tail_constants_and_nothing_else(tn, device, M):
    """tn_fill_rows (k_fill_linear: one linear stream per array, one 16-byte store per thread): slots >= first_slot of every
    row get the tail constants (cells / vertex ids 0xFFFFFFFF, distances / barycentrics 0), slots below keep their bytes;
    with and without the vertex-id array; a first slot that is not a multiple of 32 is refused; first_slot >= M is a no-op."""
    import torch

    cpp = tn.cpp
    R = 1237 if M <= 512 else 301
    for first in sorted({0, 32, 64, max(M - 32, 0)} & set(range(0, M + 1, 32))) + ([M] if M % 32 == 0 else []):
        for with_verts in (True, False):
            vc = torch.full((R, M), 7, dtype=torch.int32, device=device)
            bc = torch.full((R, M, 2, 3), 3.5, dtype=torch.float32, device=device)
            hd = torch.full((R, M, 2), 2.5, dtype=torch.float32, device=device)
            vi = torch.full((R, M, 4), 9, dtype=torch.int32, device=device) if with_verts else None
            cpp.fill_rows(vc, bc, hd, vi, first)
            torch.cuda.synchronize()
            f = min(first, M)
            assert bool((vc[:, :f] == 7).all()) and bool((vc[:, f:] == -1).all()), (M, first)
            assert bool((bc[:, :f] == 3.5).all()) and bool((bc[:, f:] == 0).all()), (M, first)
            assert bool((hd[:, :f] == 2.5).all()) and bool((hd[:, f:] == 0).all()), (M, first)
            if with_verts:
                assert bool((vi[:, :f] == 9).all()) and bool((vi[:, f:] == -1).all()), (M, first)
    if M >= 64:
        vc = torch.zeros((4, M), dtype=torch.int32, device=device)
        with pytest.raises(RuntimeError):
            cpp.fill_rows(vc, torch.zeros((4, M, 2, 3), device=device), torch.zeros((4, M, 2), device=device), None, 16)


@pytest.mark.parametrize("M", [16, 64, 128])
def test_tail_fill_variants_agree_at_small_max_ray_triangles(tn, device, scenes, M):
    """The three tail fills (a block per row: default; one linear stream per array: -2, whose row lookup is per lane below
    M = 256; persistent waves: 512) and the BVH path write identical rows at M = 16 / 64 / 128 (rays with more than M - 1
    faces included: their rows come from the BVH path's overflow rule)."""
    pts, cells = scenes.random_mesh(4000, 9)
    o, d = scenes.outside_in_rays(30000, 10)
    ref = None
    for opts in ({"walk": 0}, {"fill_blocks": -1}, {"fill_blocks": -2}, {"fill_blocks": 512}, {"fill_blocks": -2, "spec_fill": 1, "spec_k0": 32}):
        tr = _tracer(tn, device, pts, cells, walk=opts.get("walk", 2) != 0)
        for k, v in opts.items():
            if k != "walk":
                tr.set_option(k, v)
        out = _trace(tr, device, o, d, M)
        if ref is None:
            ref = out
            continue
        for k in KEYS:
            assert _bits_equal(out[k], ref[k]), (opts, k)


def test_timing_option_serialises_the_same_kernels(tn, device, scenes):
    """set_option("timing", 1) runs the kernels of a one-chunk walk call serialised on the caller's stream with an event after
    each (tn_trace_timings): same rows, every stage that ran has a positive time, the stages add up to no more than the call."""
    import torch

    pts, cells = scenes.random_mesh(6000, 3)
    o, d = scenes.outside_in_rays(60000, 5)
    tr = _tracer(tn, device, pts, cells, walk=True)
    a = _trace(tr, device, o, d, 256)
    tr.set_option("timing", 1)
    oo, dd = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = tr.trace_rays(oo, dd, 256)
    e1.record()
    torch.cuda.synchronize()
    t = tr.trace_timings()
    tr.set_option("timing", 0)
    for k in KEYS:
        assert _bits_equal(out[k].cpu().numpy(), a[k]), k
    assert set(t) == set(tr.TIMING_KEYS)
    assert t["walk"] > 0 and t["segment_writer"] > 0 and t["tail_fill"] > 0, t
    assert all(v >= 0 for v in t.values()), t
    assert sum(t.values()) <= e0.elapsed_time(e1) * 1.05 + 0.05, (t, e0.elapsed_time(e1))
