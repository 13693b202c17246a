"""GPU tests of round 5's render path: the device-side compaction of the hitting rays (tn_compact_hits), the persistent
one-launch render (tn_render_rays) against the chain of separate kernels it replaces -- BIT-IDENTICAL, because both run the
same device functions (csrc/tn_ray_ops.h, tn_mlp_fwd.h) -- and the absence of any host synchronisation in TetraRenderer.render
(reference: tetranerf/nerfstudio/model.py:520-662, which compacts with boolean indexing, :540-567)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def render():
    return importlib.import_module("tetra-nerf_amd.render")


@pytest.mark.parametrize("R", [1, 63, 64, 2047, 2048, 2049, 5000, 70001])
@pytest.mark.parametrize("p_hit", [0.0, 0.37, 1.0])
def test_compact_hits_is_a_stable_partition(tn, device, R, p_hit):
    import torch

    g = torch.Generator().manual_seed(R * 7 + int(p_hit * 100))
    nv = ((torch.rand(R, generator=g) < p_hit).to(torch.int32) * torch.randint(1, 300, (R,), generator=g, dtype=torch.int32)).to(device)
    order, count, padded = tn.cpp.compact_hits(nv, want_padded=True)
    hit = nv > 0
    want_hits = torch.nonzero(hit)[:, 0].to(torch.int32)
    want_miss = torch.nonzero(~hit)[:, 0].to(torch.int32)
    assert int(count[0]) == len(want_hits)
    assert torch.equal(order, torch.cat([want_hits, want_miss]))
    first = want_hits[:1] if len(want_hits) else torch.zeros(1, dtype=torch.int32, device=device)
    assert torch.equal(padded, torch.cat([want_hits, first.expand(len(want_miss))]))
    order2, count2 = tn.cpp.compact_hits(nv)
    assert torch.equal(order2, order) and torch.equal(count2, count)


def _setup(tn, scenes, render, device, n_pts=6000, seed=21):
    import torch

    pts, cells = scenes.random_mesh(n_pts, seed)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    torch.manual_seed(seed)
    mlp = render.TetraMLP().to(device)
    field = torch.randn(64, len(pts), device=device) * 0.5
    return tr, mlp, field


def _frame(scenes, device, w, h, dist=2.0):
    import torch

    o, d = scenes.pinhole_rays(w, h, eye=(0.5, 0.5 + dist, 0.5), lookat=(0.5, 0.5, 0.5), up=(0.0, 0.0, 1.0), fov_y=45.0)
    return torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)


@pytest.mark.parametrize("S,S_fine,biased,M", [(64, 0, False, 256), (100, 37, False, 256), (256, 256, False, 512),
                                               (128, 128, True, 512), (64, 64, True, 1024), (33, 20, False, 256)])
def test_one_launch_render_is_bit_identical_to_the_kernel_chain(tn, device, scenes, render, S, S_fine, biased, M):
    """tn_render_rays (everything after the trace in ONE persistent launch) against tn_sample_coarse ->
    tn_find_matched_cells_indexed -> tn_mlp_forward_gather -> tn_composite -> tn_sample_pdf -> ... on the same trace rows, for
    a frame with missing rays (tiles of several rays per wave, partial MLP groups), a batch smaller than the grid (blocks
    with one ray or none) and a batch that misses the mesh entirely."""
    import torch

    tr, mlp, field = _setup(tn, scenes, render, device)
    one = render.TetraRenderer(tr, field, mlp, S, M, fused=True, num_fine_samples=S_fine, biased=biased, fused_pass=True)
    chain = render.TetraRenderer(tr, field, mlp, S, M, fused=True, num_fine_samples=S_fine, biased=biased, fused_pass=False)
    assert one._one_launch_ok("fp32") and not chain._one_launch_ok("fp32")
    fo, fd = _frame(scenes, device, 120, 90)
    so, sd = fo[4000:4137].contiguous(), fd[4000:4137].contiguous()
    away = (fo + 10.0).contiguous()                 # origins moved away, same directions: no ray meets the mesh
    for name, (o, d) in (("frame", (fo, fd)), ("137 rays", (so, sd)), ("all miss", (away, fd))):
        a, b = one.render(o, d), chain.render(o, d)
        hits = int(a["ray_mask"].sum())
        if name == "frame":
            assert 0.2 * len(o) < hits < 0.95 * len(o), hits
        if name == "all miss":
            assert hits == 0
        assert torch.equal(a["ray_mask"], b["ray_mask"])
        for k in ("rgb", "accumulation", "depth"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), (name, k, float((a[k] - b[k]).abs().max()))
        miss = ~a["ray_mask"]
        assert bool((a["rgb"][miss] == 1).all()) and bool((a["accumulation"][miss] == 0).all())
        assert bool((a["depth"][miss] == one.far_plane).all())
        assert bool(torch.isfinite(a["rgb"]).all())
    # the appearance embedding's per-ray head bias (rows of ALL rays, indexed by ray inside the kernel)
    bias = torch.randn(len(fo), 128, device=device) * 0.7
    a, b = one.render(fo, fd, ray_head_bias=bias), chain.render(fo, fd, ray_head_bias=bias)
    assert torch.equal(a["rgb"].view(torch.int32), b["rgb"].view(torch.int32))
    assert float((a["rgb"] - one.render(fo, fd)["rgb"]).abs().max()) > 1e-2
    # a non-white background, evaluation clamp
    a, b = one.render(fo, fd, background=(0.1, 0.5, 0.9)), chain.render(fo, fd, background=(0.1, 0.5, 0.9))
    assert torch.equal(a["rgb"].view(torch.int32), b["rgb"].view(torch.int32))
    assert bool((a["rgb"][~a["ray_mask"]] == torch.tensor([0.1, 0.5, 0.9], device=device)).all())


@pytest.mark.parametrize("S,S_fine,biased,M", [(64, 0, False, 256), (100, 37, False, 256), (256, 256, False, 512), (128, 128, True, 512)])
def test_one_launch_render_bf16x3_is_bit_identical_to_the_bf16x3_chain(tn, device, scenes, render, S, S_fine, biased, M):
    """Round 6: the persistent launch also in the opt-in bf16x3 arithmetic (tn_render_rays_ex, mode 1: the MLP phases run
    x3::forward_group, the loop body of k_mlp_forward_x3; ray phase 1 leaves the ray's 32-float direction encoding and its
    appearance bias row in the tile's scratch, because that arithmetic's head layer takes the encoding as two k-steps of its GEMM)
    -- bit-identical to the bf16x3 kernel chain on a frame with missing rays, a 137-ray batch, an all-miss batch, with a per-ray
    head bias and a coloured background; and within 1e-5 of the fp32 launch (the bound the bf16x3 mode is tested to elsewhere)."""
    import torch

    tr, mlp, field = _setup(tn, scenes, render, device)
    kw = dict(fused=True, num_fine_samples=S_fine, biased=biased, mlp_mode="bf16x3")
    one = render.TetraRenderer(tr, field, mlp, S, M, fused_pass=True, **kw)
    chain = render.TetraRenderer(tr, field, mlp, S, M, fused_pass=False, **kw)
    fp32 = render.TetraRenderer(tr, field, mlp, S, M, fused=True, num_fine_samples=S_fine, biased=biased, fused_pass=True)
    assert one._one_launch_ok("bf16x3") and not chain._one_launch_ok("bf16x3")
    fo, fd = _frame(scenes, device, 120, 90)
    so, sd = fo[4000:4137].contiguous(), fd[4000:4137].contiguous()
    away = (fo + 10.0).contiguous()
    for name, (o, d) in (("frame", (fo, fd)), ("137 rays", (so, sd)), ("all miss", (away, fd))):
        a, b = one.render(o, d), chain.render(o, d)
        assert torch.equal(a["ray_mask"], b["ray_mask"])
        for k in ("rgb", "accumulation", "depth"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), (name, k, float((a[k] - b[k]).abs().max()))
        assert bool(torch.isfinite(a["rgb"]).all())
        if name == "frame":
            c = fp32.render(o, d)
            assert float((a["rgb"] - c["rgb"]).abs().max()) < 1e-5 and float((a["accumulation"] - c["accumulation"]).abs().max()) < 1e-5
            assert not torch.equal(a["rgb"].view(torch.int32), c["rgb"].view(torch.int32))      # it IS another arithmetic
    bias = torch.randn(len(fo), 128, device=device) * 0.7
    a, b = one.render(fo, fd, ray_head_bias=bias), chain.render(fo, fd, ray_head_bias=bias)
    assert torch.equal(a["rgb"].view(torch.int32), b["rgb"].view(torch.int32))
    assert float((a["rgb"] - one.render(fo, fd)["rgb"]).abs().max()) > 1e-2
    a, b = one.render(fo, fd, background=(0.1, 0.5, 0.9)), chain.render(fo, fd, background=(0.1, 0.5, 0.9))
    assert torch.equal(a["rgb"].view(torch.int32), b["rgb"].view(torch.int32))


@pytest.mark.parametrize("M", [4, 8, 16])
def test_one_launch_render_small_max_ray_triangles(tn, device, scenes, render, M):
    """max_ray_triangles of 4 / 8 (both accepted by trace_rays): 2 M < 28, so the 28 floats of the direction encoding at the start
    of a wave's LDS region reached into the coarse edges kept at offset 2 M (the matcher then read corrupted edges
    and the default render differed from the kernel chain).  The edges now sit behind max(2 M, 28) floats."""
    import torch

    pts, cells = scenes.cube_mesh()                      # 12 tets: rays cross at most a handful of faces
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    torch.manual_seed(5)
    mlp = render.TetraMLP().to(device)
    field = torch.randn(64, len(pts), device=device) * 0.5
    o, d = _frame(scenes, device, 96, 64, dist=2.5)
    for S, S_fine, biased in ((64, 0, False), (48, 32, False), (32, 32, True)):
        one = render.TetraRenderer(tr, field, mlp, S, M, fused=True, num_fine_samples=S_fine, biased=biased, fused_pass=True)
        chain = render.TetraRenderer(tr, field, mlp, S, M, fused=True, num_fine_samples=S_fine, biased=biased, fused_pass=False)
        assert one._one_launch_ok("fp32")
        a, b = one.render(o, d), chain.render(o, d)
        assert int(a["ray_mask"].sum()) > 500
        assert torch.equal(a["ray_mask"], b["ray_mask"])
        for k in ("rgb", "accumulation", "depth"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), (M, S, k, float((a[k] - b[k]).abs().max()))
        assert float(a["accumulation"].max()) > 0.05 and bool(torch.isfinite(a["rgb"]).all())


def test_one_launch_render_of_a_whole_frame_in_one_call(tn, device, scenes, render):
    """No chunking: 307,200 rays in ONE render call (every block works through dozens of tiles, the hitting-ray count is far
    beyond a 65,536-ray chunk's) -- still bit-identical to the kernel chain, for both shipped sample configurations."""
    import torch

    tr, mlp, field = _setup(tn, scenes, render, device, n_pts=6000, seed=9)
    o, d = _frame(scenes, device, 640, 480)
    for S, S_fine, biased in ((256, 256, False), (128, 128, True)):
        one = render.TetraRenderer(tr, field, mlp, S, 512, fused=True, num_fine_samples=S_fine, biased=biased, fused_pass=True)
        chain = render.TetraRenderer(tr, field, mlp, S, 512, fused=True, num_fine_samples=S_fine, biased=biased, fused_pass=False)
        a, b = one.render(o, d), chain.render(o, d)
        hits = int(a["ray_mask"].sum())
        assert hits > 100000, hits
        assert torch.equal(a["ray_mask"], b["ray_mask"])
        for k in ("rgb", "accumulation", "depth"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), (S, k, float((a[k] - b[k]).abs().max()))
        assert bool(torch.isfinite(a["rgb"]).all())
        del a, b, one, chain
        torch.cuda.empty_cache()


def test_one_launch_render_matches_the_host_compacting_form(tn, device, scenes, render):
    """... and the form that sizes its work on the host (PyTorch sampler statements + torch.nonzero, the parity definition of
    the sampler kernels): same frame at the samplers' tolerance."""
    import torch

    tr, mlp, field = _setup(tn, scenes, render, device, seed=5)
    o, d = _frame(scenes, device, 64, 64)
    for S, S_fine, biased in ((64, 64, False), (64, 64, True)):
        a = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased).render(o, d)
        b = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased,
                                 device_samplers=False).render(o, d)
        assert torch.equal(a["ray_mask"], b["ray_mask"])
        np.testing.assert_allclose(a["rgb"].cpu().numpy(), b["rgb"].cpu().numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(a["accumulation"].cpu().numpy(), b["accumulation"].cpu().numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("fused_pass", [True, False])
def test_render_does_not_synchronise_with_the_host(tn, device, scenes, render, fused_pass):
    """TetraRenderer.render under torch.cuda.set_sync_debug_mode("error"): the trace, the device-side compaction and either
    the one-launch render or the kernel chain -- nothing reads a device value on the host (rounds 2-4: torch.nonzero per
    chunk; the reference: boolean indexing, model.py:540-567)."""
    import torch

    tr, mlp, field = _setup(tn, scenes, render, device, seed=8)
    o, d = _frame(scenes, device, 80, 60)
    for S_fine, biased in ((0, False), (48, True)):
        rd = render.TetraRenderer(tr, field, mlp, 64, 256, fused=True, num_fine_samples=S_fine, biased=biased, fused_pass=fused_pass)
        want = rd.render(o, d)                      # first call: tables, weight packs, scratch
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            got = rd.render(o, d)
            got2 = rd.render(o, d, background=(0.2, 0.3, 0.4))
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        for k in ("rgb", "accumulation", "depth", "ray_mask"):
            assert torch.equal(got[k], want[k]), k
        assert 0 < int(got["ray_mask"].sum()) < len(o)
        assert not torch.equal(got2["rgb"], got["rgb"])


def test_count_aware_chain_kernels_leave_rows_beyond_count_alone(tn, device, scenes, render):
    """Every kernel of the chain takes the device-side ray count: rows from `count` on must stay untouched (sentinel-filled
    outputs), rows below it must equal the call that was sized on the host."""
    import torch

    cpp = tn.cpp
    tr, mlp, field = _setup(tn, scenes, render, device, seed=3)
    o, d = _frame(scenes, device, 50, 40)
    M, S = 256, 64
    out = tr.trace_rays(o, d, M, compact_rows=True)
    nv = out["num_visited_cells"]
    lists = [out[k] for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates", "hit_distances", "vertex_indices")]
    order, count = cpp.compact_hits(nv)
    r = int(count[0])
    assert 0 < r < len(o)
    ridx = order[:r].contiguous()
    # host-sized calls
    e_h, nf_h = cpp.sample_coarse(lists[0], lists[3], ridx, S)
    # device-sized calls over ALL rows; poison first (the wrappers allocate with torch.empty)
    import os
    os.environ.pop("TETRANERF_HIP_POISON", None)
    e_d, nf_d = cpp.sample_coarse(lists[0], lists[3], order, S, count=count)
    assert torch.equal(e_d[:r], e_h) and torch.equal(nf_d[:r], nf_h)
    dist_h = ((e_h[:, 1:] + e_h[:, :-1]) / 2).contiguous()
    dist_d = torch.zeros(len(o), S, device=device)
    dist_d[:r] = dist_h
    t_h = tr.find_visited_cells(*lists, dist_h, ray_index=ridx)
    t_d = tr.find_visited_cells(*lists, dist_d, ray_index=order, count=count)
    for k in t_h:
        assert torch.equal(t_d[k][:r], t_h[k]), k
    w = render.mlp_weights(mlp)
    dirs_o = d.index_select(0, order.long())
    s_h, c_h = cpp.mlp_forward_gather(t_h["vertex_indices"], t_h["barycentric_coordinates"], field, dirs_o[:r].contiguous(), w, S)
    s_d, c_d = cpp.mlp_forward_gather(t_d["vertex_indices"], t_d["barycentric_coordinates"], field, dirs_o, w, S, count=count)
    assert torch.equal(s_d[: r * S], s_h) and torch.equal(c_d[: r * S], c_h)
    rgb = torch.full((len(o), 3), 7.0, device=device); acc = torch.full((len(o), 1), 7.0, device=device); dep = torch.full((len(o), 1), 7.0, device=device)
    cpp.composite(s_d.view(-1, S), c_d.view(-1, S, 3), e_d, out=(rgb, acc, dep), ray_index=order, count=count)
    w_rgb, w_acc, w_dep = cpp.composite(s_h.view(-1, S), c_h.view(-1, S, 3), e_h)
    hit = nv > 0
    assert torch.equal(rgb[ridx.long()], w_rgb) and torch.equal(acc[ridx.long()], w_acc) and torch.equal(dep[ridx.long()], w_dep)
    assert bool((rgb[~hit] == 7.0).all()) and bool((acc[~hit] == 7.0).all()) and bool((dep[~hit] == 7.0).all())
