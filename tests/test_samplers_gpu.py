"""The sampler kernels (tn_sample_coarse / tn_sample_pdf, csrc/tn_samplers.hip) against the PyTorch statements of the same
samplers in render.py (uniform_sample_bins, biased_sample_bins -- pinned bit-for-bit by the reference's own function in
tests/test_samplers.py -- and pdf_sample_bins): same expressions element by element; the only differences are the
rounding of one prefix sum per ray (segment lengths / pdf), so the bin edges agree to a few ulps of the ray length."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(tn, device, scenes, M=256):
    import torch

    pts, cells = scenes.random_mesh(5000, 41)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    o, d = scenes.outside_in_rays(6000, 42)
    o2, d2 = scenes.pinhole_rays(40, 25, eye=(0.5, 3.0, 2.6), lookat=(0.5, 0.5, 0.5))        # many misses
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    out = tr.trace_rays(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device), M)
    nv = out["num_visited_cells"]
    idx = torch.nonzero(nv > 0)[:, 0]
    assert 1000 < len(idx) < len(o)
    near = out["hit_distances"][idx, 0, 0][:, None]
    far = out["hit_distances"][idx, nv[idx].long() - 1, 1][:, None]
    return out, nv, idx, near, far


@pytest.mark.parametrize("biased", [False, True])
@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("S", [48, 128, 256])
def test_sample_coarse_matches_torch(tn, device, scenes, S, train, biased):
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    out, nv, idx, near, far = _setup(tn, device, scenes)
    torch.manual_seed(S)
    t_rand = torch.rand(len(idx), S + 1, device=device) if train else None
    edges, nf = tn.cpp.sample_coarse(nv, out["hit_distances"], idx.to(torch.int32), S, biased=biased, t_rand=t_rand)
    assert torch.equal(nf[:, 0:1], near) and torch.equal(nf[:, 1:2], far)
    if biased:
        want = render.biased_sample_bins(near, far, S, nv[idx], out["hit_distances"][idx], t_rand)
    else:
        want = render.uniform_sample_bins(near, far, S, t_rand)
    if not biased:
        assert torch.equal(edges, want)                       # same expression tree, no reduction involved
    scale = (far - near).abs() + far.abs()
    assert float(((edges - want).abs() / scale).max()) < 4e-7, float(((edges - want).abs() / scale).max())
    assert bool((edges[:, 1:] >= edges[:, :-1] - 1e-6 * far).all())


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("S,S_fine", [(64, 64), (128, 128), (256, 256)])
def test_sample_pdf_matches_torch(tn, device, scenes, S, S_fine, train):
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    out, nv, idx, near, far = _setup(tn, device, scenes)
    torch.manual_seed(S + 7)
    r = len(idx)
    edges, nf = tn.cpp.sample_coarse(nv, out["hit_distances"], idx.to(torch.int32), S, biased=True)
    # peaky weights (a surface), flat weights, all-zero weights (padding path), a few exactly-equal cdf steps
    w = torch.rand(r, S, device=device) ** 8
    w[::5] = 0.0
    w[1::5] = 0.3 / S
    w = (w / w.sum(-1, keepdim=True).clamp_min(1.0)).contiguous()
    u_rand = torch.rand(r, S_fine + 1, device=device) if train else None
    got = tn.cpp.sample_pdf(edges, w, nf, S_fine, u_rand=u_rand)
    spacing = (edges - near) / (far - near)
    want = render.pdf_sample_bins(spacing, w, S_fine, near, far, u_rand=u_rand)
    assert got.shape == want.shape == (r, S + S_fine + 2) and not bool(torch.isnan(got).any())
    # (sorted in spacing; the euclidean map bins * far + (1 - bins) * near may invert neighbours by an ulp, in torch too)
    assert bool((got[:, 1:] >= got[:, :-1] - 1e-6 * far).all())
    err = ((got - want).abs() / (far - near)).max(dim=1).values
    # the inverse CDF amplifies the rounding of the cdf by at most (bin width) * wsum / histogram_padding ~ 1.4
    assert float(err.max()) < 2e-5, float(err.max())
    assert float(err.median()) < 2e-6


def test_renderer_with_device_samplers_equals_torch_samplers(tn, device, scenes):
    """Whole render with the sampler kernels vs with the PyTorch sampler statements (same kernels otherwise)."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    pts, cells = scenes.random_mesh(5000, 41)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    torch.manual_seed(0)
    mlp = render.TetraMLP().to(device)
    field = (torch.rand(64, len(pts), device=device) * 2 - 1)
    o, d = scenes.outside_in_rays(3000, 43)
    o, d = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    for S, S_fine, biased in ((256, 256, False), (128, 128, True), (96, 0, True)):
        outs = [render.TetraRenderer(tr, field, mlp, S, 256, num_fine_samples=S_fine, biased=biased, device_samplers=ds).render(o, d)
                for ds in (True, False)]
        assert torch.equal(outs[0]["ray_mask"], outs[1]["ray_mask"])
        np.testing.assert_allclose(outs[0]["rgb"].cpu().numpy(), outs[1]["rgb"].cpu().numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(outs[0]["accumulation"].cpu().numpy(), outs[1]["accumulation"].cpu().numpy(), rtol=0, atol=1e-5)
