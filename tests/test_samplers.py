"""Sample placement of the render path (PyTorch plumbing around the HIP ops, render.py).

biased_sample_bins is pinned by outputs of the REFERENCE's own function
(tetranerf/nerfstudio/model.py:111-122, executed by tests/golden/make_golden.py) on the bottle's
hit lists; pdf_sample_bins restates nerfstudio's PDFSampler (absent here: "parity unpinned", only
its defining properties are checked)."""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch

render = importlib.import_module("tetra-nerf_amd.render")
GOLD = Path(__file__).resolve().parent / "golden"


def test_biased_sampler_matches_reference_function():
    g = np.load(GOLD / "biased_sampler.npz")
    nv, hd = torch.from_numpy(g["num_visited_cells"]), torch.from_numpy(g["hit_distances"])
    near = hd[:, 0, 0][:, None]
    far = torch.gather(hd[:, :, 1], 1, (nv[:, None].long() - 1).clamp_min(0))
    S = g["samples"].shape[1] - 1
    np.testing.assert_array_equal(render.uniform_sample_bins(near, far, S).numpy(), g["samples"])
    got = render.map_to_biased(nv, hd, torch.from_numpy(g["samples"]))
    np.testing.assert_array_equal(got.numpy(), g["mapped"])
    np.testing.assert_array_equal(render.biased_sample_bins(near, far, S, nv, hd).numpy(), g["mapped"])
    # the inputs are not modified (the reference clamps a temporary in place)
    np.testing.assert_array_equal(hd.numpy(), g["hit_distances"])


def test_biased_sampler_equal_share_per_cell():
    # three contiguous cells of very different lengths: each receives a third of the edges
    hd = torch.tensor([[[1.0, 1.1], [1.1, 3.0], [3.0, 3.2], [0.0, 0.0]]])
    nv = torch.tensor([3], dtype=torch.int32)
    e = render.biased_sample_bins(torch.tensor([[1.0]]), torch.tensor([[3.2]]), 30, nv, hd)[0]
    assert torch.all(e[1:] >= e[:-1])
    assert int(((e >= 1.0) & (e < 1.1 - 1e-6)).sum()) == 10
    assert int(((e >= 1.1 - 1e-6) & (e < 3.0 - 1e-6)).sum()) == 10
    assert abs(float(e[0]) - 1.0) < 1e-6 and abs(float(e[-1]) - 3.2) < 1e-5


def test_pdf_sampler_properties():
    torch.manual_seed(0)
    R, S, Nf = 9, 64, 32
    near, far = torch.rand(R, 1) + 1, torch.rand(R, 1) + 3
    spacing = torch.linspace(0, 1, S + 1)[None].expand(R, -1).contiguous()
    w = torch.zeros(R, S)
    w[:, 20:24] = 0.25          # all the mass in four bins
    w[0] = 0                    # zero-weight ray: the padding alone gives a uniform pdf
    e = render.pdf_sample_bins(spacing, w, Nf, near, far)
    assert tuple(e.shape) == (R, S + Nf + 2)
    assert torch.all(e[:, 1:] >= e[:, :-1])
    assert torch.allclose(e[:, :1], near) and torch.allclose(e[:, -1:], far)
    # include_original: every coarse edge is still there
    coarse = spacing * far + (1 - spacing) * near
    for r in range(R):
        assert np.isin(coarse[r].numpy(), e[r].numpy()).all()
    # most of the new edges fall into the heavy bins: mass 1 / (1 + 64*0.01) of 33 quantiles
    lo, hi = coarse[1, 20], coarse[1, 24]
    inside = int(((e[1] > lo) & (e[1] < hi)).sum()) - 3
    assert 18 <= inside <= 22
    # uniform pdf: the quantiles are the bin-centred grid
    u = (torch.arange(Nf + 1) + 0.5) / (Nf + 1)
    want = (u * far[0] + (1 - u) * near[0]).numpy()
    assert np.abs(e[0].numpy()[None, :] - want[:, None]).min(1).max() < 1e-5


def test_ray_weights_matches_composite():
    torch.manual_seed(1)
    sigma = torch.rand(5, 40) * 10
    edges = 1 + torch.cumsum(torch.rand(5, 41) * 0.05, -1)
    w = render.ray_weights(sigma, edges)
    _, _, _, w2 = render.composite(sigma[..., None], torch.zeros(5, 40, 3), edges[:, :-1, None], edges[:, 1:, None])
    np.testing.assert_array_equal(w.numpy(), w2[..., 0].numpy())


def test_th_file_roundtrip(tmp_path):
    io = importlib.import_module("tetra-nerf_amd.tetrahedra_io")
    scenes = importlib.import_module("tetra-nerf_amd.scenes")
    pts, cells = scenes.random_mesh(200, 3)
    colors = torch.randint(0, 256, (len(pts), 4), dtype=torch.uint8)
    io.save_tetrahedra(tmp_path / "sub" / "t.th", torch.from_numpy(pts), torch.from_numpy(cells), colors)
    raw = torch.load(str(tmp_path / "sub" / "t.th"))
    assert set(raw) == {"cells", "vertices", "colors"} and raw["cells"].dtype == torch.int32
    T = torch.tensor([[0.0, 1, 0, 0.5], [1, 0, 0, -1], [0, 0, 2, 0]])
    got = io.load_tetrahedra(tmp_path / "sub" / "t.th", T, 0.25)
    want = (np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1) @ T.numpy().T) * 0.25
    np.testing.assert_allclose(got["vertices"].numpy(), want, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(got["cells"].numpy(), cells)
    field = torch.zeros(64, len(pts))
    io.init_field_from_colors(field, got["colors"])
    np.testing.assert_allclose(field[1:4].T.numpy(), colors[:, :3].float().numpy() * 2 / 255 - 1, atol=1e-6)
    np.testing.assert_allclose(field[0].numpy(), colors[:, 3].float().numpy() * 2 / 255 - 1, atol=1e-6)
    with pytest.raises(RuntimeError, match="does not exist"):
        io.load_tetrahedra(tmp_path / "missing.th")


@pytest.mark.parametrize("cfg", [(64, 0, False), (32, 32, False), (32, 32, True)])
def test_render_reference_on_cpu(oracle, scenes, bottle, cfg):
    """The plain-PyTorch statement of the model's evaluation forward (render.render_reference) runs end to end on
    the CPU with the oracle as tracer: the reference's bottle mesh + ray generator, all three sampling schemes."""
    S, S_fine, biased = cfg
    pts, cells = bottle["vertices"], bottle["cells"]
    o, d = scenes.pinhole_rays(64, 64)

    class CpuTracer:
        def __init__(self):
            self.t = oracle.OracleTracer(use_bvh=True)
            self.t.load_tetrahedra(pts, cells)

        def trace_rays(self, o_, d_, M_):
            return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in self.t.trace_rays(o_.numpy(), d_.numpy(), M_).items()}

        def find_visited_cells(self, *a):
            return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in self.t.find_visited_cells(*[x.numpy() for x in a]).items()}

    def interp(vi, bc, f):
        return torch.from_numpy(np.ascontiguousarray(oracle.interpolate_values(vi.numpy(), bc.numpy(), f.numpy())))

    torch.manual_seed(0)
    mlp = render.TetraMLP()
    field = torch.randn(64, len(pts)) * 0.5
    with torch.no_grad():
        out = render.render_reference(CpuTracer(), interp, field, mlp, torch.from_numpy(o), torch.from_numpy(d), S, 256,
                                      num_fine_samples=S_fine, biased=biased)
    hit = out["ray_mask"]
    assert int(hit.sum()) == 146                      # rays of the reference's generator that hit the bottle
    assert torch.all(out["rgb"][~hit] == 1.0) and torch.all(out["accumulation"][~hit] == 0.0)
    assert torch.all(out["depth"][~hit] == 1000.0)
    acc = out["accumulation"][hit]
    assert torch.all((acc > 0) & (acc <= 1.0 + 1e-5))
    assert torch.all((out["rgb"] >= -1e-5) & (out["rgb"] <= 1.0 + 1e-5))
    assert torch.all((out["depth"][hit] > 0.8) & (out["depth"][hit] < 1.3))   # the bottle sits ~1 from the eye
