"""The reference's OWN nerfstudio model file as the oracle of the glue above the native ops (CPU part).

`tetranerf/nerfstudio/model.py` is staged verbatim (tests/golden/ref/) and imported on a stub nerfstudio namespace
(tests/golden/nerfstudio_stub/, see tests/golden/reference_model.py).  Executed from that file here:
`TetrahedraSampler.generate_ray_samples` (model.py:141-192), `GradientScaler` (:195-205), `get_background_color`
(:504-518) and the whole `get_outputs` body (:520-662, evaluation and training mode, with the CPU oracle as tracer
and the reference's einsum definition of the gather) -- against the PyTorch statements of tetra-nerf_amd/render.py
(`biased_sample_bins`, `GradientScaler`, `render_reference`, `TetraRenderer.render_train(fused=False)`), which are what
the `-m gpu` tests compare the kernels with.  The GPU part (the same reference body on the HIP ops vs the fused
adapter) is tests/test_reference_model_gpu.py."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests" / "golden"))
import reference_model as rm   # noqa: E402

render = importlib.import_module("tetra-nerf_amd.render")
plugin = importlib.import_module("tetra-nerf_amd.nerfstudio_plugin")


@pytest.fixture(scope="module")
def ref():
    return rm.load()


def _bottle_hits(oracle, bottle, scenes, n=48):
    t = rm.OracleTorchTracer(oracle, bottle["vertices"], bottle["cells"])
    o, d = scenes.pinhole_rays(n, n)
    out = t.trace_rays(torch.from_numpy(o), torch.from_numpy(d), 256)
    hit = out["num_visited_cells"] > 0
    return torch.from_numpy(o)[hit], torch.from_numpy(d)[hit], out["num_visited_cells"][hit], out["hit_distances"][hit]


def test_staged_model_is_the_reference_file():
    """On this container the staged copy is compared with /root/reference byte for byte (absent on the GPU box)."""
    src = Path("/root/reference/tetranerf/nerfstudio/model.py")
    if not src.exists():
        pytest.skip("/root/reference is not present")
    assert (ROOT / "tests/golden/ref/tetranerf/nerfstudio/model.py").read_bytes() == src.read_bytes()


def test_state_dict_names_and_fallback_rule(ref, scenes):
    pts, cells = scenes.random_mesh(200, 3)
    model = rm.build_model(ref, pts, cells)
    sd = model.state_dict()
    for k in plugin.STATE_DICT_KEYS + (plugin.FIELD_KEY,):
        assert k in sd, k
    ws, field = plugin.weights_from_state_dict(sd)
    for a, b in zip(ws, plugin.weights_from_model(model)):
        assert a.data_ptr() == b.data_ptr()
    assert plugin.fused_config_supported(model.config) == (True, "")
    assert plugin.fused_config_supported(rm.build_model(ref, pts, cells, appearance_embed_dim=8).config) == (True, "")
    for bad in (dict(input_fourier_frequencies=2), dict(hidden_size=64),
                dict(background_color="random"), dict(num_density_layers=2), dict(num_color_layers=2)):
        m2 = rm.build_model(ref, pts, cells, **bad)
        ok, why = plugin.fused_config_supported(m2.config)
        assert not ok and why


@pytest.mark.parametrize("train", [False, True])
def test_tetrahedra_sampler_from_the_file(ref, oracle, bottle, scenes, train):
    """TetrahedraSampler.generate_ray_samples executed from the reference file == render.biased_sample_bins, bit for
    bit, in evaluation mode and with the same stratified draws in training mode; spacing bins / deltas as the reference
    hands them to the PDF sampler and the GradientScaler."""
    o, d, nv, hd = _bottle_hits(oracle, bottle, scenes)
    near = hd[:, 0, 0][:, None]
    far = torch.gather(hd[:, :, 1], 1, (nv[:, None].long() - 1).clamp_min(0))
    rb = rm.ray_bundle(ref, o, d)
    rb.nears, rb.fars = near, far
    S = 40
    sampler = ref.TetrahedraSampler(num_samples=S)
    sampler.train(train)
    torch.manual_seed(5)
    rs = sampler(rb, num_visited_cells=nv, hit_distances=hd.clone())
    got = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1)
    torch.manual_seed(5)
    t_rand = torch.rand((len(o), S + 1)) if train else None
    want = render.biased_sample_bins(near, far, S, nv, hd, t_rand)
    np.testing.assert_array_equal(got.numpy(), want.numpy())
    spacing = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1)
    np.testing.assert_array_equal(spacing.numpy(), ((want - near) / (far - near)).numpy())
    assert tuple(rs.frustums.directions.shape) == (len(o), S, 3)


def test_gradient_scaler_from_the_file(ref):
    torch.manual_seed(0)
    col, sig = torch.rand(7, 9, 3, requires_grad=True), torch.rand(7, 9, 1, requires_grad=True)
    dist = torch.rand(7, 9, 1) * 2
    outs = []
    for fn in (ref.GradientScaler, render.GradientScaler):
        c, s, r = fn.apply(col, sig, dist)
        g = torch.autograd.grad((c * torch.arange(3.0)).sum() + (s * 2).sum(), (col, sig))
        outs.append((c, s, r, *g))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_background_colour_from_the_file(ref, scenes):
    """get_background_color (model.py:504-518) of the reference + what the adapter resolves for its kernels."""
    from nerfstudio.model_components import renderers

    pts, cells = scenes.random_mesh(100, 3)
    for name, val in (("white", 1.0), ("black", 0.0)):
        model = rm.build_model(ref, pts, cells, background_color=name)
        bg = model.get_background_color((5, 3), torch.device("cpu"))
        assert tuple(bg.shape) == (5, 3) and bool((bg == val).all())
        assert plugin.resolve_background(model) == val
        with renderers.background_color_override_context(torch.tensor([0.25, 0.5, 0.75])):
            bg = model.get_background_color((5, 3), torch.device("cpu"))
            assert torch.equal(bg[2], torch.tensor([0.25, 0.5, 0.75]))
            assert plugin.resolve_background(model) == (0.25, 0.5, 0.75)
        with renderers.background_color_override_context(torch.tensor([0.5, 0.5, 0.5])):
            assert plugin.resolve_background(model) == 0.5
        assert plugin.resolve_background(model) == val
    model = rm.build_model(ref, pts, cells, background_color="random")
    assert plugin.resolve_background(model) is None
    with renderers.background_color_override_context(torch.tensor([0.0, 0.0, 1.0])):
        assert plugin.resolve_background(model) == (0.0, 0.0, 1.0)


CONFIGS = [dict(num_samples=48, num_fine_samples=0, max_intersected_triangles=256),
           dict(num_samples=32, num_fine_samples=32, max_intersected_triangles=256),
           dict(num_samples=24, num_fine_samples=24, use_biased_sampler=True, use_gradient_scaling=True,
                max_intersected_triangles=256, background_color="black")]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_reference_get_outputs_eval_equals_render_reference(ref, oracle, bottle, scenes, monkeypatch, cfg):
    """TetrahedraNerf.get_outputs (the reference's body, evaluation mode) == render.render_reference on the CPU: the
    statement every `-m gpu` render test compares the kernels with is pinned by the reference's own code."""
    model = rm.build_model(ref, bottle["vertices"], bottle["cells"], density_boost=12.0, **cfg).eval()
    tracer = rm.cpu_model(ref, model, oracle, monkeypatch)
    o, d = scenes.pinhole_rays(56, 56)
    rb = rm.ray_bundle(ref, o, d)
    with torch.no_grad():
        want = model.get_outputs(rb)
        got = render.render_reference(tracer, rm.einsum_interpolate_values, model.tetrahedra_field, plugin.ModelMLP(model),
                                      rb.origins, rb.directions, cfg["num_samples"], cfg["max_intersected_triangles"],
                                      model.collider.far_plane, cfg["num_fine_samples"], cfg.get("use_biased_sampler", False),
                                      background={"white": 1.0, "black": 0.0}[cfg.get("background_color", "white")])
    assert torch.equal(got["ray_mask"], want["ray_mask"]) and 100 < int(want["ray_mask"].sum()) < len(o)
    for k in ("rgb", "accumulation", "depth"):
        assert got[k].shape == want[k].shape
        np.testing.assert_allclose(got[k].numpy(), want[k].numpy(), rtol=0, atol=1e-6, err_msg=k)
    assert float(want["accumulation"].max()) > 0.5


@pytest.mark.parametrize("cfg", CONFIGS)
def test_reference_get_outputs_train_equals_render_train_statement(ref, oracle, bottle, scenes, monkeypatch, cfg):
    """Training mode: the reference's body (stratified samplers drawing from torch's global generator, PDF sampler on
    detached weights, GradientScaler, renderers without the evaluation clamp) and `TetraRenderer.render_train(fused=False,
    device_samplers=False)` see the same draws under the same seed, so outputs AND gradients of the field and the 12
    weight tensors must agree to fp32 round-off."""
    model = rm.build_model(ref, bottle["vertices"], bottle["cells"], density_boost=12.0, **cfg).train()
    tracer = rm.cpu_model(ref, model, oracle, monkeypatch)
    o, d = scenes.pinhole_rays(56, 56)      # (a 40x40 frame holds a ray with near == far: the reference's own biased mapping indexes with NaN there)
    rb = rm.ray_bundle(ref, o, d)
    params = [model.tetrahedra_field] + plugin.weights_from_model(model)
    target = torch.rand(len(o), 3)
    rd = render.TetraRenderer(tracer, model.tetrahedra_field, plugin.ModelMLP(model), num_samples=cfg["num_samples"],
                              max_ray_triangles=cfg["max_intersected_triangles"], far_plane=model.collider.far_plane,
                              num_fine_samples=cfg["num_fine_samples"], biased=cfg.get("use_biased_sampler", False),
                              background={"white": 1.0, "black": 0.0}[cfg.get("background_color", "white")],
                              cache_field=False, device_samplers=False, interpolate_values=rm.einsum_interpolate_values)

    def step(fn):
        for p in params:
            p.grad = None
        torch.manual_seed(11)
        out = fn()
        loss = ((out["rgb"] - target) ** 2).mean() + 0.1 * out["accumulation"].mean()
        loss.backward()
        return {k: out[k].detach().clone() for k in ("rgb", "accumulation", "depth")}, [p.grad.clone() for p in params]

    want, g_want = step(lambda: model.get_outputs(rb))
    got, g_got = step(lambda: rd.render_train(rb.origins, rb.directions, gradient_scaling=cfg.get("use_gradient_scaling", False),
                                              fused=False))
    for k in ("rgb", "accumulation", "depth"):
        np.testing.assert_allclose(got[k].numpy(), want[k].numpy(), rtol=0, atol=1e-6, err_msg=k)
    for i, (a, b) in enumerate(zip(g_got, g_want)):
        assert float(b.abs().max()) > 0, i
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=2e-6 * float(b.abs().max()), err_msg=f"grad {i}")
