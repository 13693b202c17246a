"""The nerfstudio adapter on the GPU (row b2): a stand-in TetrahedraNerf (tests/golden/nerfstudio_standins.py: the
reference's attribute names, nerfstudio-shaped modules) whose get_outputs is swapped by nerfstudio_plugin.install --
evaluation against the reference-shaped path (HIP ops + the model's own torch modules) at 1e-5, training gradients
reaching tetrahedra_field and all 12 weight tensors and matching the unfused autograd statement."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests" / "golden"))


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _setup(scenes, device, **cfg):
    import torch

    standins = importlib.import_module("nerfstudio_standins")
    plugin = importlib.import_module("tetra-nerf_amd.nerfstudio_plugin")
    pts, cells = scenes.random_mesh(6000, 31)
    torch.manual_seed(0)
    Fused = plugin.make_fused_model_class(standins.StandInTetrahedraNerf)
    model = Fused(standins.Config(**cfg), torch.from_numpy(pts), torch.from_numpy(cells)).to(device)
    with torch.no_grad():
        model.tetrahedra_field[0] = torch.rand(len(pts), device=device) * 6 - 3     # a density-driving row
    return standins, plugin, model


@pytest.mark.parametrize("cfg", [dict(num_samples=256, num_fine_samples=256),
                                 dict(num_samples=128, num_fine_samples=128, use_biased_sampler=True, use_gradient_scaling=True),
                                 dict(num_samples=96, num_fine_samples=0, background_color="black", max_intersected_triangles=256)])
def test_adapter_eval_matches_reference_path(tn, device, scenes, cfg):
    import torch

    standins, plugin, model = _setup(scenes, device, **cfg)
    model.eval()
    o, d = scenes.outside_in_rays(3600, 32)
    o2, d2 = scenes.pinhole_rays(31, 16, eye=(0.5, 3.0, 2.6), lookat=(0.5, 0.5, 0.5))     # + rays that partly miss the mesh
    o, d = np.ascontiguousarray(np.concatenate([o, o2])), np.ascontiguousarray(np.concatenate([d, d2]))
    rb = standins.RayBundle(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device))
    with torch.no_grad():
        got = model.get_outputs(rb)                                          # fused kernels
        want = standins.StandInTetrahedraNerf.get_outputs(model, rb)         # reference-shaped path
    assert model.reference_calls == 1
    assert torch.equal(got["ray_mask"], want["ray_mask"]) and float(got["ray_mask"].float().mean()) > 0.3
    miss = ~got["ray_mask"]
    bg = 0.0 if cfg.get("background_color") == "black" else 1.0
    assert bool((got["rgb"][miss] == bg).all()) and bool((got["depth"][miss] == 1000.0).all())
    np.testing.assert_allclose(got["rgb"].cpu().numpy(), want["rgb"].cpu().numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(got["accumulation"].cpu().numpy(), want["accumulation"].cpu().numpy(), rtol=0, atol=1e-5)
    n_final = cfg["num_samples"] + cfg["num_fine_samples"] + 1 if cfg["num_fine_samples"] else cfg["num_samples"]
    decided = ((want["depth_margin"] > 4e-6 * n_final)[:, 0] & want["ray_mask"]).cpu().numpy()     # see tests/test_render_gpu.py::test_render_c3
    assert decided.sum() > 200
    np.testing.assert_allclose(got["depth"].cpu().numpy()[decided], want["depth"].cpu().numpy()[decided], rtol=0, atol=1e-5)
    # a second call reuses the renderer and its packed weights; an optimiser-style in-place update is seen
    rd = model._tn_renderer
    with torch.no_grad():
        for p in plugin.weights_from_model(model):
            p.mul_(1.01)
        model.tetrahedra_field.mul_(0.99)
        got2 = model.get_outputs(rb)
        want2 = standins.StandInTetrahedraNerf.get_outputs(model, rb)
    assert model._tn_renderer is rd
    np.testing.assert_allclose(got2["rgb"].cpu().numpy(), want2["rgb"].cpu().numpy(), rtol=0, atol=1e-5)
    assert float((got2["rgb"] - got["rgb"]).abs().max()) > 1e-4


@pytest.mark.parametrize("cfg", [dict(num_samples=64, num_fine_samples=64),
                                 dict(num_samples=48, num_fine_samples=48, use_biased_sampler=True, use_gradient_scaling=True)])
def test_adapter_training_gradients(tn, device, scenes, cfg):
    import torch

    standins, plugin, model = _setup(scenes, device, **cfg)
    model.train()
    o, d = scenes.outside_in_rays(2048, 33)
    rb = standins.RayBundle(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device))
    target = torch.rand(len(o), 3, device=device)
    params = [model.tetrahedra_field] + plugin.weights_from_model(model)

    def step(fn):
        for p in params:
            p.grad = None
        torch.manual_seed(7)                    # the same stratified draws in both runs
        out = fn()
        loss = ((out["rgb"] - target) ** 2).mean() + 0.1 * out["accumulation"].mean()
        loss.backward()
        return out["rgb"].detach().clone(), [p.grad.clone() for p in params]

    rgb_f, g_f = step(lambda: model.get_outputs(rb))          # adapter -> render_train (fused autograd nodes)
    rd = model._tn_renderer
    rgb_u, g_u = step(lambda: rd.render_train(rb.origins, rb.directions, gradient_scaling=cfg.get("use_gradient_scaling", False), fused=False))
    np.testing.assert_allclose(rgb_f.cpu().numpy(), rgb_u.cpu().numpy(), rtol=0, atol=1e-5)
    names = ["field", "w1", "b1", "w2", "b2", "w3", "b3", "wd", "bd", "wh", "bh", "wr", "br"]
    for name, a, b in zip(names, g_f, g_u):
        assert a is not None and float(a.abs().max()) > 0, name
        # two fp32 evaluations of the same gradient with differently split sums (4096-sample slices + float atomics vs
        # cuBLAS-style reductions); the float64 yardstick for the fused nodes is tests/test_train_gpu.py
        cos = float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()))
        assert _rel(a, b) < 5e-3 and cos > 0.99999, (name, _rel(a, b), cos)
    # evaluation-mode call under no_grad goes through render()
    model.eval()
    with torch.no_grad():
        out = model.get_outputs(rb)
    assert not out["rgb"].requires_grad
