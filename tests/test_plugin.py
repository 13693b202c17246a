"""tetra-nerf_amd/nerfstudio_plugin.py on the CPU: the checkpoint-name -> kernel-weight mapping, the fallback rule and
the (un)installation, with duck-typed stand-ins of the nerfstudio modules (tests/golden/nerfstudio_standins.py)."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests" / "golden"))


@pytest.fixture(scope="module")
def plugin():
    return importlib.import_module("tetra-nerf_amd.nerfstudio_plugin")


@pytest.fixture(scope="module")
def standins():
    return importlib.import_module("nerfstudio_standins")


def _model(standins, **cfg):
    import torch

    torch.manual_seed(0)
    verts = torch.rand(50, 3)
    cells = torch.randint(0, 50, (80, 4), dtype=torch.int32)
    return standins.StandInTetrahedraNerf(standins.Config(**cfg), verts, cells)


def test_state_dict_mapping_reproduces_the_model(plugin, standins):
    """The 12 tensors taken from the reference checkpoint keys (model.py:436-456 module names), loaded into the plain
    TetraMLP statement, compute what the model's own modules compute."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    model = _model(standins)
    sd = {"_model." + k: v for k, v in model.state_dict().items()}
    ws, field = plugin.weights_from_state_dict(sd, prefix="_model.")
    assert field.shape == (64, 50) and torch.equal(field, model.tetrahedra_field)
    for a, b in zip(ws, plugin.weights_from_model(model)):
        assert torch.equal(a, b)
    ref = render.TetraMLP()
    with torch.no_grad():
        for dst, src in zip(render.mlp_weights(ref), ws):
            dst.copy_(src)
        feats, dirs = torch.randn(7, 11, 64), torch.nn.functional.normalize(torch.randn(7, 11, 3), dim=-1)
        s1, c1 = ref(feats, dirs)
        s2, c2 = plugin.ModelMLP(model)(feats, dirs)
    np.testing.assert_allclose(s1.numpy(), s2.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(c1.numpy(), c2.numpy(), rtol=0, atol=1e-6)
    # mlp_weights of the adapter = the model's live parameters (gradients reach them)
    for a, b in zip(render.mlp_weights(plugin.ModelMLP(model)), plugin.weights_from_model(model)):
        assert a is b


def test_wrong_architecture_is_rejected(plugin, standins):
    model = _model(standins, hidden_size=64)
    with pytest.raises(RuntimeError, match="shipped architecture"):
        plugin.weights_from_state_dict(model.state_dict())


def test_load_reference_checkpoint_bumps_versions(plugin, standins):
    import torch

    src, dst = _model(standins), _model(standins)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(1.0)
    before = [p._version for p in plugin.weights_from_model(dst)] + [dst.tetrahedra_field._version]
    plugin.load_reference_checkpoint(dst, src.state_dict())
    after = [p._version for p in plugin.weights_from_model(dst)] + [dst.tetrahedra_field._version]
    assert all(b > a for a, b in zip(before, after))     # what the weight / field caches of the extension watch
    for a, b in zip(plugin.weights_from_model(dst), plugin.weights_from_model(src)):
        assert torch.equal(a, b)


def test_fallback_rule(plugin, standins):
    ok = lambda **kw: plugin.fused_config_supported(standins.Config(**kw))[0]   # noqa: E731
    assert ok() and ok(num_samples=128, num_fine_samples=128, use_biased_sampler=True, use_gradient_scaling=True)
    assert ok(background_color="black") and ok(num_fine_samples=0) and ok(max_intersected_triangles=1024)
    assert ok(appearance_embed_dim=8)            # round 4: the embedding acts through the kernels' per-ray head bias
    for bad in (dict(input_fourier_frequencies=4), dict(hidden_size=256), dict(num_density_layers=2),
                dict(num_color_layers=2), dict(field_dim=32), dict(background_color="random"), dict(background_color="last_sample")):
        supported, why = plugin.fused_config_supported(standins.Config(**bad))
        assert not supported and why, bad


def test_install_routes_unsupported_configs_to_the_reference(plugin, standins):
    """install() swaps get_outputs; a configuration outside the fused kernels' architecture runs the reference body."""
    calls = []

    class Model(standins.StandInTetrahedraNerf):
        def get_outputs(self, ray_bundle):
            calls.append(ray_bundle)
            return {"rgb": "reference"}

    assert plugin.install(Model) is Model and plugin.install(Model) is Model       # idempotent
    assert Model.get_outputs is plugin.fused_get_outputs
    import torch

    torch.manual_seed(0)
    m = Model(standins.Config(input_fourier_frequencies=2), torch.rand(20, 3), torch.randint(0, 20, (30, 4), dtype=torch.int32))
    assert m.get_outputs("bundle") == {"rgb": "reference"} and calls == ["bundle"]
    plugin.uninstall(Model)
    assert Model.get_outputs is not plugin.fused_get_outputs
    Fused = plugin.make_fused_model_class(standins.StandInTetrahedraNerf)
    assert Fused.__name__ == "FusedStandInTetrahedraNerf" and Fused._tn_reference_get_outputs is standins.StandInTetrahedraNerf.get_outputs
