"""Loader of the reference's OWN nerfstudio model file for the tests (TEST INFRASTRUCTURE, never imported by the product).

`tests/golden/ref/tetranerf/nerfstudio/model.py` is a verbatim copy of `/root/reference/tetranerf/nerfstudio/model.py`
(staged by `stage_reference_tests.py`; `/root/reference` does not exist on the GPU box).  It imports nerfstudio,
torchmetrics and scikit-image, none of which is installed: `tests/golden/nerfstudio_stub/` provides them (a restatement
of nerfstudio 0.3.4's public algorithms for exactly the imported names, see its README).  What these helpers give the
tests is the reference's `TetrahedraNerf` (its `__init__`, `populate_modules`, `get_tetrahedra_tracer`, `get_outputs`,
`get_background_color`, state-dict names), `TetrahedraSampler`, `GradientScaler` and
`map_from_real_distances_to_biased_with_bounds` -- executed, not restated.

    ref = load()                         # the module tetranerf.nerfstudio.model
    model = build_model(ref, vertices, cells, num_samples=..., ...)     # a TetrahedraNerf on CPU; .to(device) as usual
    cpu_model(ref, model, oracle)        # CPU only: oracle-backed tracer + einsum gather instead of the HIP ops
"""
from __future__ import annotations

import importlib
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]


def load():
    """Import (once) and return the staged reference module `tetranerf.nerfstudio.model`."""
    for p in (str(ROOT), str(HERE / "nerfstudio_stub"), str(HERE / "ref")):
        if p not in sys.path:
            sys.path.append(p)
    return importlib.import_module("tetranerf.nerfstudio.model")


def build_model(ref, vertices, cells, *, num_train_data: int = 3, far_plane: float = 1000.0, seed: int = 0,
                density_boost: float = 0.0, **cfg):
    """A reference TetrahedraNerf built the way `TetrahedraNerfPipeline` builds it (pipeline.py:39-49) minus the data
    manager: config with the mesh sizes -> `TetrahedraNerf(config, scene_box=None, num_train_data=n)` -> buffers filled
    and the field initialised as `_init_tetrahedra` does (model.py:377-386: uniform(-1e-4, 1e-4), rows 1-3 = colours,
    row 0 = alpha), plus a density-driving spread in row 0 and `density_boost` on the density head's bias so that the
    render is not all-transparent on small meshes."""
    vertices = torch.as_tensor(np.asarray(vertices), dtype=torch.float32)
    cells = torch.as_tensor(np.asarray(cells).astype(np.int32))
    torch.manual_seed(seed)
    config = ref.TetrahedraNerfConfig(num_tetrahedra_vertices=len(vertices), num_tetrahedra_cells=len(cells),
                                      collider_params={"near_plane": 0.0, "far_plane": float(far_plane)}, **cfg)
    model = ref.TetrahedraNerf(config, scene_box=None, num_train_data=num_train_data)
    with torch.no_grad():
        model.tetrahedra_vertices.copy_(vertices)
        model.tetrahedra_cells.copy_(cells)
        model._init_tetrahedra_field(model.tetrahedra_field.data)
        model.tetrahedra_field.data[1:4] = torch.rand(3, len(vertices)) * 2 - 1
        model.tetrahedra_field.data[0] = torch.rand(len(vertices)) * 6 - 3
        model.tetrahedra_field.data[4:] = torch.randn(model.tetrahedra_field.shape[0] - 4, len(vertices)) * 0.3
        model.field_output_density.net.bias.add_(float(density_boost))
    model._tetrahedra_initialized = True
    return model


def ray_bundle(ref, origins, directions, device=None, camera_indices=None):
    """A (stub-)nerfstudio RayBundle over [R,3] origins / directions."""
    from nerfstudio.cameras.rays import RayBundle

    o = origins if isinstance(origins, torch.Tensor) else torch.as_tensor(np.asarray(origins), dtype=torch.float32)
    d = directions if isinstance(directions, torch.Tensor) else torch.as_tensor(np.asarray(directions), dtype=torch.float32)
    ci = None if camera_indices is None else torch.as_tensor(camera_indices).reshape(-1, 1)
    if device is not None:
        o, d = o.to(device), d.to(device)
        ci = None if ci is None else ci.to(device)
    return RayBundle(origins=o, directions=d, pixel_area=torch.ones_like(o[:, :1]), camera_indices=ci)


def einsum_interpolate_values(vertex_indices, barycentric_coordinates, field):
    """The reference's own definition of `interpolate_values` (tests/test_tetrahedra_tracer.py:410-415): differentiable
    plain PyTorch, for running the reference model on the CPU."""
    full = torch.cat((1 - barycentric_coordinates.sum(-1, keepdim=True), barycentric_coordinates), -1)
    vi = vertex_indices.long().clamp_min(0)
    return torch.einsum("j...i,...i->...j", field[:, vi], full)


class OracleTorchTracer:
    """`TetrahedraTracer` surface (trace_rays / find_visited_cells on torch CPU tensors) backed by the CPU oracle."""

    def __init__(self, oracle, vertices, cells):
        self.device = torch.device("cpu")
        self._t = oracle.OracleTracer(use_bvh=True)
        self._t.load_tetrahedra(np.ascontiguousarray(vertices, dtype=np.float32), np.ascontiguousarray(cells).astype(np.uint32))

    @staticmethod
    def _wrap(d):
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}

    def trace_rays(self, origins, directions, max_ray_triangles):
        return self._wrap(self._t.trace_rays(origins.numpy(), directions.numpy(), int(max_ray_triangles)))

    def find_visited_cells(self, *a, ray_index=None):
        a = [x.detach() for x in a]
        if ray_index is not None:      # the product's extension of the op: rows named by an index instead of compacted
            a = [x[ray_index.long()] for x in a[:5]] + [a[5]]
        return self._wrap(self._t.find_visited_cells(*[np.ascontiguousarray(x.numpy()) for x in a]))

    def set_option(self, name, value):
        pass


def cpu_model(ref, model, oracle, monkeypatch):
    """Make a reference model runnable on the CPU: its `get_outputs` body stays the reference's, only the two native ops
    it calls are served by the oracle tracer and the einsum definition (the reference refuses non-CUDA devices in
    `get_tetrahedra_tracer`, model.py:396-397, so that accessor is bound on the instance; the module-level
    `interpolate_values` is swapped through pytest's monkeypatch and restored after the test).  Returns the tracer."""
    tracer = OracleTorchTracer(oracle, model.tetrahedra_vertices.numpy(), model.tetrahedra_cells.numpy())
    model.get_tetrahedra_tracer = lambda: tracer
    # nerfstudio's get_background_color returns `colour.expand(shape).to(device)`: on a CUDA device the copy materialises
    # the rows, on the CPU `.to` is a no-op and model.py:652 (`rgb[ray_mask] = rgb_r`) would write through a stride-0 view
    # of the 3-element colour.  A CPU-only artefact of running a CUDA-only model here: materialise the rows.
    get_bg = model.get_background_color
    model.get_background_color = lambda shape, device: get_bg(shape, device).clone()
    monkeypatch.setattr(ref, "interpolate_values", einsum_interpolate_values)
    return tracer
