"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/tetranerf_hip.h declares; host-side argument validation mirrors the reference's
error behaviour.  No compute calls (there is no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "tetranerf_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tn_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    import importlib

    _lib = importlib.import_module("tetra-nerf_amd._lib")
    declared = _declared_symbols()
    assert len(declared) >= 12
    assert sorted(_lib.SYMBOLS) == declared
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tetranerf_hip.h but not exported"
    lib.tn_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.tn_version()


def test_python_surface_matches_reference(tn):
    # names of tetranerf.utils.extension (extension/__init__.py:23-26,67-73) and of the pybind
    # module (py_binding.cpp:433-449)
    for name in ("TetrahedraTracer", "triangulate", "gather_uint32", "scatter_ema_uint32_",
                 "interpolate_values", "add_barycentrics_grad"):
        assert hasattr(tn, name)
    for name in ("TetrahedraTracer", "triangulate", "find_average_spacing", "interpolate_values",
                 "interpolate_values_backward", "gather_uint32", "scatter_ema_uint32"):
        assert hasattr(tn.cpp, name)
    for meth in ("trace_rays", "find_visited_cells", "load_tetrahedra", "device"):
        assert hasattr(tn.cpp.TetrahedraTracer, meth)


def test_tracer_rejects_cpu_device(tn):
    import torch

    with pytest.raises(RuntimeError, match="must be a CUDA device"):
        tn.TetrahedraTracer(torch.device("cpu"))


def test_no_cpu_fallback(tn):
    """On a box without a GPU the product must fail loudly, never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        tn.TetrahedraTracer(torch.device("cuda:0"))
    vi = torch.zeros((4, 4), dtype=torch.int32)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        tn.cpp.interpolate_values(vi, torch.zeros((4, 3)), torch.zeros((64, 10)))


def test_product_does_not_import_oracle():
    pkg = ROOT / "tetra-nerf_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.h")):
        text = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
        assert "tn_oracle" not in text and "libtn_oracle" not in text, f


def test_triangulate_and_spacing(tn, scenes):
    import numpy as np
    import torch

    pts = torch.from_numpy(np.random.default_rng(0).random((400, 3)).astype(np.float32))
    cells = tn.triangulate(pts)
    assert cells.dtype == torch.int32 and cells.shape[1] == 4 and len(cells) > 1500
    sp = tn.cpp.find_average_spacing(pts)
    assert 0.02 < sp < 0.3


def test_barycentrics_grad(tn):
    """Mirrors tests/test_barycentrics.py of the reference: analytic gradients of the
    barycentric coordinates w.r.t. points and vertices equal autograd through linalg.solve."""
    import torch

    torch.manual_seed(0)
    verts = torch.randn(5, 4, 3, dtype=torch.float64, requires_grad=True)
    pts = torch.randn(5, 3, dtype=torch.float64, requires_grad=True)

    def bary(v, p):
        return torch.linalg.solve((v[..., 1:, :] - v[..., :1, :]).transpose(-1, -2), (p - v[..., 0, :]).unsqueeze(-1)).squeeze(-1)

    b = bary(verts, pts)
    w = torch.randn_like(b)
    (b * w).sum().backward()
    gv, gp = verts.grad.clone(), pts.grad.clone()
    verts.grad = None
    pts.grad = None
    b2 = tn.add_barycentrics_grad(b.detach(), verts, pts)
    (b2 * w).sum().backward()
    torch.testing.assert_close(pts.grad, gp)
    torch.testing.assert_close(verts.grad, gv)
