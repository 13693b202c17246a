#!/usr/bin/env python3
"""Stages the reference's OWN Python layer and tests next to the one-file shim of INTEGRATION.md section 2, so that
they can be run UNMODIFIED on top of libtetranerf_hip.so on a GPU box (where /root/reference does not exist):

    tests/golden/ref/tetranerf/__init__.py                       <- /root/reference/tetranerf/__init__.py
    tests/golden/ref/tetranerf/utils/__init__.py                 <- (empty in the reference)
    tests/golden/ref/tetranerf/utils/extension/__init__.py       <- the reference's op wrappers (autograd Function, aliases)
    tests/golden/ref/tetranerf/utils/extension/tetranerf_cpp_extension.py   <- THE SHIM (ours; written below)
    tests/golden/ref/tetranerf/nerfstudio/{__init__,model}.py    <- the reference's nerfstudio model, verbatim (imports
                                                                    nerfstudio: tests/golden/nerfstudio_stub/ provides it)
    tests/golden/ref/tests/{test_uint32,test_barycentrics,test_tetrahedra_tracer,test_tetrahedra_tracer_triangles}.py
    tests/golden/ref/tests/assets/bottle.ply
    tests/golden/ref/trimesh.py                                  <- 20-line stand-in for `trimesh.load` (not installed here)
    tests/golden/ref/LICENSE                                     <- the reference's MIT licence (required to accompany copies)

The staged reference files are verbatim copies: TEST INFRASTRUCTURE (fixtures), never imported by the product.
tests/test_reference_suite_gpu.py runs them with pytest in a subprocess.  Re-run this script to refresh them:
    python tests/golden/stage_reference_tests.py [/root/reference]
"""
import shutil
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "ref"

COPIES = [
    "LICENSE",                      # the reference is MIT-licensed: its copyright + permission notice accompanies the copies
    "tetranerf/__init__.py",
    "tetranerf/utils/__init__.py",
    "tetranerf/utils/extension/__init__.py",
    "tetranerf/nerfstudio/__init__.py",
    "tetranerf/nerfstudio/model.py",    # the model whose get_outputs / TetrahedraSampler / GradientScaler are the oracle
                                        # of the fused adapter (tests/test_reference_model*.py; needs ../nerfstudio_stub)
    "tests/test_uint32.py",
    "tests/test_barycentrics.py",
    "tests/test_tetrahedra_tracer.py",
    "tests/test_tetrahedra_tracer_triangles.py",
    "tests/assets/bottle.ply",
]

SHIM = '''# tetranerf/utils/extension/tetranerf_cpp_extension.py   (ROCm / MI355X build) -- INTEGRATION.md section 2.
# The ONE file a maintainer adds to the reference instead of the CUDA/OptiX-built pybind11 module
# (src/py_binding.cpp:433-449): every name the reference binds, backed by libtetranerf_hip.so.
import importlib

_impl = importlib.import_module("tetra-nerf_amd.tetranerf_cpp_extension")
TetrahedraTracer = _impl.TetrahedraTracer                        # src/py_binding.cpp:434-440
triangulate = _impl.triangulate                                  # :442
find_average_spacing = _impl.find_average_spacing                # :443
interpolate_values = _impl.interpolate_values                    # :444
interpolate_values_backward = _impl.interpolate_values_backward  # :445
gather_uint32 = _impl.gather_uint32                              # :446
scatter_ema_uint32 = _impl.scatter_ema_uint32                    # :447
'''

TRIMESH = '''"""Stand-in for the two trimesh calls the reference's tests make (`trimesh.load(path).vertices` on a binary
little-endian PLY point cloud, `trimesh.PointCloud(vertices=...).export(path)`); trimesh is not installed in this image."""
import numpy as np


class _Cloud:
    def __init__(self, vertices):
        self.vertices = vertices


class PointCloud:
    def __init__(self, vertices=None, colors=None):
        self.vertices = np.asarray(vertices, dtype=np.float64)

    def export(self, path):
        v = self.vertices.reshape(-1, 3)
        with open(path, "w") as f:
            f.write("ply\\nformat ascii 1.0\\nelement vertex %d\\nproperty float x\\nproperty float y\\nproperty float z\\nend_header\\n" % len(v))
            np.savetxt(f, v, fmt="%.7g")


def load(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        n, props, fmt = 0, [], None
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property") and "list" not in line:
                props.append(line.split()[1:3])
            elif line == "end_header":
                break
        assert fmt == "binary_little_endian", fmt
        kinds = {"float": "<f4", "float32": "<f4", "double": "<f8", "uchar": "u1", "uint8": "u1", "int": "<i4", "uint": "<u4"}
        dt = np.dtype([(name, kinds[t]) for t, name in props])
        data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    return _Cloud(np.stack([data["x"], data["y"], data["z"]], -1).astype(np.float64))
'''


def main():
    if OUT.exists():
        shutil.rmtree(OUT)
    for rel in COPIES:
        dst = OUT / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copyfile(REF / rel, dst)
    (OUT / "tetranerf/utils/extension/tetranerf_cpp_extension.py").write_text(SHIM)
    (OUT / "trimesh.py").write_text(TRIMESH)
    (OUT / "pytest.ini").write_text("[pytest]\npythonpath = .\n")
    (OUT / "README.md").write_text(
        "Verbatim copies of the reference's Python op layer and tests (jkulhanek/tetra-nerf, `tetranerf/__init__.py`,\n"
        "`tetranerf/utils/extension/__init__.py`, `tests/test_{uint32,barycentrics,tetrahedra_tracer,tetrahedra_tracer_triangles}.py`,\n"
        "`tests/assets/bottle.ply`, `tetranerf/nerfstudio/{__init__,model}.py`), staged by `tests/golden/stage_reference_tests.py` as TEST FIXTURES so that they can run\n"
        "unmodified against the MI355X library on a GPU box.  Ours in this directory: `tetranerf/utils/extension/tetranerf_cpp_extension.py`\n"
        "(the one-file shim of INTEGRATION.md section 2), `trimesh.py` (stand-in for `trimesh.load`), `pytest.ini`, this file.\n")
    print("staged", len(COPIES), "reference files under", OUT)


if __name__ == "__main__":
    main()
