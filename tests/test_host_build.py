"""Host-side build products of load_tetrahedra, checked without a GPU: tetra-nerf_amd/csrc/tn_mesh.cpp is plain
C++ (face table, adjacency records, entry-face-specialised walk records, threaded hull tree, wide BVH), so
tests/host/host_build_check.cpp is compiled with g++ and run on a few meshes.  It verifies the structural
invariants the kernels rely on (symmetric neighbour links, walk-record codes that decode to the face table's
stored triples, every hull / mesh face covered exactly once by the trees, boxes containing their triangles) and
dumps the face table, which must equal the oracle's first-seen table (tetrahedra_tracer.cpp:45-71) bit for bit."""
import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "tetra-nerf_amd" / "csrc"


@pytest.fixture(scope="module")
def host_check(tmp_path_factory):
    if shutil.which("g++") is None or not Path("/opt/rocm/include/hip/hip_runtime.h").exists():
        pytest.skip("needs g++ and the HIP headers")
    exe = tmp_path_factory.mktemp("host") / "host_build_check"
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{CSRC}", "-o", str(exe),
           str(ROOT / "tests" / "host" / "host_build_check.cpp"), str(CSRC / "tn_mesh.cpp")]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def _run(exe, tmp_path, pts, cells):
    mesh, out = tmp_path / "mesh.bin", tmp_path / "faces.bin"
    with open(mesh, "wb") as f:
        f.write(struct.pack("<QQ", len(pts), len(cells)))
        f.write(np.ascontiguousarray(pts, np.float32).tobytes())
        f.write(np.ascontiguousarray(cells).astype(np.uint32).tobytes())
    r = subprocess.run([str(exe), str(mesh), str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.stdout, r.stderr)
    raw = out.read_bytes()
    F = struct.unpack("<Q", raw[:8])[0]
    faces = np.frombuffer(raw, np.uint32, 3 * F, 8).reshape(F, 3)
    face_tets = np.frombuffer(raw, np.uint32, 2 * F, 8 + 12 * F).reshape(F, 2)
    return r.stdout, faces, face_tets


@pytest.mark.parametrize("mesh", ["cube", "bottle", "random-3000", "random-20000"])
def test_host_structures(host_check, tmp_path, oracle, scenes, bottle, mesh):
    if mesh == "cube":
        pts, cells = scenes.cube_mesh()
    elif mesh == "bottle":
        pts, cells = bottle["vertices"], bottle["cells"]   # 375 zero-volume tets, duplicated positions
    else:
        pts, cells = scenes.random_mesh(int(mesh.split("-")[1]), 7)
    out, faces, face_tets = _run(host_check, tmp_path, pts, cells)
    want_faces, want_ft = oracle.build_faces(cells)
    np.testing.assert_array_equal(faces, want_faces)
    np.testing.assert_array_equal(face_tets, want_ft)
    assert f"variants {4 * len(cells)}" in out


def test_host_rejects_triple_face(host_check, tmp_path):
    cells = np.array([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 5]], np.int32)
    mesh = tmp_path / "mesh.bin"
    with open(mesh, "wb") as f:
        f.write(struct.pack("<QQ", 6, 3))
        f.write(np.zeros((6, 3), np.float32).tobytes())
        f.write(cells.astype(np.uint32).tobytes())
    r = subprocess.run([str(host_check), str(mesh), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode != 0 and "shared by more than two" in (r.stdout + r.stderr)


@pytest.fixture(scope="module")
def emul_check(tmp_path_factory):
    if shutil.which("g++") is None or not Path("/opt/rocm/include/hip/hip_runtime.h").exists():
        pytest.skip("needs g++ and the HIP headers")
    exe = tmp_path_factory.mktemp("host") / "gpu_build_emul"
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{CSRC}", "-o", str(exe),
           str(ROOT / "tests" / "host" / "gpu_build_emul.cpp"), str(CSRC / "tn_mesh.cpp")]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


@pytest.mark.parametrize("mesh", ["cube", "bottle", "random-3000", "random-20000", "lattice", "near-duplicates"])
def test_device_build_emulated_equals_host_build(emul_check, tmp_path, scenes, bottle, mesh):
    """The element functions of the DEVICE build (csrc/tn_build_core.h, the bodies of the kernels of csrc/tn_build.hip)
    driven from CPU loops -- hash insertions in a shuffled order -- produce byte-identical face tables, walk records
    and hull trees to the host build; the level-synchronous BVH build keeps its invariants.  (The real kernels are
    compared with the host build on the GPU by tests/test_build_gpu.py.)"""
    if mesh == "cube":
        pts, cells = scenes.cube_mesh()
    elif mesh == "bottle":
        pts, cells = bottle["vertices"], bottle["cells"]
    elif mesh == "lattice":
        pts, cells = scenes.grid_mesh(10)
    elif mesh == "near-duplicates":
        pts, cells = scenes.near_duplicates_mesh(2000)
    else:
        pts, cells = scenes.random_mesh(int(mesh.split("-")[1]), 7)
    path = tmp_path / "mesh.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<QQ", len(pts), len(cells)))
        f.write(np.ascontiguousarray(pts, np.float32).tobytes())
        f.write(np.ascontiguousarray(cells).astype(np.uint32).tobytes())
    for leaf_width in (16, 64):     # 16 = the default of the face BVH, 64 = one leaf per wave instruction
        r = subprocess.run([str(emul_check), str(path), str(leaf_width)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("OK"), (leaf_width, r.stdout, r.stderr)
        assert f"variants {4 * len(cells)}" in r.stdout
