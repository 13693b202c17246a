"""Generates the committed fixtures under tests/golden/ (run in the build container only;
/root/reference does not exist on the GPU box).

  bottle_mesh.npz    vertices f32 [2549,3] parsed from the reference's test asset
                     (tests/assets/bottle.ply, binary little-endian, 8 floats per vertex)
                     + cells i32 [T,4] = scipy Delaunay of those points (the reference's
                     fixture runs CGAL Delaunay on the same points,
                     tests/test_tetrahedra_tracer.py:13-20; CGAL is absent here).
  bottle_oracle.npz  oracle outputs for config C1 (64x64 rays, M=256): counts, leading
                     segment cells and a checksum -- regression pins for the oracle itself.
  sort_vectors.npz   the hit-list vectors t0..t4 of the reference's tests/test_sort.py:3-687 (real OptiX hit lists
                     of an earlier per-tet-face design: (t, tet, local face) per side of a crossed face) together
                     with the tetrahedra its pairing prototype (test_sort, :690-760, executed from the file) pairs,
                     in order.
  biased_sampler.npz inputs/outputs of the REFERENCE's own map_from_real_distances_to_biased_with_bounds
                     (tetranerf/nerfstudio/model.py:111-122; the function is extracted from the file
                     with ast and executed -- the module itself needs nerfstudio, absent here) on the
                     bottle's hit lists: pins render.biased_sample_bins.
"""
import ast
import hashlib
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import tn_oracle  # noqa: E402

scenes = importlib.import_module("tetra-nerf_amd.scenes")


def read_ply_vertices(path):
    raw = Path(path).read_bytes()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode("ascii").splitlines()
    assert "format binary_little_endian 1.0" in header
    nv = next(int(l.split()[-1]) for l in header if l.startswith("element vertex"))
    props = []
    in_vertex = False
    for l in header:
        if l.startswith("element"):
            in_vertex = l.startswith("element vertex")
        elif l.startswith("property") and in_vertex:
            assert l.split()[1] == "float"
            props.append(l.split()[2])
    v = np.frombuffer(raw, dtype="<f4", count=nv * len(props), offset=end).reshape(nv, len(props))
    return np.ascontiguousarray(v[:, [props.index("x"), props.index("y"), props.index("z")]])


def checksum(res):
    h = hashlib.sha256()
    for k in ("num_visited_cells", "visited_cells", "vertex_indices", "hit_distances",
              "barycentric_coordinates"):
        h.update(np.ascontiguousarray(res[k]).tobytes())
    return h


def main():
    out = Path(__file__).resolve().parent
    verts = read_ply_vertices("/root/reference/tests/assets/bottle.ply").astype(np.float32)
    cells = scenes.delaunay_cells(verts)
    np.savez_compressed(out / "bottle_mesh.npz", vertices=verts, cells=cells)
    print("bottle:", verts.shape, cells.shape)

    tr = tn_oracle.OracleTracer()
    tr.load_tetrahedra(verts, cells)
    o, d = scenes.pinhole_rays(64, 64)
    res = tr.trace_rays(o, d, 256)
    n = res["num_visited_cells"]
    h = checksum(res)
    np.savez_compressed(out / "bottle_oracle.npz", num_visited_cells=n,
                        visited_cells=res["visited_cells"][:, :96].copy(),
                        num_faces=np.int64(len(tr.faces)),
                        sha256=np.frombuffer(h.digest(), np.uint8))
    print("faces", len(tr.faces), "rays hit", int((n > 0).sum()), "max segs", int(n.max()),
          h.hexdigest()[:16])


def reference_function(path, name):
    """Execute one top-level function of a reference source file without importing the module."""
    import torch

    tree = ast.parse(Path(path).read_text())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), str(path), "exec"), ns)
    return ns[name]


def sort_vectors_golden():
    """tests/test_sort.py of the reference: vectors + the result of its own pairing prototype."""
    import copy

    out = Path(__file__).resolve().parent
    tree = ast.parse(Path("/root/reference/tests/test_sort.py").read_text())
    vecs = {n.targets[0].id: ast.literal_eval(n.value) for n in tree.body
            if isinstance(n, ast.Assign) and n.targets[0].id.startswith("t")}
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "test_sort")
    fn.decorator_list = []
    fn.body.append(ast.Return(value=ast.Name(id="o", ctx=ast.Load())))   # hand the paired list back
    ast.fix_missing_locations(fn)
    ns = {"print": lambda *a, **k: None}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "test_sort.py", "exec"), ns)
    data = {}
    for name in ("t0", "t1", "t2", "t3", "t4"):   # the vectors the reference's test is parametrised over (:690)
        v = vecs[name]
        o = ns["test_sort"](copy.deepcopy(v))
        data[name + "_hits"] = np.array([(e[0], e[1], e[2]) for e in v], np.float64)
        data[name + "_paired_tets"] = np.array([o[i][1] for i in range(0, len(o), 2)], np.int64)
        assert all(o[i][1] == o[i + 1][1] for i in range(0, len(o), 2))
        print("sort vector", name, len(v), "sides ->", len(o) // 2, "paired tetrahedra")
    np.savez_compressed(out / "sort_vectors.npz", **data)


def biased_sampler_golden():
    import torch

    out = Path(__file__).resolve().parent
    mesh = np.load(out / "bottle_mesh.npz")
    tr = tn_oracle.OracleTracer()
    tr.load_tetrahedra(mesh["vertices"], mesh["cells"])
    o, d = scenes.pinhole_rays(64, 64)
    res = tr.trace_rays(o, d, 256)
    hit = res["num_visited_cells"] > 0
    nv = res["num_visited_cells"][hit]
    width = int(nv.max())
    hd = np.ascontiguousarray(res["hit_distances"][hit][:, :width])
    S = 48
    near = hd[:, 0, 0][:, None]
    far = np.take_along_axis(hd[:, :, 1], (nv[:, None].astype(np.int64) - 1).clip(0), 1)
    bins = torch.linspace(0.0, 1.0, S + 1)[None].numpy()   # model.py:166,177-178
    samples = (bins * far + (1 - bins) * near).astype(np.float32)
    fn = reference_function("/root/reference/tetranerf/nerfstudio/model.py", "map_from_real_distances_to_biased_with_bounds")
    mapped = fn(torch.from_numpy(nv).long(), torch.from_numpy(hd.copy()), torch.from_numpy(samples.copy())).numpy()
    np.savez_compressed(out / "biased_sampler.npz", num_visited_cells=nv, hit_distances=hd, samples=samples, mapped=mapped)
    print("biased sampler:", mapped.shape, float(mapped.min()), float(mapped.max()))


if __name__ == "__main__":
    main()
    biased_sampler_golden()
    sort_vectors_golden()
