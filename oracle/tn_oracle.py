"""numpy/ctypes front-end of the CPU oracle (oracle/tn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under tetra-nerf_amd/ may import this.

Function names and result dictionaries follow the reference's Python surface
(/root/reference/src/py_binding.cpp:41-76,163-216,298-372) so that parity tests read
like the reference's own tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so = _HERE / "libtn_oracle.so"
    src = _HERE / "tn_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _declare(C.CDLL(str(build())))
    return _LIB


def _declare(L):
    L.tno_bvh_build.restype = C.c_void_p
    L.tno_bvh_free.argtypes = [C.c_void_p]
    return L


_FAST = None


def lib_fast():
    """TIMING-ONLY build of the same source: -O3 -march=native (FMA contraction allowed, so NOT bit-exact and never
    used as a checker) -- what bench.py's cpu_baseline leg times.  -march=native is only valid on
    the machine that compiled it, so it is built at run time into the temp directory, not shipped in-tree."""
    global _FAST
    if _FAST is None:
        import hashlib
        import tempfile

        src = _HERE / "tn_oracle.c"
        tag = hashlib.sha1(src.read_bytes()).hexdigest()[:12]
        so = Path(tempfile.gettempdir()) / f"libtn_oracle_fast_{tag}.so"
        if not so.exists():
            tmp = so.with_suffix(f".{os.getpid()}.tmp")
            subprocess.run([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-std=c11", "-fPIC", "-fopenmp",
                            "-shared", "-o", str(tmp), str(src), "-lm"], check=True)
            os.replace(tmp, so)
        _FAST = _declare(C.CDLL(str(so)))
    return _FAST


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.int32:
        return a.view(np.uint32)
    return np.ascontiguousarray(a, dtype=np.uint32)


def build_faces(cells: np.ndarray):
    """tetrahedra -> (faces [F,3] u32 first-seen unsorted, face_tets [F,2] u32).
    Raises RuntimeError with the reference's message on a triply-shared face."""
    cells = _u32(cells).reshape(-1, 4)
    T = cells.shape[0]
    faces = np.empty((4 * T + 1, 3), np.uint32)
    ft = np.empty((4 * T + 1, 2), np.uint32)
    F = C.c_uint64(0)
    rc = lib().tno_build_faces(C.c_uint64(T), _p(cells), _p(faces), _p(ft), C.byref(F))
    if rc == 1:
        raise RuntimeError("A triangle is shared by more than two tetrahedra!")
    if rc:
        raise RuntimeError(f"tno_build_faces failed rc={rc}")
    return faces[: F.value].copy(), ft[: F.value].copy()


class OracleTracer:
    """Mirror of tetranerf_cpp_extension.TetrahedraTracer on numpy arrays."""

    def __init__(self, use_bvh: bool = False, threads: int = 0, fast: bool = False):
        self.use_bvh = use_bvh
        self.threads = threads
        self._bvh = None
        self.xyz = None
        self._L = lib_fast() if fast else lib()   # fast: timing-only -O3 -march=native build (not bit-exact)

    def load_tetrahedra(self, xyz, cells):
        self.xyz = _f32(xyz).reshape(-1, 3)
        self.cells = _u32(cells).reshape(-1, 4)
        self.faces, self.face_tets = build_faces(self.cells)
        self._free()
        if self.use_bvh:
            self._bvh = C.c_void_p(self._L.tno_bvh_build(
                C.c_uint64(len(self.xyz)), _p(self.xyz), C.c_uint64(len(self.faces)), _p(self.faces)))

    def _free(self):
        if self._bvh is not None:
            self._L.tno_bvh_free(self._bvh)
            self._bvh = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def trace_rays(self, origins, directions, max_ray_triangles: int, with_raw: bool = False):
        M = int(max_ray_triangles)
        if M <= 0 or (M & (M - 1)) != 0:
            raise RuntimeError("max_ray_triangles must be a power of 2.")
        o = _f32(origins).reshape(-1, 3)
        d = _f32(directions).reshape(-1, 3)
        R = o.shape[0]
        out = {
            "num_visited_cells": np.zeros(R, np.uint32),
            "visited_cells": np.zeros((R, M), np.uint32),
            "barycentric_coordinates": np.zeros((R, M, 2, 3), np.float32),
            "hit_distances": np.zeros((R, M, 2), np.float32),
            "vertex_indices": np.zeros((R, M, 4), np.uint32),
        }
        raw = None
        if with_raw:
            raw = {"count": np.zeros(R, np.uint32), "ids": np.zeros((R, M), np.uint32),
                   "t": np.zeros((R, M), np.float32), "uv": np.zeros((R, M, 2), np.float32)}
        rc = self._L.tno_trace_rays(
            C.c_uint64(len(self.xyz)), _p(self.xyz), C.c_uint64(len(self.faces)), _p(self.faces),
            _p(self.face_tets), self._bvh, C.c_uint64(R), C.c_uint32(M), _p(o), _p(d),
            _p(out["num_visited_cells"]), _p(out["visited_cells"]), _p(out["barycentric_coordinates"]),
            _p(out["hit_distances"]), _p(out["vertex_indices"]),
            _p(raw["count"]) if raw else None, _p(raw["ids"]) if raw else None,
            _p(raw["t"]) if raw else None, _p(raw["uv"]) if raw else None, C.c_int(self.threads))
        if rc:
            raise RuntimeError(f"tno_trace_rays failed rc={rc}")
        res = {k: (v.view(np.int32) if v.dtype == np.uint32 else v) for k, v in out.items()}
        if with_raw:
            res["raw"] = raw
        return res

    def trace_rays_triangles(self, origins, directions, max_ray_triangles):
        """Sorted all-hits list (reference: trace_rays_triangles, py_binding.cpp:78-113)."""
        raw = self.trace_rays(origins, directions, max_ray_triangles, with_raw=True)["raw"]
        ids = raw["ids"].copy()
        live = np.arange(ids.shape[1])[None, :] < raw["count"][:, None]
        ids[~live] = 0  # slots >= count keep the zeros of the reference's torch::zeros outputs (py_binding.cpp:90-94)
        v3 = np.where(live[..., None], self.faces[np.minimum(ids, len(self.faces) - 1).astype(np.int64)], 0)
        return {"num_visited_triangles": raw["count"].view(np.int32), "visited_triangles": ids.view(np.int32),
                "barycentric_coordinates": raw["uv"], "hit_distances": raw["t"],
                "vertex_indices": np.ascontiguousarray(v3.astype(np.uint32)).view(np.int32)}

    def find_tetrahedra(self, positions):
        pts = _f32(positions).reshape(-1, 3)
        N = len(pts)
        tets = np.empty(N, np.uint32)
        bary = np.empty((N, 3), np.float32)
        verts = np.empty((N, 4), np.uint32)
        rc = lib().tno_find_tetrahedra(_p(self.xyz), C.c_uint64(len(self.faces)), _p(self.faces), _p(self.face_tets),
                                       C.c_uint64(N), _p(pts), _p(tets), _p(bary), _p(verts))
        if rc:
            raise RuntimeError(f"tno_find_tetrahedra failed rc={rc}")
        t = tets.view(np.int32)
        return {"tetrahedra": t, "barycentric_coordinates": bary, "vertex_indices": verts.view(np.int32), "valid_mask": t != -1}

    def find_visited_cells(self, num_visited_cells, visited_cells, barycentric_coordinates,
                           hit_distances, vertex_indices, distances):
        return find_visited_cells(num_visited_cells, visited_cells, barycentric_coordinates,
                                  hit_distances, vertex_indices, distances)


def postprocess(faces, face_tets, hit_count, hit_ids, hit_t, hit_uv):
    """Run the dedupe/pairing stage on caller-supplied sorted hit rows [R,M]."""
    hit_ids = _u32(hit_ids)
    R, M = hit_ids.shape
    out = {
        "num_visited_cells": np.zeros(R, np.uint32),
        "visited_cells": np.zeros((R, M), np.uint32),
        "barycentric_coordinates": np.zeros((R, M, 2, 3), np.float32),
        "hit_distances": np.zeros((R, M, 2), np.float32),
        "vertex_indices": np.zeros((R, M, 4), np.uint32),
    }
    rc = lib().tno_postprocess(_p(_u32(faces)), _p(_u32(face_tets)), C.c_uint64(R), C.c_uint32(M),
                               _p(_u32(hit_count)), _p(hit_ids), _p(_f32(hit_t)), _p(_f32(hit_uv)),
                               _p(out["num_visited_cells"]), _p(out["visited_cells"]),
                               _p(out["barycentric_coordinates"]), _p(out["hit_distances"]),
                               _p(out["vertex_indices"]))
    if rc:
        raise RuntimeError(f"tno_postprocess failed rc={rc}")
    return {k: (v.view(np.int32) if v.dtype == np.uint32 else v) for k, v in out.items()}


def find_visited_cells(num_visited_cells, visited_cells, barycentric_coordinates, hit_distances,
                       vertex_indices, distances, fast: bool = False):
    """fast = the timing-only build (bench.py's cpu_baseline; never used as a checker)."""
    nv = _u32(num_visited_cells)
    vc = _u32(visited_cells)
    R, M = vc.shape
    dist = _f32(distances)
    assert dist.ndim == 2 and dist.shape[0] == R
    S = dist.shape[1]
    cells = np.empty((R, S), np.uint32)
    verts = np.empty((R, S, 4), np.uint32)
    mask = np.empty((R, S), np.uint8)
    bary = np.empty((R, S, 3), np.float32)
    rc = (lib_fast() if fast else lib()).tno_find_matched_cells(
        C.c_uint64(R), C.c_uint64(S), C.c_uint64(M), _p(nv), _p(vc), _p(_f32(hit_distances)),
        _p(_f32(barycentric_coordinates)), _p(dist), _p(_u32(vertex_indices)),
        _p(cells), _p(verts), _p(mask), _p(bary))
    if rc:
        raise RuntimeError(f"tno_find_matched_cells failed rc={rc}")
    return {"cell_indices": cells.view(np.int32), "vertex_indices": verts.view(np.int32),
            "mask": mask.astype(bool), "barycentric_coordinates": bary}


def interpolate_values(vertex_indices, barycentric_coordinates, field, fast: bool = False):
    vi = _u32(vertex_indices)
    bc = _f32(barycentric_coordinates)
    field = _f32(field)
    D = vi.shape[-1]
    assert bc.shape[-1] + 1 == D
    n = vi.size // D
    Fd, V = field.shape
    out = np.empty((Fd, n), np.float32)
    rc = (lib_fast() if fast else lib()).tno_interpolate_values(C.c_uint32(D), C.c_uint32(V), C.c_uint32(n), C.c_uint32(Fd),
                                      _p(vi), _p(bc), _p(field), _p(out))
    if rc:
        raise RuntimeError(f"Unsupported interpolation dimension with value {D}")
    return np.moveaxis(out.reshape((Fd,) + vi.shape[:-1]), 0, -1)


def interpolate_values_backward(vertex_indices, barycentric_coordinates, field, grad_in, fast: bool = False):
    vi = _u32(vertex_indices)
    bc = _f32(barycentric_coordinates)
    field = _f32(field)
    D = vi.shape[-1]
    n = vi.size // D
    Fd, V = field.shape
    g = np.ascontiguousarray(np.moveaxis(_f32(grad_in), -1, 0)).reshape(Fd, n)
    out = np.empty((Fd, V), np.float32)
    rc = (lib_fast() if fast else lib()).tno_interpolate_values_backward(C.c_uint32(D), C.c_uint32(V), C.c_uint32(n),
                                               C.c_uint32(Fd), _p(vi), _p(bc), _p(g), _p(out))
    if rc:
        raise RuntimeError(f"Unsupported interpolation dimension with value {D}")
    return out


def num_threads() -> int:
    return int(lib().tno_num_threads())
