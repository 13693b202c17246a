"""Drop-in for the reference's pybind11 module `tetranerf_cpp_extension`
(/root/reference/src/py_binding.cpp:433-449), backed by libtetranerf_hip.so on MI355X.

Same names, argument meaning, result dictionaries, dtypes/shapes and error behaviour:

    TetrahedraTracer(device)                      py_binding.cpp:28-40
      .device                                     :218-220, 436
      .load_tetrahedra(xyz, cells)                :144-161
      .trace_rays(origins, directions, M)         :41-76
      .find_visited_cells(...)                    :163-216
    interpolate_values(vi, bc, field)             :298-330
    interpolate_values_backward(vi, bc, field, g) :341-372
    triangulate(points), find_average_spacing(points)   :229-256 (CGAL there; scipy here, CPU, offline)
    gather_uint32 / scatter_ema_uint32            :374-431 (not on the model path)

Host code is PyTorch plumbing only: tensor checks, output allocation (torch.empty -- the
kernels write every byte, so the reference's torch::zeros memset pass is not needed) and raw
pointers + the current HIP stream handed to the C-ABI.  Nothing here computes on the CPU and
nothing imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib


_POISON = os.environ.get("TETRANERF_HIP_POISON", "") not in ("", "0")


def _empty(*shape, dtype=torch.float32, device=None):
    """torch.empty for outputs and scratch (every kernel writes each byte it owns).  TETRANERF_HIP_POISON=1 (tests,
    debugging) pre-fills them with NaN / 0x7f7f7f7f so that a byte a kernel failed to write, or read before writing,
    shows up instead of inheriting whatever the allocator's block held."""
    t = torch.empty(*shape, dtype=dtype, device=device)
    if _POISON and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype == torch.bool:
            t.fill_(True)
        else:
            t.fill_(0x7F7F7F7F if t.dtype in (torch.int32, torch.int64) else 0x7F)
    return t


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_input(x, name):
    _check(isinstance(x, torch.Tensor), f"{name} must be a tensor")
    _check(x.device.type == "cuda", f"{name} must be a CUDA tensor")
    _check(x.is_contiguous(), f"{name} must be contiguous")


# The wrappers below are on the path of every op call: at 4096 x 513 samples find_visited_cells is an 18 us kernel behind what
# used to be 21-27 us of Python per call (profiles/r06av_ops_host.txt), so the helpers avoid object construction: the raw
# stream handle instead of a torch Stream object, plain ints for pointers (every entry point has ctypes argtypes, which
# convert an int to a void pointer), and a device context only when the tracer's device is not the current one.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device) -> int:
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(device):
    """`with _on(dev):` = torch.cuda.device(dev) when another device is current, nothing otherwise."""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


class TetrahedraTracer:
    """OptiX-free tetrahedra tracer; API of PyTetrahedraTracer (py_binding.cpp:28-227)."""

    def __init__(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("The device argument must be a CUDA device.")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._device = device
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.tn_tracer_create(int(device.index), C.byref(h)))
        self._h = h
        self.tetrahedra_vertices = None
        self.tetrahedra_cells = None

    # read-only property, compared by the model to decide re-creation (model.py:398-402)
    @property
    def device(self):
        return self._device

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.tn_tracer_destroy(h)
        self.tetrahedra_vertices = None
        self.tetrahedra_cells = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_float_dim3(self, x, name):
        _check_input(x, name)
        _check(x.device == self._device, f"{name} must be on the same device")
        _check(x.dtype == torch.float32, f"{name} must have float32 type")
        _check(x.dim() >= 1 and x.size(-1) == 3, f"{name} must have last dimension with size 3")

    def load_tetrahedra(self, xyz, cells):
        self._check_float_dim3(xyz, "xyz")
        _check_input(cells, "cells")
        _check(cells.device == self._device, "cells must be on the same device")
        _check(cells.dim() >= 1 and cells.size(-1) == 4, "indices must have last dimension with size 4")
        _check(cells.dtype == torch.int32, "indices must have int32 type")
        # borrowed, not copied (tetrahedra_tracer.h:300-303): keep them alive
        self.tetrahedra_cells = cells
        self.tetrahedra_vertices = xyz
        invalidate_field_cache()   # a new mesh comes with a (re-)initialised field (model.py:349-392 writes it through .data)
        _lib.check(self._lib.tn_load_tetrahedra(
            self._h, xyz.numel() // 3, cells.numel() // 4, _ptr(xyz), _ptr(cells), _stream(self._device)))

    supports_compact_rows = True

    def trace_rays(self, ray_origins, ray_directions, max_ray_triangles, compact_rows: bool = False):
        """PyTetrahedraTracer::trace_rays (py_binding.cpp:41-76).  compact_rows (no reference counterpart; per call):
        slots >= num_visited_cells are left unwritten (tn_trace_rays_ex + TN_TRACE_COMPACT_ROWS) for consumers that read
        the rows only through num_visited_cells (the samplers, find_visited_cells(ray_index=...), render_rays)."""
        M = int(max_ray_triangles)
        if M <= 0 or (M & (M - 1)) != 0:
            raise RuntimeError("max_ray_triangles must be a power of 2.")
        with torch.no_grad():
            self._check_float_dim3(ray_origins, "ray_origins")
            self._check_float_dim3(ray_directions, "ray_directions")
            R = ray_origins.numel() // 3
            _check(ray_directions.numel() // 3 == R, "ray_origins and ray_directions must have the same number of rays")
            dev = self._device
            num_visited_cells = _empty((R,), dtype=torch.int32, device=dev)
            visited_cells = _empty((R, M), dtype=torch.int32, device=dev)
            barycentric_coordinates = _empty((R, M, 2, 3), dtype=torch.float32, device=dev)
            hit_distances = _empty((R, M, 2), dtype=torch.float32, device=dev)
            vertex_indices = _empty((R, M, 4), dtype=torch.int32, device=dev)
            _lib.check(self._lib.tn_trace_rays_ex(
                self._h, R, M, _ptr(ray_origins), _ptr(ray_directions), _ptr(num_visited_cells),
                _ptr(visited_cells), _ptr(barycentric_coordinates), _ptr(hit_distances),
                _ptr(vertex_indices), 1 if compact_rows else 0, _stream(dev)))
        return {
            "num_visited_cells": num_visited_cells,
            "visited_cells": visited_cells,
            "barycentric_coordinates": barycentric_coordinates,
            "vertex_indices": vertex_indices,
            "hit_distances": hit_distances,
        }

    def trace_rays_triangles(self, ray_origins, ray_directions, max_ray_triangles):
        """PyTetrahedraTracer::trace_rays_triangles (py_binding.cpp:78-113): the sorted all-hits list."""
        M = int(max_ray_triangles)
        if M <= 0 or (M & (M - 1)) != 0:
            raise RuntimeError("max_ray_triangles must be a power of 2.")
        with torch.no_grad():
            self._check_float_dim3(ray_origins, "ray_origins")
            self._check_float_dim3(ray_directions, "ray_directions")
            R = ray_origins.numel() // 3
            _check(ray_directions.numel() // 3 == R, "ray_origins and ray_directions must have the same number of rays")
            dev = self._device
            num = _empty((R,), dtype=torch.int32, device=dev)
            vis = _empty((R, M), dtype=torch.int32, device=dev)
            bary = _empty((R, M, 2), dtype=torch.float32, device=dev)
            dist = _empty((R, M), dtype=torch.float32, device=dev)
            verts = _empty((R, M, 3), dtype=torch.int32, device=dev)
            _lib.check(self._lib.tn_trace_rays_triangles(self._h, R, M, _ptr(ray_origins), _ptr(ray_directions), _ptr(num),
                                                         _ptr(vis), _ptr(bary), _ptr(dist), _ptr(verts), _stream(dev)))
        return {"num_visited_triangles": num, "visited_triangles": vis, "barycentric_coordinates": bary,
                "vertex_indices": verts, "hit_distances": dist}

    def find_tetrahedra(self, positions):
        """PyTetrahedraTracer::find_tetrahedra (py_binding.cpp:115-142): point location."""
        with torch.no_grad():
            self._check_float_dim3(positions, "positions")
            shape = tuple(positions.shape[:-1])
            N = positions.numel() // 3
            dev = self._device
            bary = _empty(shape + (3,), dtype=torch.float32, device=dev)
            verts = _empty(shape + (4,), dtype=torch.int32, device=dev)
            tets = _empty(shape, dtype=torch.int32, device=dev)
            _lib.check(self._lib.tn_find_tetrahedra(self._h, N, _ptr(positions), _ptr(tets), _ptr(bary), _ptr(verts),
                                                    _stream(dev)))
        return {"tetrahedra": tets, "barycentric_coordinates": bary, "vertex_indices": verts, "valid_mask": tets != -1}

    def find_visited_cells(self, num_visited_cells, visited_cells, barycentric_coordinates,
                           hit_distances, vertex_indices, distances, ray_index=None, count=None):
        """py_binding.cpp:163-216.  Addition: `ray_index` (int32 [r]) matches a SUBSET of the traced rays without
        compacting their rows first -- `distances` and the results are [r, S...], the trace tensors stay [R, M...];
        `count` (int32 [1] device tensor, with ray_index): only the first count[0] rows are matched (compact_hits)."""
        sdev = self._device
        for x, name in ((num_visited_cells, "num_visited_cells"), (visited_cells, "visited_cells"),
                        (barycentric_coordinates, "barycentric_coordinates"),
                        (hit_distances, "hit_distances"), (distances, "distances"),
                        (vertex_indices, "vertex_indices")):
            if not (isinstance(x, torch.Tensor) and x.is_cuda and x.is_contiguous() and x.device == sdev):
                _check_input(x, name)                                   # (raises with the reference's messages)
                _check(x.device == sdev, f"{name} must be on the same device")
        _check(distances.dtype == torch.float32, "distances must have float32 type")
        R = num_visited_cells.size(0)
        _check(count is None or ray_index is not None, "find_visited_cells: `count` needs `ray_index` (it counts rows of that list)")
        if ray_index is not None:
            _check_input(ray_index, "ray_index")
            _check(ray_index.dtype == torch.int32 and ray_index.dim() == 1, "ray_index must be int32 [r]")
            _check(ray_index.device == self._device, "ray_index must be on the same device")
            R = ray_index.size(0)
        _check(distances.dim() == 2 and distances.size(0) == R,
               "distances must be of [num_rays, num_samples_per_ray] shape")
        _check(vertex_indices.size(-1) == 4, "vertex_indices must have last dimension with size 4")
        _check(self.tetrahedra_vertices is not None, "load_tetrahedra must be called first")
        _check(num_visited_cells.dtype == torch.int32 and visited_cells.dtype == torch.int32
               and vertex_indices.dtype == torch.int32, "index tensors must have int32 type")
        _check(hit_distances.dtype == torch.float32 and barycentric_coordinates.dtype == torch.float32,
               "hit_distances / barycentric_coordinates must have float32 type")
        S = distances.size(-1)
        M = visited_cells.size(1)
        dev = self._device
        mask = _empty((R, S), dtype=torch.bool, device=dev)
        matched_cells = _empty((R, S), dtype=torch.int32, device=dev)
        barycentric_coordinates_out = _empty((R, S, 3), dtype=torch.float32, device=dev)
        vertex_indices_out = _empty((R, S, 4), dtype=torch.int32, device=dev)
        # the C entry points take no device argument: launch with the tracer's device current (its stream, its pointers)
        with _on(dev):
            if ray_index is None:
                _lib.check(self._lib.tn_find_matched_cells(
                    R, S, M, _ptr(num_visited_cells), _ptr(visited_cells), _ptr(hit_distances),
                    _ptr(barycentric_coordinates), _ptr(distances), _ptr(vertex_indices), _ptr(matched_cells),
                    _ptr(vertex_indices_out), _ptr(mask), _ptr(barycentric_coordinates_out), _stream(dev)))
            else:
                _lib.check(self._lib.tn_find_matched_cells_indexed(
                    R, S, M, _ptr(ray_index), _ptr(num_visited_cells), _ptr(visited_cells), _ptr(hit_distances),
                    _ptr(barycentric_coordinates), _ptr(distances), _ptr(vertex_indices), _ptr(matched_cells),
                    _ptr(vertex_indices_out), _ptr(mask), _ptr(barycentric_coordinates_out), _ptr(count), _stream(dev)))
        return {
            "cell_indices": matched_cells,
            "vertex_indices": vertex_indices_out,
            "mask": mask,
            "barycentric_coordinates": barycentric_coordinates_out,
        }

    # -- additions (not in the reference surface) ------------------------------------------
    def trace_stats(self):
        """Counters of the last trace_rays call: rays certified by the walk / all others (literal pairing of the log +
        BVH re-trace) / rays whose pairing ran the serial literal branch / rays that overflowed M-1 hits."""
        arr = (C.c_uint64 * 4)()
        _lib.check(self._lib.tn_trace_stats(self._h, C.byref(arr)))
        return {"walk": arr[0], "general": arr[1], "serial": arr[2], "overflow": arr[3]}

    def flag_reasons(self):
        """Why the walk did not certify rays of the last trace_rays (reason code 1..12 -> count; include/tetranerf_hip.h);
        7 = sound chain with a gap below eps / a tie / an inversion, 13 = such rays whose logged hits went through the
        literal sort + pairing (the others were re-traced through the BVH); with option verify_stride: 15 = certified rays
        cross-checked against a count-only BVH traversal, 14 = those whose face count differed (handed to the BVH path)."""
        arr = (C.c_uint64 * 16)()
        _lib.check(self._lib.tn_trace_flag_reasons(self._h, C.byref(arr)))
        return {k: int(arr[k]) for k in range(1, 16) if arr[k]}

    def set_option(self, name: str, value: int):
        _lib.check(self._lib.tn_set_option(self._h, name.encode(), int(value)))

    def cross_check(self):
        """The cross-check of the walk's certification in the last trace_rays call (tn_trace_cross_check): the blind sample
        (every stride-th certified ray) and the risk classes (EVERY certified ray inside the wide band of a guard)."""
        arr = (C.c_uint64 * 8)()
        _lib.check(self._lib.tn_trace_cross_check(self._h, C.byref(arr)))
        return {"stride": int(arr[0]), "checked": int(arr[1]), "mismatches": int(arr[2]),
                "risk": {"hull_near_miss_rays": int(arr[3]), "thin_neighbourhood_rays": int(arr[4]), "checked": int(arr[5]),
                         "mismatches": int(arr[6])}}

    TIMING_KEYS = ("speculative_fill", "walk", "bvh_fallback", "cross_check", "segment_writer", "literal_pairing", "tail_fill",
                   "cross_check_retrace")

    def trace_timings(self):
        """Per-kernel milliseconds of the last one-chunk walk call traced with set_option("timing", 1) (the kernels then run
        serialised on the caller's stream; tn_trace_timings)."""
        arr = (C.c_float * 8)()
        _lib.check(self._lib.tn_trace_timings(self._h, C.byref(arr)))
        return {k: float(arr[i]) for i, k in enumerate(self.TIMING_KEYS)}

    def face_tables(self):
        """(faces [F,3], face_tets [F,2]) int64 CPU tensors of the loaded mesh (debug aid)."""
        F = self._lib.tn_num_faces(self._h)
        faces = _empty((F, 3), dtype=torch.int32)
        ft = _empty((F, 2), dtype=torch.int32)
        _lib.check(self._lib.tn_get_faces(self._h, _ptr(faces), _ptr(ft)))
        return faces, ft

    def build_table(self, which: int):
        """One of the structures load_tetrahedra built, as a uint8 CPU tensor (test aid; see tn_get_build_table)."""
        n = C.c_size_t(0)
        _lib.check(self._lib.tn_get_build_table(self._h, int(which), None, C.byref(n)))
        out = _empty((n.value,), dtype=torch.uint8)
        if n.value:
            _lib.check(self._lib.tn_get_build_table(self._h, int(which), _ptr(out), C.byref(n)))
        return out

    def postprocess_hits(self, hit_count, hit_ids, hit_t, hit_uv, faces=None, face_tets=None):
        """Run only the dedupe/pairing stage on sorted hit rows (test aid); with `faces` [F,3] / `face_tets` [F,2]
        (int32) on those tables instead of the loaded mesh's."""
        R, M = hit_ids.shape
        dev = self._device
        out = {
            "num_visited_cells": _empty((R,), dtype=torch.int32, device=dev),
            "visited_cells": _empty((R, M), dtype=torch.int32, device=dev),
            "barycentric_coordinates": _empty((R, M, 2, 3), dtype=torch.float32, device=dev),
            "hit_distances": _empty((R, M, 2), dtype=torch.float32, device=dev),
            "vertex_indices": _empty((R, M, 4), dtype=torch.int32, device=dev),
        }
        if faces is not None:
            _check(face_tets is not None and faces.dtype == torch.int32 and face_tets.dtype == torch.int32
                   and faces.is_contiguous() and face_tets.is_contiguous(), "faces / face_tets must be contiguous int32")
            _lib.check(self._lib.tn_postprocess_hits_tables(
                dev.index or 0, R, M, _ptr(faces), _ptr(face_tets), _ptr(hit_count), _ptr(hit_ids), _ptr(hit_t),
                _ptr(hit_uv), _ptr(out["num_visited_cells"]), _ptr(out["visited_cells"]),
                _ptr(out["barycentric_coordinates"]), _ptr(out["hit_distances"]), _ptr(out["vertex_indices"]),
                _stream(dev)))
            return out
        _lib.check(self._lib.tn_postprocess_hits(
            self._h, R, M, _ptr(hit_count), _ptr(hit_ids), _ptr(hit_t), _ptr(hit_uv),
            _ptr(out["num_visited_cells"]), _ptr(out["visited_cells"]), _ptr(out["barycentric_coordinates"]),
            _ptr(out["hit_distances"]), _ptr(out["vertex_indices"]), _stream(dev)))
        return out


def fill_rows(visited_cells, barycentric_coordinates, hit_distances, vertex_indices=None, first_slot=0):
    """The constant tails of trace_rays' dense rows (tn_fill_rows) from slot `first_slot` (a multiple of 32: a 128-byte line
    boundary of all four arrays; anything else raises) on, in place, for every row."""
    for x, name in ((visited_cells, "visited_cells"), (barycentric_coordinates, "barycentric_coordinates"),
                    (hit_distances, "hit_distances")) + (() if vertex_indices is None else ((vertex_indices, "vertex_indices"),)):
        _check_input(x, name)
    R, M = visited_cells.shape
    dev = visited_cells.device
    with _on(dev):
        _lib.check(_lib.load().tn_fill_rows(R, M, int(first_slot), _ptr(visited_cells), _ptr(barycentric_coordinates),
                                            _ptr(hit_distances), _ptr(vertex_indices), _stream(dev)))


# Vertex-major shadow copies [V, F] of feature-major fields [F, V] (the checkpoint layout, model.py:269-271).
#
# The gather kernels want one 256-byte row per vertex, the model keeps `tetrahedra_field` feature-major.  By default the
# shadow is made PER CALL (tn_transpose_f32: 23 MB of traffic at V = 45k, ~10 us) -- always correct.  A caller that owns
# the field can opt in to a cached shadow with `register_field(t)`: the copy is then refreshed only when the tensor's
# version counter or storage pointer moves (every in-place optimiser step, copy_ and load_state_dict bump the counter).
# What does NOT bump it are writes through `.data` (the reference model initialises the field that way,
# model.py:336-343,379-386: `self.tetrahedra_field.data[1:4, :] = ...`), so the owner of a registered field must call
# `invalidate_field_cache()` after any such write.  TetraRenderer and the nerfstudio adapter register their field and
# invalidate it from the model's initialisation hook; `TetrahedraTracer.load_tetrahedra` invalidates everything.
# A cached shadow remembers the event that ends its transposition: a consumer on another stream waits for it.
_FIELD_VM = {}        # id(tensor) -> (weakref, version, data_ptr, shadow, event)
# TETRANERF_HIP_CHECK_CACHES=1 (debug): every use of a cached field shadow compares it with the field (a device-wide
# comparison + synchronisation per call: for hunting stale caches, not for production)
_CHECK_CACHES = os.environ.get("TETRANERF_HIP_CHECK_CACHES", "0") == "1"


def register_field(field):
    """Opt in to a cached vertex-major shadow of `field` (see above).  Returns `field`."""
    import weakref

    hit = _FIELD_VM.get(id(field))
    if hit is None or hit[0]() is not field:      # (a dead entry whose id() was reused is replaced)
        key = id(field)

        def _drop(ref, key=key):                  # the field died: its [V,64] shadow (V * 256 bytes of HBM) goes with it
            cur = _FIELD_VM.get(key)
            if cur is not None and cur[0] is ref:
                del _FIELD_VM[key]

        _FIELD_VM[key] = (weakref.ref(field, _drop), None, None, None, None)
    return field


def unregister_field(field):
    _FIELD_VM.pop(id(field), None)


def invalidate_field_cache(field=None):
    """Forget the cached shadow of `field` (all registered fields if None): required after a write through `.data`
    or a raw pointer; the registration itself stays."""
    for k, v in list(_FIELD_VM.items()):
        if v[0]() is None:
            del _FIELD_VM[k]
        elif field is None or v[0]() is field:
            _FIELD_VM[k] = (v[0], None, None, None, None)


def _transpose_field(field):
    Fd, V = field.shape
    ft = _empty((V, Fd), dtype=torch.float32, device=field.device)
    with _on(field.device):
        _lib.check(_lib.load().tn_transpose_f32(Fd, V, _ptr(field), _ptr(ft), _stream(field.device)))
    return ft


def field_vertex_major(field):
    """[V, F] copy of `field` [F, V]: per call, or cached per (tensor object, version, pointer) for registered fields."""
    hit = _FIELD_VM.get(id(field))
    if hit is None or hit[0]() is not field:
        # a detach()-ed alias of a registered field (autograd hands those to Function.backward): same storage, same
        # version counter
        hit = next((v for v in _FIELD_VM.values() if v[0]() is not None and v[2] == field.data_ptr()
                    and v[0]().shape == field.shape and v[0]()._version == field._version), None)
        if hit is None:
            return _transpose_field(field)
        field = hit[0]()
    if hit[1] == field._version and hit[2] == field.data_ptr() and hit[3] is not None:
        cur = torch.cuda.current_stream(field.device)
        cur.wait_event(hit[4])   # no-op on the producing stream; orders a consumer on another stream
        if _CHECK_CACHES and not torch.equal(hit[3], field.detach().t()):
            # (debug aid, TETRANERF_HIP_CHECK_CACHES=1: a write through `.data` / a raw pointer bumps no version counter)
            raise RuntimeError("the cached vertex-major shadow of a registered field is STALE: the field was written through "
                               ".data or a raw pointer without cpp.invalidate_field_cache(field)")
        return hit[3]
    ft = _transpose_field(field)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(field.device))
    _FIELD_VM[id(field)] = (hit[0], field._version, field.data_ptr(), ft, ev)
    return ft


def interpolate_values(vertex_indices, barycentric_coordinates, field):
    """py_interpolate_values (py_binding.cpp:298-330): returns [..., field_dim] as a
    moveaxis(0,-1) view of a contiguous [field_dim, n] buffer."""
    for x, name in ((vertex_indices, "vertex_indices"), (barycentric_coordinates, "barycentric_coordinates"),
                    (field, "field")):
        _check_input(x, name)
    _check(vertex_indices.dtype == torch.int32, "vertex_indices must be a tensor of type int32")
    _check(barycentric_coordinates.dtype == torch.float32, "barycentric_coordinates must be a tensor of type float32")
    _check(barycentric_coordinates.size(-1) + 1 == vertex_indices.size(-1),
           "barycentric_coordinates must have the same last dimension as vertex_indices - 1")
    _check(field.dtype == torch.float32, "field must be a tensor of type float32")
    D = vertex_indices.size(-1)
    _check(D in (2, 3, 4, 6), f"Unsupported interpolation dimension with value {D}")
    n = vertex_indices.numel() // D
    Fd = field.size(0)
    result = _empty((Fd,) + tuple(vertex_indices.shape[:-1]), dtype=field.dtype, device=field.device)
    ft = field_vertex_major(field)
    with _on(field.device):
        _lib.check(_lib.load().tn_interpolate_values_vm(
            D, n, Fd, _ptr(vertex_indices), _ptr(barycentric_coordinates), _ptr(ft), _ptr(result), _stream(field.device)))
    return result.moveaxis(0, -1)


def deterministic_gradients() -> bool:
    """Whether the field gradient (the gather's adjoint) is summed without float atomics: bit-identical from run to run, 3-5x
    the time of the atomic kernel (the twelve weight gradients are bit-reproducible either way).  On when the caller asked
    PyTorch for deterministic algorithms (`torch.use_deterministic_algorithms(True)`) or set DETERMINISTIC_FIELD_GRADIENT."""
    return bool(DETERMINISTIC_FIELD_GRADIENT) or torch.are_deterministic_algorithms_enabled()


DETERMINISTIC_FIELD_GRADIENT = False


def _gather_adjoint_vm(lib, D, V, n, Fd, vi, bc, rows, grad_vm, stream):
    if deterministic_gradients():
        _lib.check(lib.tn_interpolate_values_backward_vm_det(D, V, n, Fd, _ptr(vi), _ptr(bc), _ptr(rows), _ptr(grad_vm), stream))
    else:
        _lib.check(lib.tn_interpolate_values_backward_vm(D, n, Fd, _ptr(vi), _ptr(bc), _ptr(rows), _ptr(grad_vm), stream))


def interpolate_values_backward(vertex_indices, barycentric_coordinates, field, grad_in):
    """py_interpolate_values_backward (py_binding.cpp:341-372)."""
    for x, name in ((vertex_indices, "vertex_indices"), (barycentric_coordinates, "barycentric_coordinates"),
                    (field, "field"), (grad_in, "grad_in")):
        _check_input(x, name)
    _check(vertex_indices.dtype == torch.int32, "vertex_indices must be a tensor of type int32")
    _check(barycentric_coordinates.dtype == torch.float32, "barycentric_coordinates must be a tensor of type float32")
    _check(field.dtype == torch.float32, "field must be a tensor of type float32")
    _check(grad_in.dtype == torch.float32, "grad_in must be a tensor of type float32")
    _check(barycentric_coordinates.size(-1) + 1 == vertex_indices.size(-1),
           "barycentric_coordinates must have the same last dimension as vertex_indices - 1")
    D = vertex_indices.size(-1)
    _check(D in (2, 3, 4, 6), f"Unsupported interpolation dimension with value {D}")
    n = vertex_indices.numel() // D
    Fd, V = field.size(0), field.size(-1)
    _check(grad_in.size(-1) == Fd, "grad_in must have shape [..., field_dim]")
    grad_field_out = _empty((Fd, V), dtype=grad_in.dtype, device=grad_in.device)
    lib = _lib.load()
    if grad_in.moveaxis(-1, 0).is_contiguous() and Fd > 1 and not deterministic_gradients():
        # the reference's layout: a [Fd, n] buffer viewed as [..., Fd] (what py_binding.cpp:369 produces)
        with _on(field.device):
            _lib.check(lib.tn_interpolate_values_backward(D, V, n, Fd, _ptr(vertex_indices), _ptr(barycentric_coordinates),
                                                          _ptr(grad_in.moveaxis(-1, 0)), _ptr(grad_field_out),
                                                          _stream(field.device)))
        return grad_field_out
    # sample-major rows, the usual autograd gradient: consumed as is, accumulated vertex-major, transposed back once
    g = grad_in.contiguous()
    grad_vm = torch.zeros((V, Fd), dtype=torch.float32, device=grad_in.device)
    with _on(field.device):
        _gather_adjoint_vm(lib, D, V, n, Fd, vertex_indices, barycentric_coordinates, g, grad_vm, _stream(field.device))
        _lib.check(lib.tn_transpose_f32(V, Fd, _ptr(grad_vm), _ptr(grad_field_out), _stream(field.device)))
    return grad_field_out


class _MlpWeightsStruct(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w1", "b1", "w2", "b2", "w3", "b3", "wd", "bd", "wh", "bh", "wr", "br")]


_WEIGHT_SHAPES = [(128, 64), (128,), (128, 128), (128,), (128, 128), (128,), (1, 128), (1,), (128, 155), (128,), (3, 128), (3,)]
_MODES = {"fp32": 0, "bf16x3": 1, 0: 0, 1: 1}


def _mode(mode):
    m = _MODES.get(mode)
    _check(m is not None, 'mlp mode must be "fp32" or "bf16x3"')
    return m


class FusedMLP:
    """Handle of the fused MLP kernels (tn_mlp_*): holds the packed forms of ONE set of weights on the device.
    `sync(weights)` packs them when they changed since the last call -- detected through the tensors' identity, version
    counter and storage pointer, i.e. every in-place optimiser step / copy_ / load_state_dict is seen; a write through
    `.data` is not (call `sync(weights, force=True)` or `invalidate()` then).  weights = the 12 tensors
    (w1,b1,w2,b2,w3,b3,wd,bd,wh,bh,wr,br) in nn.Linear layout, contiguous fp32 on the handle's device."""

    def __init__(self, device):
        device = torch.device(device)
        _check(device.type == "cuda", "The device argument must be a CUDA device.")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.tn_mlp_create(int(device.index), C.byref(h)))
        self._h = h
        self._key = None

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.tn_mlp_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def invalidate(self):
        self._key = None

    def sync(self, weights, force=False):
        _check(len(weights) == 12, "weights must hold 12 tensors")
        import weakref

        # identity through weak references: id() of a collected tensor can be reused by a new one at the same address
        key = tuple((w._version, w.data_ptr()) for w in weights)
        if (not force and self._key is not None and key == self._key[0]
                and all(r() is w for r, w in zip(self._key[1], weights))):
            # the packs were made on the stream that was current then: a consumer on another stream waits for them
            torch.cuda.current_stream(self.device).wait_event(self._packed)
            return self
        st = _MlpWeightsStruct()
        keep = []
        for (name, _), w, shp in zip(_MlpWeightsStruct._fields_, weights, _WEIGHT_SHAPES):
            w = w.detach()
            _check_input(w, name)
            _check(w.device == self.device, f"{name} must be on the handle's device")
            _check(w.dtype == torch.float32 and tuple(w.shape) == shp, f"{name} must be f32 {shp}")
            keep.append(w)
            setattr(st, name, w.data_ptr())
        with _on(self.device):
            _lib.check(self._lib.tn_mlp_set_weights(self._h, C.byref(st), _stream(self.device)))
        del keep
        self._packed = torch.cuda.Event()
        self._packed.record(torch.cuda.current_stream(self.device))
        self._key = (key, [weakref.ref(w) for w in weights])
        return self

    @property
    def handle(self):
        return self._h


_DEFAULT_MLP = {}    # id(first weight tensor) -> (weakref to it, FusedMLP): one handle (packed weights + scratch) per MODEL
_MAX_HANDLES = 8     # (each holds ~1.5 MB of packs + 42 MB of gradient scratch once it has been trained with)


def fused_mlp(weights) -> FusedMLP:
    """The handle of this weight set, synchronised with `weights` (re-packed only when they changed).  Handles are keyed
    by the identity of the first weight tensor, so two models alternating on a device each keep their packs (a single
    per-device handle re-packed four weight forms on every switch); a handle dies with its model, the oldest one is
    dropped beyond _MAX_HANDLES.  Single-stream contract per handle: set_weights and the kernels are enqueued on the
    caller's current stream; a call from another stream waits for the packing event (FusedMLP.sync), but two streams
    must not run kernels of ONE handle concurrently (they share its per-call scratch)."""
    import weakref

    w0 = weights[0]
    key = id(w0)
    hit = _DEFAULT_MLP.get(key)
    if hit is None or hit[0]() is not w0:
        def _drop(ref, key=key):
            cur = _DEFAULT_MLP.get(key)
            if cur is not None and cur[0] is ref:
                del _DEFAULT_MLP[key]

        while len(_DEFAULT_MLP) >= _MAX_HANDLES:
            _DEFAULT_MLP.pop(next(iter(_DEFAULT_MLP)))
        hit = _DEFAULT_MLP[key] = (weakref.ref(w0, _drop), FusedMLP(w0.device))
    return hit[1].sync(weights)


def invalidate_weight_cache():
    """Needed only after writing MLP parameters through `.data` / a raw pointer."""
    for _, m in _DEFAULT_MLP.values():
        m.invalidate()


def mlp_forward(feats_fm, dirs, weights, samples_per_ray, mode="fp32"):
    """Fused MFMA forward of mlp_base + density head + mlp_head + rgb head (addition to the
    reference surface; the reference runs these through nerfstudio/PyTorch, model.py:602-621).
    feats_fm f32 [64, n] feature-major (interpolate_values(...).moveaxis(-1, 0) is that buffer),
    dirs f32 [n // samples_per_ray, 3], weights: 12 contiguous fp32 CUDA tensors in nn.Linear layout
    (w1,b1,w2,b2,w3,b3,wd,bd,wh,bh,wr,br).  mode: "fp32" (exact fp32 MFMA chain) or "bf16x3" (bf16 MFMA on 3-way
    split operands; opt-in).  Returns sigma [n], rgb [n,3]."""
    _check_input(feats_fm, "feats")
    _check_input(dirs, "dirs")
    _check(feats_fm.dtype == torch.float32 and feats_fm.dim() == 2 and feats_fm.size(0) == 64, "feats must be f32 [64, n]")
    n = feats_fm.size(1)
    S = int(samples_per_ray)
    _check(S > 0 and n % S == 0, "n must be a multiple of samples_per_ray")
    _check(dirs.dtype == torch.float32 and tuple(dirs.shape) == (n // S, 3), "dirs must be f32 [n/samples_per_ray, 3]")
    m = fused_mlp(weights)
    dev = feats_fm.device
    sigma = _empty((n,), dtype=torch.float32, device=dev)
    rgb = _empty((n, 3), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.load().tn_mlp_forward(m.handle, n, S, _ptr(feats_fm), _ptr(dirs), _mode(mode), _ptr(sigma), _ptr(rgb),
                                              _stream(dev)))
    return sigma, rgb


def _ray_bias(ray_head_bias, rays, dev):
    """f32 [rays, 128] per-ray bias of the head layer (appearance embedding: include/tetranerf_hip.h) or None"""
    if ray_head_bias is None:
        return None
    _check_input(ray_head_bias, "ray_head_bias")
    _check(ray_head_bias.dtype == torch.float32 and tuple(ray_head_bias.shape) == (rays, 128) and ray_head_bias.device == dev,
           "ray_head_bias must be f32 [rays, 128] on the field's device")
    return ray_head_bias


def mlp_forward_gather(vertex_indices, barycentric_coordinates, field, dirs, weights, samples_per_ray, mode="fp32",
                       ray_head_bias=None, count=None):
    """interpolate_values + mlp_forward in ONE kernel: the wave gathers its samples' features from the
    (vertex-major shadow of the) field straight into MFMA operand registers; the [64, n] feature buffer is never
    written.  vertex_indices i32 [..., 4], barycentric_coordinates f32 [..., 3], field f32 [64, V].
    dirs=None: density only (the coarse pass of the model, model.py:577-581) -> sigma [n]."""
    density_only = dirs is None
    for x, name in ((vertex_indices, "vertex_indices"), (barycentric_coordinates, "barycentric_coordinates"),
                    (field, "field")) + (() if density_only else ((dirs, "dirs"),)):
        _check_input(x, name)
    _check(vertex_indices.dtype == torch.int32 and vertex_indices.size(-1) == 4, "vertex_indices must be i32 [...,4]")
    _check(barycentric_coordinates.dtype == torch.float32 and barycentric_coordinates.size(-1) == 3,
           "barycentric_coordinates must be f32 [...,3]")
    _check(field.dtype == torch.float32 and field.dim() == 2 and field.size(0) == 64, "field must be f32 [64, V]")
    n = vertex_indices.numel() // 4
    S = int(samples_per_ray)
    _check(S > 0 and n % S == 0, "n must be a multiple of samples_per_ray")
    _check(density_only or (dirs.dtype == torch.float32 and tuple(dirs.shape) == (n // S, 3)),
           "dirs must be f32 [n/samples_per_ray, 3]")
    m = fused_mlp(weights)
    dev = field.device
    field_vm = field_vertex_major(field)
    sigma = _empty((n,), dtype=torch.float32, device=dev)
    rgb = None if density_only else _empty((n, 3), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.load().tn_mlp_forward_gather(m.handle, n, S, _ptr(vertex_indices), _ptr(barycentric_coordinates),
                                                     _ptr(field_vm), _ptr(dirs), _mode(mode), _ptr(sigma), _ptr(rgb),
                                                     _ptr(None if density_only else _ray_bias(ray_head_bias, n // S, dev)),
                                                     _ptr(count), _stream(dev)))
    return sigma if density_only else (sigma, rgb)


class _RgbBackground(C.Structure):   # tn_rgb_background
    _fields_ = [("r", C.c_float), ("g", C.c_float), ("b", C.c_float), ("clamp", C.c_int)]


def _background(background, clamp=False):
    """tn_rgb_background from a grey level (float) or an (r, g, b) triple (sequence or HOST tensor; a device tensor would
    force a sync here) + the RGB renderer's evaluation-mode flag."""
    if isinstance(background, torch.Tensor):
        background = background.detach().reshape(-1).tolist()
    if isinstance(background, (int, float)):
        background = (background,) * 3
    r, g, b = (float(x) for x in background)
    return C.byref(_RgbBackground(r, g, b, 1 if clamp else 0))


def render_rays(trace_lists, order, count, field, directions, weights, num_samples, num_fine=0, biased=False, out=None,
                background=1.0, clamp=True, ray_head_bias=None, histogram_padding=0.01, eps=1e-5, mode="fp32"):
    """Everything between trace_rays and the frame as ONE persistent launch (tn_render_rays_ex; mode "fp32" | "bf16x3"): coarse sampler -> match -> gather
    + MLP -> weights -> PDF sampler -> match -> gather + MLP + heads -> composite, for the hitting rays order[:count] (compact_hits)
    whose number stays on the device.  trace_lists as returned by trace_rays; directions f32 [R,3] and ray_head_bias f32 [R,128]
    over ALL rays; out = (rgb [R,3], accumulation [R,1] or [R], depth) pre-filled with the background values."""
    nv, _cells, bary, dist, verts = trace_lists
    for x, name in ((nv, "num_visited_cells"), (bary, "barycentric_coordinates"), (dist, "hit_distances"), (verts, "vertex_indices"),
                    (order, "order"), (field, "field"), (directions, "directions")):
        _check_input(x, name)
    R, M = dist.size(0), dist.size(1)
    _check(order.dtype == torch.int32 and order.dim() == 1 and order.numel() <= R, "order must be i32 [<= R]")
    _check(count is None or (count.dtype == torch.int32 and count.numel() >= 1 and count.is_cuda), "count must be an i32 device tensor")
    _check(directions.dtype == torch.float32 and tuple(directions.shape) == (R, 3), "directions must be f32 [R, 3]")
    _check(field.dtype == torch.float32 and field.dim() == 2 and field.size(0) == 64, "field must be f32 [64, V]")
    S, Sf = int(num_samples), int(num_fine)
    m = fused_mlp(weights)
    dev = field.device
    field_vm = field_vertex_major(field)
    rgb, acc, depth = out
    for x, name in ((rgb, "rgb"), (acc, "accumulation"), (depth, "depth")):
        _check_input(x, name)
        _check(x.dtype == torch.float32 and x.size(0) == R, f"{name} must be f32 over all rays")
    hb = None
    if ray_head_bias is not None:
        hb = _ray_bias(ray_head_bias, R, dev)
    with _on(dev):
        _lib.check(_lib.load().tn_render_rays_ex(
            m.handle, M, _ptr(nv), _ptr(dist), _ptr(bary), _ptr(verts), _ptr(order), _ptr(count), order.numel(), S, Sf, 1 if biased else 0,
            _ptr(_linspace_table(S, dev)), _ptr(_quantile_table(Sf + 1, True, dev)) if Sf else None, float(histogram_padding), float(eps),
            _ptr(field_vm), _ptr(directions), _background(background, clamp), _ptr(rgb), _ptr(acc), _ptr(depth), _ptr(hb), _mode(mode), _stream(dev)))


_TABLES = {}    # (kind, n, device) -> small constant tables of the samplers (the values the PyTorch statements use)


def _linspace_table(S, device):
    key = ("lin", S, device)
    if key not in _TABLES:
        _TABLES[key] = torch.linspace(0.0, 1.0, S + 1, dtype=torch.float32, device=device)
    return _TABLES[key]


def _quantile_table(num_bins, centred, device):
    key = ("q", num_bins, centred, device)
    if key not in _TABLES:
        u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins, dtype=torch.float32, device=device)
        _TABLES[key] = (u + 1.0 / (2 * num_bins)).contiguous() if centred else u
    return _TABLES[key]


def compact_hits(num_visited_cells, want_padded=False):
    """Stable partition of the rays of a trace call by num_visited_cells > 0, on the device (tn_compact_hits; replaces
    torch.nonzero / boolean indexing: no host synchronisation).  Returns (order i32 [R], count i32 [1] on the device[,
    padded i32 [R]]): order[:count] = the hitting rays in ray order, order[count:] = the others; padded = order with the tail
    replaced by order[0]."""
    _check_input(num_visited_cells, "num_visited_cells")
    _check(num_visited_cells.dtype == torch.int32 and num_visited_cells.dim() == 1, "num_visited_cells must be i32 [R]")
    R, dev = num_visited_cells.numel(), num_visited_cells.device
    order = _empty((R,), dtype=torch.int32, device=dev)
    count = _empty((1,), dtype=torch.int32, device=dev)
    padded = _empty((R,), dtype=torch.int32, device=dev) if want_padded else None
    n_scratch = 2 * ((R + 2047) // 2048)
    scratch = _empty((max(n_scratch, 1),), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(_lib.load().tn_compact_hits(R, _ptr(num_visited_cells), _ptr(order), _ptr(count), _ptr(padded), _ptr(scratch),
                                               scratch.numel(), _stream(dev)))
    return (order, count, padded) if want_padded else (order, count)


def sample_coarse(num_visited_cells, hit_distances, ray_index, num_samples, biased=False, t_rand=None, count=None):
    """Coarse sampler as ONE kernel on the trace rows in place (tn_sample_coarse; model.py:531-557, 111-192): returns
    (edges f32 [r, S+1] euclidean bin edges, near_far f32 [r, 2]) for the hitting rays ray_index i32 [r].  t_rand
    [r, S+1]: training-mode stratified bins; biased: the TetrahedraSampler mapping.  Same values as
    render.uniform_sample_bins / biased_sample_bins up to the rounding of one prefix sum."""
    for x, name in ((num_visited_cells, "num_visited_cells"), (hit_distances, "hit_distances"), (ray_index, "ray_index")):
        _check_input(x, name)
    _check(ray_index.dtype == torch.int32 and ray_index.dim() == 1, "ray_index must be i32 [r]")
    _check(hit_distances.dtype == torch.float32 and hit_distances.dim() == 3 and hit_distances.size(2) == 2, "hit_distances must be f32 [R,M,2]")
    r, S, M = ray_index.numel(), int(num_samples), hit_distances.size(1)
    dev = hit_distances.device
    if t_rand is not None:
        _check_input(t_rand, "t_rand")
        _check(t_rand.dtype == torch.float32 and tuple(t_rand.shape) == (r, S + 1), "t_rand must be f32 [r, S+1]")
    edges = _empty((r, S + 1), dtype=torch.float32, device=dev)
    near_far = _empty((r, 2), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.load().tn_sample_coarse(r, S, M, _ptr(ray_index), _ptr(num_visited_cells), _ptr(hit_distances),
                                                _ptr(_linspace_table(S, dev)), _ptr(t_rand), 1 if biased else 0, _ptr(edges),
                                                _ptr(near_far), _ptr(count), _stream(dev)))
    return edges, near_far


def sample_pdf(edges, weights, near_far, num_fine, u_rand=None, histogram_padding=0.01, eps=1e-5, count=None):
    """nerfstudio's PDFSampler (include_original) as ONE kernel (tn_sample_pdf; model.py:582-586): edges f32 [r, S+1]
    euclidean coarse edges, weights f32 [r, S], near_far f32 [r, 2] -> f32 [r, S + num_fine + 2] merged, sorted euclidean
    edges.  u_rand [r, num_fine+1]: training-mode stratified quantiles."""
    for x, name in ((edges, "edges"), (weights, "weights"), (near_far, "near_far")):
        _check_input(x, name)
        _check(x.dtype == torch.float32, f"{name} must have float32 type")
    r, S = weights.shape
    nb = int(num_fine) + 1
    _check(tuple(edges.shape) == (r, S + 1) and tuple(near_far.shape) == (r, 2), "shape mismatch")
    dev = edges.device
    if u_rand is not None:
        _check_input(u_rand, "u_rand")
        _check(u_rand.dtype == torch.float32 and tuple(u_rand.shape) == (r, nb), "u_rand must be f32 [r, num_fine+1]")
    out = _empty((r, S + 1 + nb), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.load().tn_sample_pdf(r, S, int(num_fine), _ptr(edges), _ptr(weights), _ptr(near_far),
                                             _ptr(_quantile_table(nb, u_rand is None, dev)), _ptr(u_rand), float(histogram_padding),
                                             float(eps), _ptr(out), _ptr(count), _stream(dev)))
    return out


def composite(sigma, rgb, edges, background=1.0, return_weights=False, clamp=False, out=None, ray_index=None, count=None):
    """RaySamples.get_weights + RGB (background blend) / accumulation / median-depth renderers
    (model.py:632-638) in one kernel.  sigma f32 [R,S], rgb f32 [R,S,3], edges f32 [R,S+1].
    rgb=None: only the weights [R,S] are computed and returned (get_weights of the coarse pass, model.py:582).
    out = (rgb [R_all,3], accumulation, depth over ALL rays of the trace call) + ray_index i32 [R]: row q is written at
    ray_index[q] (returns None); count i32 [1] on the device: only the first count[0] rows are processed (compact_hits)."""
    for x, name in ((sigma, "sigma"), (edges, "edges")) + (() if rgb is None else ((rgb, "rgb"),)):
        _check_input(x, name)
        _check(x.dtype == torch.float32, f"{name} must have float32 type")
    R, S = sigma.shape
    _check((rgb is None or tuple(rgb.shape) == (R, S, 3)) and tuple(edges.shape) == (R, S + 1), "shape mismatch")
    dev = sigma.device
    if rgb is None:
        weights = _empty((R, S), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.load().tn_composite(R, S, _ptr(sigma), None, _ptr(edges), None, None, None, None,
                                                _ptr(weights), None, _ptr(count), _stream(dev)))
        return weights
    if out is not None:
        _check(ray_index is not None and ray_index.dtype == torch.int32 and ray_index.numel() == R, "out= needs ray_index i32 [R]")
        o_rgb, o_acc, o_depth = out
        for x, name in ((o_rgb, "rgb"), (o_acc, "accumulation"), (o_depth, "depth")):
            _check_input(x, name)
            _check(x.dtype == torch.float32, f"{name} must have float32 type")
        with _on(dev):
            _lib.check(_lib.load().tn_composite(R, S, _ptr(sigma), _ptr(rgb), _ptr(edges), _background(background, clamp), _ptr(o_rgb),
                                                _ptr(o_acc), _ptr(o_depth), None, _ptr(ray_index), _ptr(count), _stream(dev)))
        return None
    out_rgb = _empty((R, 3), dtype=torch.float32, device=dev)
    acc = _empty((R, 1), dtype=torch.float32, device=dev)
    depth = _empty((R, 1), dtype=torch.float32, device=dev)
    weights = _empty((R, S), dtype=torch.float32, device=dev) if return_weights else None
    with _on(dev):
        _lib.check(_lib.load().tn_composite(R, S, _ptr(sigma), _ptr(rgb), _ptr(edges), _background(background, clamp), _ptr(out_rgb),
                                            _ptr(acc), _ptr(depth), _ptr(weights), None, _ptr(count), _stream(dev)))
    return (out_rgb, acc, depth, weights) if return_weights else (out_rgb, acc, depth)


class _MlpBackwardBuffers(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("x0", "h1", "h2", "h3", "h4", "masks", "d1", "d2", "d3", "d4", "dhead", "dx0")]


class MlpSaved:
    """What mlp_forward_gather_train leaves for mlp_backward: the layer inputs x0 [64,n], h1..h4 [128,n] (quad-major
    [F/4][n][4], opaque to the caller: the operands of the weight-gradient GEMMs) and the ReLU masks [4,n,2] (all the dX kernel needs).  (`sigma` / `rgb` are the
    forward's outputs as returned; an autograd node must hold them through save_for_backward, not through this object.)"""
    __slots__ = ("acts", "masks", "sigma", "rgb", "n", "S")


def mlp_forward_gather_train(vertex_indices, barycentric_coordinates, field, dirs, weights, samples_per_ray, ray_head_bias=None):
    """mlp_forward_gather (fp32) for training: returns (sigma [n], rgb [n,3], saved) -- `saved` holds 2.3 KB per sample for
    mlp_backward, which then recomputes nothing."""
    for x, name in ((vertex_indices, "vertex_indices"), (barycentric_coordinates, "barycentric_coordinates"),
                    (field, "field"), (dirs, "dirs")):
        _check_input(x, name)
    _check(vertex_indices.dtype == torch.int32 and vertex_indices.size(-1) == 4, "vertex_indices must be i32 [...,4]")
    _check(barycentric_coordinates.dtype == torch.float32 and barycentric_coordinates.size(-1) == 3,
           "barycentric_coordinates must be f32 [...,3]")
    _check(field.dtype == torch.float32 and field.dim() == 2 and field.size(0) == 64, "field must be f32 [64, V]")
    n = vertex_indices.numel() // 4
    S = int(samples_per_ray)
    _check(S > 0 and n % S == 0, "n must be a multiple of samples_per_ray")
    _check(dirs.dtype == torch.float32 and tuple(dirs.shape) == (n // S, 3), "dirs must be f32 [n/samples_per_ray, 3]")
    m = fused_mlp(weights)
    dev = field.device
    field_vm = field_vertex_major(field)
    sv = MlpSaved()
    sv.n, sv.S = n, S
    sv.sigma = _empty((n,), dtype=torch.float32, device=dev)
    sv.rgb = _empty((n, 3), dtype=torch.float32, device=dev)
    sv.acts = _empty((64 + 4 * 128, n), dtype=torch.float32, device=dev)
    sv.masks = _empty((4, n, 2), dtype=torch.int64, device=dev)
    a = sv.acts
    bs = _MlpBackwardBuffers(a[0:64].data_ptr(), a[64:192].data_ptr(), a[192:320].data_ptr(), a[320:448].data_ptr(),
                             a[448:576].data_ptr(), sv.masks.data_ptr(), None, None, None, None, None, None)
    with _on(dev):
        _lib.check(_lib.load().tn_mlp_forward_gather_train(m.handle, n, S, _ptr(vertex_indices), _ptr(barycentric_coordinates),
                                                           _ptr(field_vm), _ptr(dirs.contiguous()), _ptr(sv.sigma), _ptr(sv.rgb),
                                                           C.byref(bs), _ptr(_ray_bias(ray_head_bias, n // S, dev)), _stream(dev)))
    return sv.sigma, sv.rgb, sv


def mlp_backward(saved, vertex_indices, barycentric_coordinates, field, dirs, weights, sigma, rgb, d_sigma, d_rgb,
                 want_ray_head_grad=False):
    """Adjoint of mlp_forward_gather_train (addition; the reference leaves this to PyTorch autograd, model.py:602-630):
    given the forward's outputs sigma [n] / rgb [n,3], dL/dsigma [n] and dL/drgb [n,3] returns (grad_field [64,V], [12 weight
    gradients in the order of `weights`]).
    tn_mlp_backward -- the dX chain on the fp32 matrix cores from the saved ReLU masks, nothing recomputed -- then
    tn_mlp_param_grads -- the twelve parameter gradients as sample-streaming fp32-MFMA GEMMs over the saved layer inputs,
    summed without atomics (bit-reproducible) -- then the gather's adjoint.  2.1 KB of gradient buffers per sample."""
    mh = fused_mlp(weights)
    keep = [w.detach() for w in weights]
    n, S = saved.n, saved.S
    _check(vertex_indices.numel() == 4 * n, "vertex_indices do not belong to the saved forward pass")
    _check(d_sigma.numel() == n and d_rgb.numel() == 3 * n, "d_sigma / d_rgb must have n / 3n elements")
    dev = field.device
    V = field.size(1)
    vi = vertex_indices.reshape(n, 4)
    bc = barycentric_coordinates.reshape(n, 3)
    d_sigma = d_sigma.reshape(n).contiguous().float()
    d_rgb = d_rgb.reshape(n, 3).contiguous().float()
    dirs = dirs.contiguous()
    lib = _lib.load()
    # the twelve gradients as views of ONE zero-filled buffer (tn_mlp_param_grads accumulates): one fill launch, not twelve
    sizes = [w.numel() for w in keep]
    offs = [0]
    for k in sizes:
        offs.append(offs[-1] + (k + 3) // 4 * 4)       # 16-byte aligned pieces
    flat = torch.zeros((offs[-1],), dtype=torch.float32, device=dev)
    grads = [flat[o:o + k].view(tuple(w.shape)) for o, k, w in zip(offs, sizes, keep)]
    gs = _MlpWeightsStruct(*[g.data_ptr() for g in grads])
    grad_vm = torch.zeros((V, 64), dtype=torch.float32, device=dev)
    a = saved.acts
    buf = _empty((4 * 128 + 4, n), dtype=torch.float32, device=dev)
    rows = _empty((n, 64), dtype=torch.float32, device=dev)     # d x0, sample-major
    bs = _MlpBackwardBuffers(a[0:64].data_ptr(), a[64:192].data_ptr(), a[192:320].data_ptr(), a[320:448].data_ptr(),
                             a[448:576].data_ptr(), saved.masks.data_ptr(), buf[0:128].data_ptr(), buf[128:256].data_ptr(),
                             buf[256:384].data_ptr(), buf[384:512].data_ptr(), buf[512:516].data_ptr(), rows.data_ptr())
    stream = _stream(dev)
    with _on(dev):
        _lib.check(lib.tn_mlp_backward(mh.handle, n, _ptr(sigma.contiguous()), _ptr(rgb.contiguous()), _ptr(d_sigma), _ptr(d_rgb),
                                       C.byref(bs), stream))
        _lib.check(lib.tn_mlp_param_grads(mh.handle, n, S, _ptr(dirs), C.byref(bs), C.byref(gs), stream))
        d_ray_bias = None
        if want_ray_head_grad:      # gradient of the per-ray head bias: per-ray sums of d4
            d_ray_bias = _empty((n // S, 128), dtype=torch.float32, device=dev)
            _lib.check(lib.tn_mlp_ray_head_grad(n, S, C.byref(bs), _ptr(d_ray_bias), stream))
        # gradient of the gathered features -> field (vertex-major accumulation)
        _gather_adjoint_vm(lib, 4, V, n, 64, vi, bc, rows, grad_vm, stream)
        grad_field = _empty((64, V), dtype=torch.float32, device=dev)
        _lib.check(lib.tn_transpose_f32(V, 64, _ptr(grad_vm), _ptr(grad_field), stream))
    if want_ray_head_grad:
        return grad_field, grads, d_ray_bias
    return grad_field, grads


def composite_backward(sigma, rgb, edges, d_out_rgb, d_out_acc, background=1.0):
    """Adjoint of composite() w.r.t. sigma [R,S] and rgb [R,S,3] (the median depth has no gradient)."""
    R, S = sigma.shape
    dev = sigma.device
    d_sigma = _empty((R, S), dtype=torch.float32, device=dev)
    d_rgb = _empty((R, S, 3), dtype=torch.float32, device=dev)
    g_rgb = None if d_out_rgb is None else d_out_rgb.contiguous().float()
    g_acc = None if d_out_acc is None else d_out_acc.contiguous().float()
    with _on(dev):
        _lib.check(_lib.load().tn_composite_backward(R, S, _ptr(sigma.contiguous()), _ptr(rgb.contiguous()), _ptr(edges.contiguous()),
                                                     _background(background), _ptr(g_rgb), _ptr(g_acc), _ptr(d_sigma), _ptr(d_rgb),
                                                     _stream(dev)))
    return d_sigma, d_rgb


def triangulate(points):
    """py_triangulate (py_binding.cpp:239-256).  The reference runs CGAL's Delaunay on the CPU
    (src/triangulation.cpp:34-75); CGAL is not part of this build, Qhull (scipy) stands in.
    Offline preprocessing, not on the hot path."""
    _check(points.dim() == 2 and points.size(1) == 3, "points must have shape [num_points, 3]")
    from .scenes import delaunay_cells

    cells = delaunay_cells(points.detach().cpu().contiguous().numpy())
    return torch.from_numpy(cells).to(points.device)


def find_average_spacing(points):
    """py_find_average_spacing (py_binding.cpp:229-237): mean distance to the 6 nearest
    neighbours (CGAL::compute_average_spacing<6>, src/triangulation.cpp:121-134)."""
    _check(points.is_contiguous(), "points must be contiguous")
    _check(points.device.type == "cpu", "points must be a CPU tensor")
    _check(points.dim() == 2 and points.size(1) == 3, "points must have shape [num_points, 3]")
    from scipy.spatial import cKDTree

    p = points.numpy()
    d, _ = cKDTree(p).query(p, k=7)
    return float(d[:, 1:].mean())


def gather_uint32(self, dim, index):
    """py_gather_uint32 (py_binding.cpp:374-399); not called by the model."""
    _check(index.dtype == torch.int32, "index must be a tensor of type int32")
    _check(self.is_floating_point(), "self must be a tensor of a floating-point type")
    _check(self.device == index.device, "self and index must be on the same device")
    _check(self.is_contiguous(), "self must be contiguous")
    _check(index.is_contiguous(), "index must be contiguous")
    _check(self.device.type == "cuda", "self must be on CUDA")
    _check(index.device.type == "cuda", "index must be on CUDA")
    _check(self.dim() == 1, "self must be 1-dimensional")
    _check(index.dim() == self.dim(), "self and index must have the same number of dimensions")
    _check(dim == 0, "dim must be 0")
    _check(self.dtype in (torch.float32, torch.float64), "self must be float32 or float64")
    result = _empty(index.shape, dtype=self.dtype, device=self.device)
    with _on(self.device):
        _lib.check(_lib.load().tn_gather_uint32(self.element_size(), self.numel(), index.numel(), _ptr(index), _ptr(self),
                                                _ptr(result), _stream(self.device)))
    return result


def scatter_ema_uint32(self, dim, index, decay, values):
    """py_scatter_ema_uint32 (py_binding.cpp:405-431): in-place x[idx] = x[idx]*decay + (1-decay)*v."""
    _check(dim == 0, "dim must be 0")
    _check(self.is_floating_point(), "self must be a tensor of a floating-point type")
    _check(self.is_contiguous(), "self must be contiguous")
    _check(self.dim() == 1, "self must be 1-dimensional")
    _check(self.device.type == "cuda", "self must be on CUDA")
    _check(index.dtype == torch.int32, "index must be a tensor of type int32")
    _check(index.is_contiguous(), "index must be contiguous")
    _check(index.device.type == "cuda", "index must be on CUDA")
    _check(values.is_contiguous(), "values must be contiguous")
    _check(self.device == index.device, "self and index must be on the same device")
    _check(values.device == index.device, "values and index must be on the same device")
    _check(values.dtype == self.dtype, "values and self must have the same dtype")
    _check(index.dim() == self.dim(), "self and index must have the same number of dimensions")
    _check(values.shape == index.shape, "values and index must have the same shape")
    _check(self.dtype in (torch.float32, torch.float64), "self must be float32 or float64")
    with _on(self.device):
        _lib.check(_lib.load().tn_scatter_ema_uint32(self.element_size(), self.numel(), index.numel(), _ptr(index),
                                                     float(decay), _ptr(values), _ptr(self), _stream(self.device)))
