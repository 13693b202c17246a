"""ctypes binding of libtetranerf_hip.so (the C-ABI declared in include/tetranerf_hip.h).

The library is the product: if it is missing or fails to load, everything here fails
loudly -- there is no CPU fallback and no route through oracle/.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

_HERE = Path(__file__).resolve().parent
# TETRANERF_HIP_LIB: another build of the same library (profiling: interleaved A/B runs of two builds on one box)
LIB_PATH = Path(os.environ["TETRANERF_HIP_LIB"]).resolve() if os.environ.get("TETRANERF_HIP_LIB") else _HERE / "libtetranerf_hip.so"

# every symbol include/tetranerf_hip.h declares
SYMBOLS = (
    "tn_last_error", "tn_version", "tn_abi_version", "tn_tracer_create", "tn_tracer_destroy", "tn_load_tetrahedra",
    "tn_num_faces", "tn_get_faces", "tn_get_build_table", "tn_trace_rays", "tn_trace_rays_ex", "tn_trace_rays_triangles", "tn_find_tetrahedra",
    "tn_find_matched_cells", "tn_find_matched_cells_indexed",
    "tn_interpolate_values", "tn_interpolate_values_backward", "tn_interpolate_values_backward_rows",
    "tn_transpose_f32", "tn_interpolate_values_vm", "tn_interpolate_values_backward_vm", "tn_interpolate_values_backward_vm_det",
    "tn_postprocess_hits", "tn_postprocess_hits_tables",
    "tn_trace_stats", "tn_trace_flag_reasons", "tn_set_option", "tn_mlp_create", "tn_mlp_destroy", "tn_mlp_set_weights",
    "tn_mlp_forward", "tn_mlp_forward_gather", "tn_composite", "tn_gather_uint32", "tn_scatter_ema_uint32",
    "tn_mlp_forward_gather_train", "tn_mlp_backward", "tn_mlp_ray_head_grad", "tn_mlp_param_grads", "tn_composite_backward", "tn_sample_coarse", "tn_sample_pdf",
    "tn_trace_timings", "tn_trace_cross_check", "tn_fill_rows", "tn_compact_hits", "tn_render_rays", "tn_render_rays_ex",
)

ABI_VERSION = 6          # include/tetranerf_hip.h: TN_ABI_VERSION this binding was written against

_lib = None


def load():
    """dlopen the HIP library (once) and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"ERROR: {LIB_PATH} is missing. Build it first: python __graft_entry__.py "
            "(or make -C tetra-nerf_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    lib.tn_last_error.restype = C.c_char_p
    lib.tn_last_error.argtypes = []
    lib.tn_version.restype = C.c_char_p
    lib.tn_version.argtypes = []
    if not hasattr(lib, "tn_abi_version") or lib.tn_abi_version() != ABI_VERSION:
        got = lib.tn_abi_version() if hasattr(lib, "tn_abi_version") else "none (a build older than ABI 6)"
        raise RuntimeError(f"ERROR: {LIB_PATH} exports ABI {got}, this binding needs ABI {ABI_VERSION} "
                           "(include/tetranerf_hip.h: TN_ABI_VERSION). Rebuild it: python __graft_entry__.py")
    lib.tn_abi_version.argtypes = []
    lib.tn_tracer_create.argtypes = [i32, C.POINTER(vp)]
    lib.tn_tracer_destroy.argtypes = [vp]
    lib.tn_load_tetrahedra.argtypes = [vp, sz, sz, vp, vp, vp]
    lib.tn_num_faces.restype = sz
    lib.tn_num_faces.argtypes = [vp]
    lib.tn_get_faces.argtypes = [vp, vp, vp]
    lib.tn_get_build_table.argtypes = [vp, i32, vp, C.POINTER(sz)]
    lib.tn_trace_rays.argtypes = [vp, sz, u32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tn_trace_rays_ex.argtypes = [vp, sz, u32, vp, vp, vp, vp, vp, vp, vp, u32, vp]
    lib.tn_trace_rays_triangles.argtypes = [vp, sz, u32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tn_find_tetrahedra.argtypes = [vp, sz, vp, vp, vp, vp, vp]
    lib.tn_find_matched_cells.argtypes = [sz, sz, sz] + [vp] * 11
    lib.tn_find_matched_cells_indexed.argtypes = [sz, sz, sz] + [vp] * 13
    lib.tn_interpolate_values.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, vp]
    lib.tn_interpolate_values_backward.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, vp]
    lib.tn_interpolate_values_backward_rows.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, vp]
    lib.tn_transpose_f32.argtypes = [u32, u32, vp, vp, vp]
    lib.tn_interpolate_values_vm.argtypes = [u32, u32, u32, vp, vp, vp, vp, vp]
    lib.tn_interpolate_values_backward_vm.argtypes = [u32, u32, u32, vp, vp, vp, vp, vp]
    lib.tn_interpolate_values_backward_vm_det.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, vp]
    lib.tn_postprocess_hits.argtypes = [vp, sz, u32] + [vp] * 10
    lib.tn_postprocess_hits_tables.argtypes = [i32, sz, u32] + [vp] * 12
    lib.tn_trace_stats.argtypes = [vp, C.POINTER(C.c_uint64 * 4)]
    lib.tn_trace_flag_reasons.argtypes = [vp, C.POINTER(C.c_uint64 * 16)]
    lib.tn_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.tn_trace_timings.argtypes = [vp, C.POINTER(C.c_float * 8)]
    lib.tn_trace_cross_check.argtypes = [vp, C.POINTER(C.c_uint64 * 8)]
    lib.tn_fill_rows.argtypes = [sz, u32, u32, vp, vp, vp, vp, vp]
    lib.tn_mlp_create.argtypes = [i32, C.POINTER(vp)]
    lib.tn_mlp_destroy.argtypes = [vp]
    lib.tn_mlp_set_weights.argtypes = [vp, vp, vp]
    lib.tn_mlp_forward.argtypes = [vp, sz, u32, vp, vp, i32, vp, vp, vp]
    lib.tn_mlp_forward_gather.argtypes = [vp, sz, u32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.tn_gather_uint32.argtypes = [i32, u32, u32, vp, vp, vp, vp]
    lib.tn_scatter_ema_uint32.argtypes = [i32, u32, u32, vp, C.c_double, vp, vp, vp]
    lib.tn_composite.argtypes = [sz, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tn_mlp_forward_gather_train.argtypes = [vp, sz, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tn_mlp_ray_head_grad.argtypes = [sz, u32, vp, vp, vp]
    lib.tn_mlp_backward.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
    lib.tn_mlp_param_grads.argtypes = [vp, sz, u32, vp, vp, vp, vp]
    lib.tn_composite_backward.argtypes = [sz, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tn_sample_coarse.argtypes = [sz, u32, u32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.tn_sample_pdf.argtypes = [sz, u32, u32, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]
    lib.tn_compact_hits.argtypes = [sz, vp, vp, vp, vp, vp, sz, vp]
    lib.tn_render_rays.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, sz, u32, u32, i32, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tn_render_rays_ex.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, sz, u32, u32, i32, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("tn_last_error", "tn_version", "tn_abi_version", "tn_num_faces"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int):
    """Turn a non-zero return code into the RuntimeError the reference would raise."""
    if rc != 0:
        msg = load().tn_last_error().decode("utf-8", "replace")
        raise RuntimeError(msg or f"libtetranerf_hip call failed (rc={rc})")
