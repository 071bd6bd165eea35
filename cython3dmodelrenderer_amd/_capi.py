"""ctypes binding of libcrender_hip.so (C ABI: include/crender_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing
a symbol from here raises.  Device pointers are plain integers (``tensor.data_ptr()``).
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

ABI_VERSION = 6
OK, EINVAL, EHIP, ENOMEM, EBUSY, ESTATE = 0, 1, 2, 3, 4, 5
FUSED_CLEAR = 1
NO_DIRECT_BINS = 2
OVERLAPPED_FRAMES = 4
FUSED_GURO = 8
STATIC_INPUTS = 16
# raster kernels of a 32-pixel plan (crender_plan_set_raster_path): both exact on every tile
PATH_AUTO, PATH_GENERAL, PATH_OWNERS = -1, 0, 1

_vp, _i32, _i64, _u32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint, C.c_size_t
_f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes); mirrors include/crender_hip.h one to one
SIGNATURES = {
    "crender_abi_version": (_i32, []),
    "crender_last_error": (C.c_char_p, []),
    "crender_projection_matrix": (_i32, [C.c_double, C.c_double, C.c_double, _i32, _i32, _f32p]),
    "crender_project": (_i32, [_vp, _vp, _i64, _f32p, _i32, _i32, _vp]),
    "crender_clear": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "crender_plan_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i64, _i64, _i32]),
    "crender_plan_create": (_i32, [C.POINTER(_vp), _i32, _i32, _i32, _i32, _i64, _i64, _i32,
                                   _vp, _sz, _vp]),
    "crender_plan_destroy": (None, [_vp]),
    "crender_plan_last_bin_usage": (_i32, [_vp, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "crender_plan_last_frame_direct": (_i32, [_vp]),
    "crender_plan_last_frame_binning": (_i32, [_vp]),
    "crender_plan_frame_ticket": (C.c_uint64, [_vp]),
    "crender_plan_poll_bin_usage": (_i32, [_vp, C.c_uint64, C.POINTER(_i64), C.POINTER(_i64)]),
    "crender_plan_set_light": (_i32, [_vp, _f32p]),
    "crender_plan_debug_check": (_i32, [_vp, _vp, C.c_char_p, _sz]),
    "crender_plan_set_raster_path": (_i32, [_vp, _i32]),
    "crender_plan_last_raster_path": (_i32, [_vp]),
    "crender_set_default_raster_path": (_i32, [_i32]),
    "crender_plan_set_triangle_order": (_i32, [_vp, _vp, _vp]),
    "crender_plan_set_normal_z": (_i32, [_vp, _vp]),
    "crender_tile_order_keys": (_i32, [_vp, _i64, _f32p, _i32, _i32, _vp, _vp]),
    "crender_plan_timing_begin": (_i32, [_vp, _i32]),
    "crender_plan_timing_end": (_i32, [_vp, _vp, C.POINTER(_i32), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]),
    "crender_raster": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _u32, _vp]),
    "crender_render_model": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32p, _vp, _vp, _vp, _vp, _u32, _vp]),
    "crender_prepare": (_i32, [_vp, _vp, _vp, _i64, _f32p, _u32, _vp]),
    "crender_draw": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _u32, _vp]),
    "crender_pipeline_create": (_i32, [C.POINTER(_vp), C.POINTER(_vp), _i32]),
    "crender_pipeline_destroy": (None, [_vp]),
    "crender_pipeline_frame": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32p, _vp, _vp, _vp, _vp, _u32, _vp]),
    "crender_pipeline_join": (_i32, [_vp, _vp]),
    "crender_pipeline_bind": (_i32, [_vp, _i32, _vp, _vp, _vp, _i64, _f32p, _vp, _vp, _vp, _vp, _u32]),
    "crender_pipeline_submit": (_i32, [_vp, _vp]),
    "crender_pipeline_timing_begin": (_i32, [_vp, _i32]),
    "crender_pipeline_timing_end": (_i32, [_vp, C.POINTER(_i32), C.POINTER(C.c_double)]),
    "crender_atomic_scratch_bytes": (_sz, [_i32, _i32]),
    "crender_raster_atomic": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                     _u32, _vp, _vp]),
    "crender_selfcheck_division": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "crender_present_u8": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "crender_guro_illumination": (_i32, [_vp, _vp, _f32p, _i32, _i32, _i32, _i32, _vp]),
    "crender_model_shift": (_i32, [_vp, _i64, C.POINTER(C.c_double), _i32, _vp]),
    "crender_model_scale": (_i32, [_vp, _i64, _vp, C.c_float, _i32, _vp]),
    "crender_model_stats": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "crender_pipeline_set_lookahead": (_i32, [_vp, _vp, _i32]),
    "crender_model_gather": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "crender_model_rotate": (_i32, [_vp, _i64, C.POINTER(C.c_double), _vp]),
    "crender_model_vertex_normals": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "crender_model_texture_colors": (_i32, [_vp, _i32, _i64, _vp, _i32, _i32, _vp, _vp]),
}

_lib = None


class CrenderError(RuntimeError):
    pass


def lib_path() -> str:
    # CRENDER_LIB: load another build of the same ABI (diagnostic builds, A/B variants)
    return os.environ.get("CRENDER_LIB") or _build.LIB_PATH


def load():
    """Load the shared library (never builds it: see __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise CrenderError(
            f"{path} is missing: build it with `python -m cython3dmodelrenderer_amd._build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the rasterizer.")
    L = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = L.crender_abi_version()
    if got != ABI_VERSION:
        raise CrenderError(f"libcrender_hip.so ABI {got}, binding expects {ABI_VERSION}")
    # CRENDER_RASTER_PATH=0|1: every plan that has not been told otherwise renders with that kernel
    # (crender_set_default_raster_path) — how the parity suite is run once per kernel
    forced = os.environ.get("CRENDER_RASTER_PATH")
    if forced not in (None, ""):
        if L.crender_set_default_raster_path(int(forced)) != OK:
            raise CrenderError(f"CRENDER_RASTER_PATH={forced}: expected -1, 0 or 1")
    _lib = L
    return L


def check(status: int, what: str):
    if status != OK:
        msg = load().crender_last_error()
        raise CrenderError(f"{what} failed (code {status}): {msg.decode() if msg else ''}")


def plan_debug_check(plan_handle, stream):
    """crender_plan_debug_check: raises CrenderError with the findings if the plan's cross-frame state is
    inconsistent (synchronises `stream`)."""
    buf = C.create_string_buffer(2048)
    rc = load().crender_plan_debug_check(plan_handle, stream, buf, len(buf))
    if rc == ESTATE:
        raise CrenderError("plan state: " + buf.value.decode(errors="replace"))
    check(rc, "crender_plan_debug_check")


def f32_16(mat):
    """Host float32[16] ctypes array from a 4x4 numpy matrix."""
    import numpy as np
    a = np.ascontiguousarray(mat, dtype=np.float32).reshape(16)
    return (C.c_float * 16)(*a.tolist())
