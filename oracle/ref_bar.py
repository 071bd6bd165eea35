"""Loader for oracle/_ref/math_utils.so — the reference's own barycentric function
(reference: crender/cy/pixel_buffer_filler/math_utils.pyx:8-34) compiled by
oracle/build_ref.sh from the source where it lies.

TEST INFRASTRUCTURE ONLY (used by tests/ to pin oracle_bar bit for bit).  The cdef
function is not Python-callable; it is reached through the C-API capsule Cython
exports for cimport-ing modules (``__pyx_capi__``).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "math_utils.so")


class Vec3(C.Structure):
    # reference: math_utils.pxd:1-4
    _fields_ = [("x1", C.c_float), ("x2", C.c_float), ("x3", C.c_float)]


def available() -> bool:
    return os.path.exists(_SO)


_fn = None


def _load():
    global _fn
    if _fn is not None:
        return _fn
    spec = importlib.util.spec_from_file_location("math_utils", _SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cap = mod.__pyx_capi__["compute_bar_coords_single_pixel"]
    api = C.pythonapi
    api.PyCapsule_GetName.restype = C.c_char_p
    api.PyCapsule_GetName.argtypes = [C.py_object]
    api.PyCapsule_GetPointer.restype = C.c_void_p
    api.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    name = api.PyCapsule_GetName(cap)
    assert b"(float *, int, int)" in name, name
    ptr = api.PyCapsule_GetPointer(cap, name)
    _fn = C.CFUNCTYPE(Vec3, C.POINTER(C.c_float), C.c_int, C.c_int)(ptr)
    _fn._keepalive = mod
    return _fn


def bar(tri9, x, y):
    """Barycentrics of integer pixel (x, y) as the reference computes them."""
    t = np.ascontiguousarray(tri9, dtype=np.float32).reshape(9)
    v = _load()(t.ctypes.data_as(C.POINTER(C.c_float)), int(x), int(y))
    return np.array([v.x1, v.x2, v.x3], np.float32)
