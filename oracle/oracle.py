"""ctypes front end of the CPU oracle (oracle/crender_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  The product package never imports this module.

The class mirrors the reference's ``AdvancedPixelBufferFiller`` (reference:
crender/cy/pixel_buffer_filler/advanced_pixel_buffer_filler.pyx:20-253) so the same
test body can drive the oracle and the HIP filler.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("culled", "empty", "drawn", "bbox_samples", "inside", "writes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "crender_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.oracle_projection_matrix.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _f32p]
    L.oracle_projection_matrix.restype = None
    L.oracle_project.argtypes = [_f32p, _f32p, C.c_int64, _f32p, C.c_int, C.c_int, C.c_int]
    L.oracle_project.restype = None
    L.oracle_bbox.argtypes = [_f32p, C.c_int, C.c_int, _i32p]
    L.oracle_bbox.restype = None
    L.oracle_bar.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
    L.oracle_bar.restype = None
    L.oracle_raster_serial.argtypes = [_f32p, _f32p, _f32p, C.c_int64, _f32p, _f32p, _f32p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.oracle_raster_serial.restype = None
    L.oracle_locks_create.argtypes = [C.c_int, C.c_int]
    L.oracle_locks_create.restype = C.c_void_p
    L.oracle_locks_destroy.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.oracle_locks_destroy.restype = None
    L.oracle_raster_omp.argtypes = [_f32p, _f32p, _f32p, C.c_int64, _f32p, _f32p, _f32p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.oracle_raster_omp.restype = None
    L.oracle_render_model.argtypes = [_f32p, _f32p, _f32p, C.c_int64, _f32p, _f32p, _f32p, _f32p,
                                      C.c_int, C.c_int, _f32p, C.c_void_p, C.c_int]
    L.oracle_render_model.restype = None
    L.oracle_clear.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int]
    L.oracle_clear.restype = None
    L.oracle_guro.argtypes = [_f32p, _f32p, _f32p, C.c_int64]
    L.oracle_guro.restype = None
    _lib = L
    return L


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def projection_matrix(fov, z_near, z_far, h, w):
    P = np.zeros((4, 4), np.float32)
    lib().oracle_projection_matrix(float(fov), float(z_near), float(z_far), int(h), int(w), P)
    return P


def project(tri, P, w, h, n_threads=1):
    tri = _c(tri)
    out = np.empty_like(tri)
    lib().oracle_project(tri, out, tri.shape[0], _c(P), int(w), int(h), int(n_threads))
    return out


def bbox(tri9, w, h):
    out = np.zeros(4, np.int32)
    lib().oracle_bbox(_c(tri9).reshape(9), int(w), int(h), out)
    return out


def bar(tri9, x, y):
    out = np.zeros(3, np.float32)
    lib().oracle_bar(_c(tri9).reshape(9), int(x), int(y), out)
    return out


def guro_light(light_direction):
    """GuroIllumination.__init__ (guro_illumination.py:15-18): the light vector flipped and
    normalised, float32."""
    flipped = -np.asarray(light_direction, dtype="float32")
    return (flipped / np.linalg.norm(flipped)).astype(np.float32)


def guro(color_buffer, normals_buffer, light_direction):
    """GuroIllumination(light_direction).draw_illumination(color_buffer, normals_buffer)
    (guro_illumination.py:20-27), in place, by the C restatement oracle_guro."""
    assert color_buffer.dtype == np.float32 and color_buffer.flags.c_contiguous
    lib().oracle_guro(color_buffer.reshape(-1), _c(normals_buffer).reshape(-1), guro_light(light_direction),
                      color_buffer.size // 3)
    return color_buffer


class OracleFiller:
    """CPU twin of the reference's AdvancedPixelBufferFiller (.pyx:20-253).

    ``mode='serial'`` is the deterministic 1-thread order (the parity reference);
    ``mode='omp'`` is the Version-C shape (dynamic OpenMP loop + per-pixel locks) used
    as the timed CPU baseline.  Buffers persist across calls like the reference's.
    """

    def __init__(self, h, w, fov=90.0, z_near=0.1, z_far=1000.0, n_threads=1, mode="serial"):
        self.h, self.w = int(h), int(w)
        self.n_threads = int(n_threads)
        self.mode = mode
        self.proj_mat = projection_matrix(fov, z_near, z_far, h, w)
        self.z_buffer = np.empty((self.h, self.w), np.float32)
        self.color_buffer = np.empty((self.h, self.w, 3), np.float32)
        self.normals_buffer = np.empty((self.h, self.w, 3), np.float32)
        self.winner = np.full((self.h, self.w), -1, np.int32)
        self.stats = Stats()
        self._locks = None
        if mode == "omp":
            self._locks = lib().oracle_locks_create(self.h, self.w)
        self.clear()

    def __del__(self):
        if getattr(self, "_locks", None):
            lib().oracle_locks_destroy(self._locks, self.h, self.w)
            self._locks = None

    def get_size(self):
        return self.h, self.w

    def clear(self):
        lib().oracle_clear(self.z_buffer, self.color_buffer, self.normals_buffer,
                           self.h, self.w, self.n_threads)
        self.winner.fill(-1)

    def render_arrays(self, tri, col, nrm, y0=0, y1=None):
        tri, col, nrm = _c(tri), _c(col), _c(nrm)
        T = tri.shape[0]
        y1 = self.h if y1 is None else y1
        proj = project(tri, self.proj_mat, self.w, self.h, self.n_threads)
        self.projected = proj
        if self.mode == "omp":
            lib().oracle_raster_omp(proj, col, nrm, T, self.z_buffer, self.color_buffer,
                                    self.normals_buffer, self.h, self.w, y0, y1,
                                    self._locks, self.n_threads)
        else:
            lib().oracle_raster_serial(proj, col, nrm, T, self.z_buffer, self.color_buffer,
                                       self.normals_buffer, self.h, self.w, y0, y1,
                                       self.winner.ctypes.data_as(C.c_void_p),
                                       C.cast(C.pointer(self.stats), C.c_void_p))

    def render_model(self, model):
        self.render_arrays(model._vertices_by_triangles, model._colors_by_triangles,
                           model._normals_by_triangles)

    def get_normals_buffer(self):
        return self.normals_buffer

    def get_color_buffer(self):
        return self.color_buffer

    def get_z_buffer(self):
        return self.z_buffer
