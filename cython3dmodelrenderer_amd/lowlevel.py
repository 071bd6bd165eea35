"""Thin tensor-level wrappers over the C ABI (include/crender_hip.h): one Python function
per entry point, torch-ROCm tensors in, work enqueued on torch's current stream.
The filler class and the tests are built on these."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _chk_f32(t, name, shape_tail=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float32 device tensor")
    if shape_tail is not None and tuple(t.shape[1:]) != shape_tail:
        raise ValueError(f"{name}: expected shape [*, {shape_tail}], got {tuple(t.shape)}")


def projection_matrix(fov, z_near, z_far, h, w):
    """float32 [4,4] numpy matrix of the reference's _init_projection_matrix (.pyx:83-90)."""
    P = (C.c_float * 16)()
    _capi.check(_capi.load().crender_projection_matrix(float(fov), float(z_near), float(z_far),
                                                       int(h), int(w), P), "crender_projection_matrix")
    return np.array(P[:], dtype=np.float32).reshape(4, 4)


def project(tri, P, w, h, out=None):
    """K1 (.pyx:106-130) on a [T,3,3] device tensor; ``out`` may be ``tri`` (in place)."""
    _chk_f32(tri, "tri", (3, 3))
    out = torch.empty_like(tri) if out is None else out
    _chk_f32(out, "out", (3, 3))
    with torch.cuda.device(tri.device):
        _capi.check(_capi.load().crender_project(tri.data_ptr(), out.data_ptr(), tri.shape[0],
                                                 _capi.f32_16(P), int(w), int(h), _stream(tri.device)),
                    "crender_project")
    return out


class FrameBuffers:
    """z [H,W], colour [H,W,3], normal [H,W,3] (+ optional winner [H,W] int32) on a device,
    initialised like __cinit__ (.pyx:65-67)."""

    def __init__(self, h, w, device="cuda:0", winner=True):
        self.h, self.w = int(h), int(w)
        self.device = torch.device(device)
        self.z = torch.full((h, w), 1e6, dtype=torch.float32, device=self.device)
        self.color = torch.zeros((h, w, 3), dtype=torch.float32, device=self.device)
        self.normals = torch.zeros((h, w, 3), dtype=torch.float32, device=self.device)
        self.winner = torch.full((h, w), -1, dtype=torch.int32, device=self.device) if winner else None

    def load(self, z, color, normals):
        self.z.copy_(torch.as_tensor(z))
        self.color.copy_(torch.as_tensor(color))
        self.normals.copy_(torch.as_tensor(normals))

    def numpy(self):
        torch.cuda.synchronize(self.device)
        return (self.z.cpu().numpy(), self.color.cpu().numpy(), self.normals.cpu().numpy(),
                self.winner.cpu().numpy() if self.winner is not None else None)

    def clear(self, y0=0, y1=None):
        y1 = self.h if y1 is None else y1
        with torch.cuda.device(self.device):
            _capi.check(_capi.load().crender_clear(
                self.z.data_ptr(), self.color.data_ptr(), self.normals.data_ptr(),
                self.winner.data_ptr() if self.winner is not None else None,
                self.h, self.w, y0, y1, _stream(self.device)), "crender_clear")


class Plan:
    """Owner of a crender_plan and its device workspace."""

    def __init__(self, h, w, max_T, y0=0, y1=None, bin_capacity=0, tile=0, device="cuda:0"):
        self._lib = _capi.load()
        self.h, self.w = int(h), int(w)
        self.y0, self.y1 = int(y0), int(h if y1 is None else y1)
        self.device = torch.device(device)
        nbytes = self._lib.crender_plan_workspace_bytes(self.h, self.w, self.y0, self.y1, int(max_T),
                                                        int(bin_capacity), int(tile))
        if nbytes == 0:
            raise _capi.CrenderError("crender_plan_workspace_bytes: bad geometry")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_plan_create(C.byref(self.handle), self.h, self.w, self.y0,
                                                      self.y1, int(max_T), int(bin_capacity), int(tile),
                                                      self.workspace.data_ptr(), nbytes,
                                                      _stream(self.device)), "crender_plan_create")

    def __del__(self):
        handle, self.handle = getattr(self, "handle", None), None     # (at interpreter exit `C` may be gone already)
        if handle:
            self._lib.crender_plan_destroy(handle)

    def last_frame_direct(self):
        return bool(self._lib.crender_plan_last_frame_direct(self.handle))

    def last_frame_binning(self):
        """0 = count / scan / fill, 1 = direct bins, 2 = pair bins."""
        return int(self._lib.crender_plan_last_frame_binning(self.handle))

    def debug_check(self):
        """crender_plan_debug_check on the current stream: raises if the plan's cross-frame state is inconsistent."""
        with torch.cuda.device(self.device):
            _capi.plan_debug_check(self.handle, _stream(self.device))

    def set_raster_path(self, path):
        """-1 = the plan chooses (default), 0 = general kernel, 1 = pixel owners only; every choice renders
        every tile exactly (crender_plan_set_raster_path)."""
        _capi.check(self._lib.crender_plan_set_raster_path(self.handle, int(path)), "crender_plan_set_raster_path")

    def last_raster_path(self):
        return int(self._lib.crender_plan_last_raster_path(self.handle))

    def bin_usage(self):
        need, cap = C.c_int64(), C.c_int64()
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_plan_last_bin_usage(self.handle, _stream(self.device),
                                                              C.byref(need), C.byref(cap)),
                        "crender_plan_last_bin_usage")
        return need.value, cap.value


def _tri_ptrs(tri, col, nrm):
    _chk_f32(tri, "tri", (3, 3))
    _chk_f32(col, "col", (3, 3))
    _chk_f32(nrm, "nrm", (3, 3))
    if not (tri.shape == col.shape == nrm.shape):
        raise ValueError("tri, col, nrm must have the same shape")
    return tri.data_ptr(), col.data_ptr(), nrm.data_ptr(), tri.shape[0]


def _flags(clear, direct_bins, extra=0):
    return (_capi.FUSED_CLEAR if clear else 0) | (0 if direct_bins else _capi.NO_DIRECT_BINS) | int(extra)


def raster(plan, proj, col, nrm, fb, clear=False, direct_bins=True, flags=0):
    """K2 (.pyx:177-244), tile path, on already projected triangles."""
    p, c, n, T = _tri_ptrs(proj, col, nrm)
    with torch.cuda.device(fb.device):
        _capi.check(plan._lib.crender_raster(
            plan.handle, p, c, n, T, fb.z.data_ptr(), fb.color.data_ptr(), fb.normals.data_ptr(),
            fb.winner.data_ptr() if fb.winner is not None else None,
            _flags(clear, direct_bins, flags), _stream(fb.device)), "crender_raster")


def render_model(plan, tri, col, nrm, P, fb, clear=False, direct_bins=True, flags=0):
    """render_model (.pyx:92-104): K1 fused into the binning pass + K2."""
    p, c, n, T = _tri_ptrs(tri, col, nrm)
    with torch.cuda.device(fb.device):
        _capi.check(plan._lib.crender_render_model(
            plan.handle, p, c, n, T, _capi.f32_16(P), fb.z.data_ptr(), fb.color.data_ptr(),
            fb.normals.data_ptr(), fb.winner.data_ptr() if fb.winner is not None else None,
            _flags(clear, direct_bins, flags), _stream(fb.device)), "crender_render_model")


def prepare(plan, tri, nrm, P, stream=None, direct_bins=True):
    """First half of render_model: K1 + binning into the plan (``P=None``: ``tri`` is already
    projected).  ``stream``: a torch.cuda.Stream, default the current one."""
    _chk_f32(tri, "tri", (3, 3))
    _chk_f32(nrm, "nrm", (3, 3))
    s = C.c_void_p((stream or torch.cuda.current_stream(plan.device)).cuda_stream)
    with torch.cuda.device(plan.device):
        _capi.check(plan._lib.crender_prepare(plan.handle, tri.data_ptr(), nrm.data_ptr(), tri.shape[0],
                                              None if P is None else _capi.f32_16(P),
                                              _flags(False, direct_bins), s), "crender_prepare")


def draw(plan, col, nrm, T, fb, proj=None, clear=False, stream=None, direct_bins=True):
    """Second half of render_model: K2 from the plan's bins (``proj=None``: the vertices that
    ``prepare`` projected into the plan)."""
    s = C.c_void_p((stream or torch.cuda.current_stream(fb.device)).cuda_stream)
    with torch.cuda.device(fb.device):
        _capi.check(plan._lib.crender_draw(
            plan.handle, None if proj is None else proj.data_ptr(), col.data_ptr(), nrm.data_ptr(), int(T),
            fb.z.data_ptr(), fb.color.data_ptr(), fb.normals.data_ptr(),
            fb.winner.data_ptr() if fb.winner is not None else None,
            _flags(clear, direct_bins), s), "crender_draw")


def raster_atomic(proj, col, nrm, fb, y0=0, y1=None, clear=False, keys=None):
    """K2 by the independent global-atomic path (cross-check implementation)."""
    lib = _capi.load()
    p, c, n, T = _tri_ptrs(proj, col, nrm)
    y1 = fb.h if y1 is None else y1
    if keys is None:
        keys = torch.empty(lib.crender_atomic_scratch_bytes(fb.h, fb.w), dtype=torch.uint8,
                           device=fb.device)
    with torch.cuda.device(fb.device):
        _capi.check(lib.crender_raster_atomic(
            p, c, n, T, fb.z.data_ptr(), fb.color.data_ptr(), fb.normals.data_ptr(),
            fb.winner.data_ptr() if fb.winner is not None else None, fb.h, fb.w, int(y0), int(y1),
            _capi.FUSED_CLEAR if clear else 0, keys.data_ptr(), _stream(fb.device)),
            "crender_raster_atomic")
    return keys


def guro_illumination(fb, light3, y0=0, y1=None):
    y1 = fb.h if y1 is None else y1
    light = (C.c_float * 3)(*[float(v) for v in light3])
    with torch.cuda.device(fb.device):
        _capi.check(_capi.load().crender_guro_illumination(
            fb.color.data_ptr(), fb.normals.data_ptr(), light, fb.h, fb.w, int(y0), int(y1),
            _stream(fb.device)), "crender_guro_illumination")
