"""Device-resident ``Model``: the reference's mesh container
(crender/cy/data_structures/model.py:118-256) with its vertex, index and normal arrays kept in HBM
and the transforms that are exactly reproducible there running as HIP kernels:

    shift, scale, the mean vertex, the max span, the three ``*_by_triangles`` gathers, and (row f4)
    the per-texture-coordinate colour lookup of a textured model (model.py:143-151)

each in numpy's own operation order (float32 elementwise; float64 where numpy promotes; the mean
as numpy's row-after-row float32 sum), so the arrays handed to the filler are, bit for bit, what
the host ``Model`` produces.

``rotate`` — the matrix product and the vertex-normal computation it triggers (model.py:175-208,
238-256: 1.2 s of Python loops for T-Rex on the host) — runs on the device too
(``crender_model_rotate``, ``crender_model_vertex_normals``), to 1e-5 of the host ``Model`` rather
than bit for bit: numpy takes ``matmul``, ``np.linalg.norm`` and ``np.dot`` from its BLAS build,
whose summation order is not a property of the reference, and the de-duplication test
``dot >= 1`` is discontinuous, so that a 1-ulp difference can change which face normals a vertex
averages (a handful of T-Rex's 6 909 vertices; counted in the tests).  ``rotate(angles,
on_host=True)`` takes the host ``Model``'s path (download, numpy, upload) for callers who need the
host's bits.

The filler takes torch tensors as they are: ``render_model(device_model)`` uploads nothing.
The transforms rewrite the resident arrays IN PLACE on the current stream: with frames of a swap
chain still in flight (``render_frame`` on a pipelined filler) call ``filler.join()`` first — the
chain's streams are not ordered against the caller's between joins — and hand the model to
``render_model`` / ``render_arrays`` again afterwards, which re-binds the chain's slots (what was
binned ahead from the old vertices is dropped).

The mean vertex and the max span are worked out when asked for (``get_mean_vertex``,
``get_max_span``, ``scale(keep_position=True)``), not after every transform: the mean is numpy's
sequential float32 sum, one lane per component, which a multi-million-vertex model should not pay
on every ``shift``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _capi
from .model import Model


class DeviceModel:
    def __init__(self, model: Model, device="cuda:0"):
        self._lib = _capi.load()
        if not torch.cuda.is_available():
            raise _capi.CrenderError("DeviceModel needs a ROCm GPU")
        self.device = torch.device(device)
        self._host = model                      # parser / rotate / normals live there
        # Counts every rewrite of the *_by_triangles arrays.  The transforms write them in place
        # through raw pointers, which torch's version counter never sees: a filler that keeps a
        # sorted snapshot of the arrays (tile-coherent order) compares this number instead.
        self.generation = 0
        self._upload()

    # ------------------------------------------------------------------ plumbing --
    def _dev(self, a, dtype):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.device)

    def _dev_index(self, idx, n):
        """Face indices as the gather kernel takes them: numpy's fancy indexing counts a negative
        index (an .obj file's relative reference, model.py:276-279) from the end."""
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size and (idx.min() < -n or idx.max() >= n):
            raise IndexError(f"index out of bounds for axis 0 with size {n}")
        return self._dev(np.where(idx < 0, idx + n, idx), np.int32)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _upload(self):
        m = self._host
        self._vertices = self._dev(m._vertices, np.float32)
        self._triangles_vertices = self._dev_index(m._triangles_vertices, len(m._vertices))
        self._normals = self._dev(m._normals, np.float32)
        self._triangles_normals = self._dev_index(m._triangles_normals, len(m._normals))
        if m._texture is not None:
            self._sample_texture(m)
        elif m._colors_by_triangles is not None:
            self._colors_by_triangles = self._dev(m._colors_by_triangles, np.float32)
        elif not hasattr(self, "_colors_by_triangles"):
            self._colors_by_triangles = None          # (colours set on the device model survive a rotate)
        T = self._triangles_vertices.shape[0]
        self._vertices_by_triangles = torch.empty((T, 3, 3), dtype=torch.float32, device=self.device)
        self._normals_by_triangles = torch.empty((T, 3, 3), dtype=torch.float32, device=self.device)
        self._stats = torch.zeros(4, dtype=torch.float32, device=self.device)   # mean[3], max span
        self._csr = None                 # (offsets, occurrences, face normals, taken): made at the first rotate
        self._gather(self._normals, self._triangles_normals, self._normals_by_triangles)
        self._update()

    def _sample_texture(self, m):
        """model.py:143-151 on the device: texture (uint8 BGR), texture coordinates and the faces'
        texture indices go up as they were parsed; colours and colours-by-triangles are made there."""
        uv = self._dev(m._texture_coords, np.float32)
        tex = torch.from_numpy(np.ascontiguousarray(m._texture[:, :, :3], dtype=np.uint8)).to(self.device)
        th, tw = int(tex.shape[0]), int(tex.shape[1])
        self._colors = torch.empty((uv.shape[0], 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_model_texture_colors(uv.data_ptr(), int(uv.shape[1]), uv.shape[0],
                                                               tex.data_ptr(), th, tw, self._colors.data_ptr(),
                                                               self._stream()), "crender_model_texture_colors")
        idx = self._dev_index(m._triangles_texture_coords, uv.shape[0])
        self._colors_by_triangles = torch.empty((idx.shape[0], 3, 3), dtype=torch.float32, device=self.device)
        self._gather(self._colors, idx, self._colors_by_triangles)

    def _gather(self, attr, index, out):
        self.generation += 1
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_model_gather(attr.data_ptr(), index.data_ptr(), out.data_ptr(),
                                                       index.shape[0], self._stream()), "crender_model_gather")

    def _update(self):
        """model.py:153-160 after a change of the vertices: the gather now; mean vertex and max span
        when somebody asks (``_ensure_stats``)."""
        self._gather(self._vertices, self._triangles_vertices, self._vertices_by_triangles)
        self._stats_valid = False
        self._stats_host = None

    def _ensure_stats(self):
        if not self._stats_valid:
            with torch.cuda.device(self.device):
                _capi.check(self._lib.crender_model_stats(self._vertices.data_ptr(), self._vertices.shape[0],
                                                          self._stats.data_ptr(), self._stats.data_ptr() + 12,
                                                          self._stream()), "crender_model_stats")
            self._stats_valid = True

    def _fetch_stats(self):
        if self._stats_host is None:
            self._ensure_stats()
            self._stats_host = self._stats.cpu().numpy()
        return self._stats_host

    def _vertex_face_csr(self):
        """Per vertex, the faces of its (face, corner) occurrences in ascending order — the order
        in which the reference's loop meets them (model.py:178-186); made once per upload."""
        if self._csr is None:
            faces = self._triangles_vertices.cpu().numpy().astype(np.int64)          # (non-negative already)
            flat = faces.ravel()
            order = np.argsort(flat, kind="stable")           # by vertex; (face, corner) order kept
            occ = (order // 3).astype(np.int32)
            V = self._vertices.shape[0]
            offs = np.zeros(V + 1, np.int64)
            np.cumsum(np.bincount(flat, minlength=V), out=offs[1:])
            T = faces.shape[0]
            self._csr = (self._dev(offs, np.int32), self._dev(occ, np.int32),
                         torch.empty((T, 3), dtype=torch.float32, device=self.device),
                         torch.empty(max(3 * T, 1), dtype=torch.uint8, device=self.device))
        return self._csr

    # ----------------------------------------------------------- reference API --
    def shift(self, shift):
        arr = np.asarray(shift)
        if arr.shape != (3,):
            arr = np.broadcast_to(arr, (3,))
        is_f32 = arr.dtype == np.float32 or arr.dtype == np.float16
        s3 = (C.c_double * 3)(*[float(v) for v in arr])
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_model_shift(self._vertices.data_ptr(), self._vertices.shape[0], s3,
                                                      1 if is_f32 else 0, self._stream()), "crender_model_shift")
        self._update()

    def scale(self, scale_coef, keep_position=True):
        coef = np.float32(scale_coef)            # vtx *= coef is a float32 in-place multiply
        if keep_position:
            self._ensure_stats()                 # (the mean the reference subtracts: of the vertices as they are)
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_model_scale(self._vertices.data_ptr(), self._vertices.shape[0],
                                                      self._stats.data_ptr(), C.c_float(float(coef)),
                                                      1 if keep_position else 0, self._stream()),
                        "crender_model_scale")
        self._update()

    def rotate(self, angles, on_host=False):
        """Model.rotate (model.py:238-256): rotate about x, then y, then z (degrees) and recompute
        the vertex normals — on the device (see the module docstring), or with ``on_host=True``
        through the host ``Model`` (download, numpy, upload) for the host's bits."""
        assert len(angles) == 3
        if on_host:
            m = self._host
            m._set_geometry(self._vertices.cpu().numpy(), m._triangles_vertices, m._normals,
                            m._triangles_normals, recalc=False)
            m.rotate(angles)
            self._upload()
            return
        from .model import _rot2
        ax, ay, az = angles
        rx, ry, rz = np.eye(3), np.eye(3), np.eye(3)
        rx[1:, 1:] = _rot2(ax)
        ry[::2, ::2] = _rot2(ay)
        rz[:2, :2] = _rot2(az)
        rot = np.matmul(np.matmul(rx, ry), rz)             # float64, composed as the reference composes it
        R9 = (C.c_double * 9)(*[float(v) for v in rot.reshape(9)])     # new = v @ rot.T: row j of rot per output j
        offs, occ, fn, taken = self._vertex_face_csr()
        V, T = self._vertices.shape[0], self._triangles_vertices.shape[0]
        if self._normals.shape[0] != V:
            self._normals = torch.empty((V, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_model_rotate(self._vertices.data_ptr(), V, R9, self._stream()),
                        "crender_model_rotate")
            _capi.check(self._lib.crender_model_vertex_normals(
                self._vertices.data_ptr(), V, self._triangles_vertices.data_ptr(), T, offs.data_ptr(),
                occ.data_ptr(), fn.data_ptr(), taken.data_ptr(), self._normals.data_ptr(), self._stream()),
                "crender_model_vertex_normals")
        self._triangles_normals = self._triangles_vertices      # (model.py:168: recalculated normals are per vertex)
        self._gather(self._normals, self._triangles_normals, self._normals_by_triangles)
        self._update()

    def touch(self):
        """Tell whoever keeps a snapshot of the by-triangle arrays (a filler's tile-coherent copy)
        that they were rewritten behind this class's back — a write through ``data_ptr()``, a kernel
        of the caller's.  Every method of this class that rewrites them does the same: ``generation``
        is part of the contract between a model and ``AdvancedPixelBufferFiller``."""
        self.generation += 1

    def get_mean_vertex(self):
        return self._fetch_stats()[:3].copy()

    def get_max_span(self):
        return np.float32(self._fetch_stats()[3])

    def n_triangles(self):
        return int(self._triangles_vertices.shape[0])

    def n_vertices(self):
        return int(self._vertices.shape[0])

    def set_uniform_color(self, bgr=(255.0, 255.0, 255.0)):
        T = self.n_triangles()
        self._colors_by_triangles = torch.tensor(bgr, dtype=torch.float32, device=self.device) \
            .expand(T, 3, 3).contiguous()
        self.generation += 1
