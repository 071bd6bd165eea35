"""MI355X drop-in for the reference's Version-C ``AdvancedPixelBufferFiller``.

Mirrors, name for name, the class at
crender/cy/pixel_buffer_filler/advanced_pixel_buffer_filler.pyx:20-253 of the reference:

    AdvancedPixelBufferFiller(h, w, fov=90.0, z_near=0.1, z_far=1000.0, n_threads=1)
    .get_size()  .render_model(model)
    .get_normals_buffer()  .get_color_buffer()  .get_z_buffer()

The framebuffers and the model arrays live in HBM as torch-ROCm tensors; every call
goes through the C ABI of libcrender_hip.so (include/crender_hip.h) — the per-frame calls
(render_model, the swap chain's submit) through the torch C++ extension ``crender_torch``
(csrc/crender_torch.cpp), which unwraps the tensors and torch's current stream and calls the
same C entry points; plan management and the rarely used calls through ctypes.  torch supplies
device memory and the HIP stream only.  There is no CPU path: without the HIP library, the
extension or a GPU the constructor raises.

Behaviour kept from the reference
  * buffers persist across ``render_model`` calls and are never cleared, so successive
    renders composite (``Renderer.reset_buffers`` is a no-op there, renderer.py:51-52);
  * getters hand out writable numpy arrays that stay valid, that callers may change in place
    (GuroIllumination does, guro_illumination.py:27) and that show every later render, like the
    reference's views of its own buffers (.pyx:246-253): in-place changes are carried to the
    device before the next render, and every array handed out so far is refreshed at the end of
    each ``render_model``.  The arrays are views of PINNED host buffers allocated once per plane;
    only planes that have actually been handed out cross PCIe (one asynchronous copy each way
    per render), a filler nobody asked a buffer of copies nothing;
  * the three model arrays are read afresh on every ``render_model`` call, as the reference's
    per-call ``.copy()`` does (.pyx:94-96): an in-place edit of ``model._vertices_by_triangles``
    is honoured (``cache_inputs=True`` restores the upload cache keyed by array identity).
    numpy arrays go up through ONE pinned staging buffer and one asynchronous copy;
  * ``model._colors_by_triangles is None`` raises AttributeError, float64 arrays raise
    ValueError (.pyx:94-96 binds ``float[:, :, :]`` after ``.copy()``);
  * ``n_threads`` is accepted and ignored.
Not kept: the debug printf lines (.pyx:112,197) and the swallowed ZeroDivisionError for
a vertex with z == 0 (out of contract; IEEE inf/NaN semantics apply instead).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np
import torch

from .. import _capi, _torch_ext


def _check_host_f32(a, name):
    """numpy view of a [T, 3, 3] float32 host array (any strides), with the reference's errors."""
    if a is None:
        raise AttributeError("'NoneType' object has no attribute 'copy'")     # .pyx:94-96
    arr = np.asarray(a)
    if arr.dtype != np.float32:
        kind = "double" if arr.dtype == np.float64 else str(arr.dtype)
        raise ValueError(f"Buffer dtype mismatch, expected 'float' but got '{kind}'")
    if arr.ndim != 3 or arr.shape[1] != 3 or arr.shape[2] != 3:
        raise ValueError(f"{name} must have shape [T, 3, 3], got {tuple(arr.shape)}")
    return arr


def _as_device_f32(a, name, device):
    """[T, 3, 3] float32 contiguous tensor on `device` from numpy / torch input."""
    if a is None:
        # what `None.copy()` raises in the reference (.pyx:94-96)
        raise AttributeError("'NoneType' object has no attribute 'copy'")
    if isinstance(a, torch.Tensor):
        if a.dtype != torch.float32:
            raise ValueError(f"Buffer dtype mismatch, expected 'float' but got '{a.dtype}' ({name})")
        t = a.to(device=device).contiguous()
    else:
        arr = np.asarray(a)
        if arr.dtype != np.float32:
            kind = "double" if arr.dtype == np.float64 else str(arr.dtype)
            raise ValueError(f"Buffer dtype mismatch, expected 'float' but got '{kind}'")
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    if t.dim() != 3 or t.shape[1] != 3 or t.shape[2] != 3:
        raise ValueError(f"{name} must have shape [T, 3, 3], got {tuple(t.shape)}")
    return t


# Handle of torch's current stream on a device, and the current device index: the private fast
# paths when this torch has them (the public ones build a Stream object / go through Python
# wrappers per call, ~1 us of a 7 us frame on the host).
_current_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream",
                              lambda index: torch.cuda.current_stream(index).cuda_stream)
_current_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)


_DIRECT_MAX_TRIANGLES = 1 << 16      # kDirectMaxTriangles of csrc/plan.h: beyond, count / scan / fill


class _Stage:
    """One staging slot of numpy inputs: pinned host [3, T, 3, 3] -> device [3, T, 3, 3], ONE copy.
    A slot made for `cap` triangles serves any T <= cap (the arrays are the first 27 T floats of it)."""
    __slots__ = ("cap", "T", "_pin", "_dev", "pin", "host", "dev", "done", "busy")

    def __init__(self, cap, device):
        self.cap = max(int(cap), 1)
        with torch.cuda.device(device):
            self._pin = torch.empty(27 * self.cap, dtype=torch.float32, pin_memory=True)
            self._dev = torch.empty(27 * self.cap, dtype=torch.float32, device=device)
            self.done = torch.cuda.Event()       # the last copy out of the pinned side has been read
        self.busy = False
        self.T = -1

    def shape(self, T):
        if T != self.T:
            self.T = T
            self.pin = self._pin[:27 * T].view(3, T, 3, 3)
            self.dev = self._dev[:27 * T].view(3, T, 3, 3)
            self.host = self.pin.numpy()
        return self


class _Frame:
    """A launched frame whose bin lists have not been verified: its number on the plan, and what a
    redo needs — the flags and the inputs it was rendered from, kept alive."""
    __slots__ = ("ticket", "flags", "inputs", "order", "private", "light", "stage")

    def __init__(self, ticket, flags, inputs, order, private, light, stage):
        self.ticket, self.flags, self.inputs, self.order = ticket, flags, inputs, order
        self.private, self.light, self.stage = private, light, stage


class _FramePipeline:
    """Swap chain for ``render_frame`` (``crender_pipeline_*``): frame i renders on the library's
    stream i % depth with plan i % depth into framebuffer set i % depth, so up to `depth` frames
    overlap on the GPU with no event between them (T-Rex 1024^2 on MI355X: 30 / 16 / 12.2 us per
    frame at depth 1 / 2 / 3, 11.0 at depth 4 with GPU_MAX_HW_QUEUES=8).  Every frame still does all of its work
    (clear + project + bin + rasterize) into a complete framebuffer; the filler's
    ``z_buffer / color_buffer / normals_buffer`` always name the most recently submitted frame's
    set.  ``join`` orders the caller's stream after all submitted frames."""

    def __init__(self, filler, T, depth=2):
        self.lib = filler._lib
        self.device = filler.device
        self.depth = int(depth)
        self.plans, self.workspaces = [], []
        # look-ahead (crender_pipeline_set_lookahead): a second plan per slot, so that the launch that
        # rasterizes a frame also bins the slot's next one — scenes that fit the direct bins
        self.lookahead = 0 < int(T) <= _DIRECT_MAX_TRIANGLES and filler._extra_flags == 0 \
            if filler._lookahead is None else bool(filler._lookahead)
        for _ in range(self.depth * (2 if self.lookahead else 1)):
            cap = max(filler._bin_request, filler._bin_floor)
            tile = self.tile = filler._chain_tile()
            nbytes = self.lib.crender_plan_workspace_bytes(filler.h, filler.w, filler.y0, filler.y1,
                                                           max(int(T), 1), cap, tile)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            plan = C.c_void_p()
            _capi.check(self.lib.crender_plan_create(C.byref(plan), filler.h, filler.w, filler.y0,
                                                     filler.y1, max(int(T), 1), cap,
                                                     tile, ws.data_ptr(), nbytes, filler._stream()),
                        "crender_plan_create")
            if filler._raster_path is not None:
                _capi.check(self.lib.crender_plan_set_raster_path(plan, int(filler._raster_path)),
                            "crender_plan_set_raster_path")
            self.plans.append(plan)
            self.workspaces.append(ws)
        self.max_T = max(int(T), 1)
        self.handle = C.c_void_p()
        arr = (C.c_void_p * self.depth)(*[p.value for p in self.plans[:self.depth]])
        with torch.cuda.device(self.device):       # the pipeline's streams live on this device
            _capi.check(self.lib.crender_pipeline_create(C.byref(self.handle), arr, self.depth),
                        "crender_pipeline_create")
            if self.lookahead:
                more = (C.c_void_p * self.depth)(*[p.value for p in self.plans[self.depth:]])
                _capi.check(self.lib.crender_pipeline_set_lookahead(self.handle, more, self.depth),
                            "crender_pipeline_set_lookahead")
        # framebuffer sets of the swap chain: the filler's own buffers and copies of them
        front = (filler.z_buffer, filler.color_buffer, filler.normals_buffer, filler.winner_buffer)
        self.sets = [front] + [tuple(None if t is None else t.clone() for t in front)
                               for _ in range(self.depth - 1)]
        self.k = 0                 # set / stream / plan of the next frame (the library counts alike)
        self.n = 0
        self.pending = False
        self._args = None          # (inputs, flags) the slots are bound to
        self._submit = filler._ext.pipeline_submit       # (handle, device index): torch's current stream
        self._handle_int = self.handle.value
        self._index = self.device.index if self.device.index is not None else torch.cuda.current_device()

    def close(self):
        if self.handle:
            self.lib.crender_pipeline_destroy(self.handle)
            self.handle = C.c_void_p()
        for plan in self.plans:
            self.lib.crender_plan_destroy(plan)
        self.plans = []

    def frame(self, filler):
        want = (filler._inputs, filler._extra_flags | (_capi.STATIC_INPUTS if filler._inputs_private else 0),
                filler._fused_light, filler._order)
        if self._args is None or self._args[0] is not want[0] or self._args[1:3] != want[1:3] or self._args[3] is not want[3]:
            # everything but the stream is fixed while the resident model is: bind the arguments
            # of every slot once, the per-frame call then passes two (ctypes spends ~0.3 us per
            # argument; twelve per frame were a third of the host's time per frame)
            tri, col, nrm = filler._inputs
            o = filler._order
            for plan in self.plans:
                _capi.check(self.lib.crender_plan_set_triangle_order(
                    plan, None if o is None else o[0].data_ptr(), None if o is None else o[1].data_ptr()),
                    "crender_plan_set_triangle_order")
                _capi.check(self.lib.crender_plan_set_normal_z(plan, None if o is None else o[2].data_ptr()),
                            "crender_plan_set_normal_z")
            guro = 0
            if filler._fused_light is not None:
                guro = _capi.FUSED_GURO
                for plan in self.plans:
                    _capi.check(self.lib.crender_plan_set_light(plan, (C.c_float * 3)(*filler._fused_light)),
                                "crender_plan_set_light")
            # (a chain of depth 1 runs its frames one after another: lone frames, dispatched as such)
            overlapped = _capi.OVERLAPPED_FRAMES if self.depth > 1 else 0
            for k, (z, c, n, w) in enumerate(self.sets):
                filler._ext.pipeline_bind(self.handle.value, k, tri, col, nrm, filler._P_t, z, c, n, w,
                                          _capi.FUSED_CLEAR | overlapped | guro | want[1])
            self._args = want
        if _current_device() == self._index:
            self._submit(self._handle_int, self._index)
        else:                              # the library launches on the calling thread's device
            with torch.cuda.device(self.device):
                self._submit(self._handle_int, self._index)
        # the filler's buffers are now this frame's
        filler.z_buffer, filler.color_buffer, filler.normals_buffer, filler.winner_buffer = self.sets[self.k]
        self.k = (self.k + 1) % self.depth
        self.n += 1
        self.pending = True

    def join(self, filler):
        _capi.check(self.lib.crender_pipeline_join(self.handle, filler._stream()), "crender_pipeline_join")
        self.pending = False
        self.k = 0                 # the library restarts its frame parity at a join

    def overflow(self, filler):
        """Synchronising check of every plan's bin lists: None, or (direct bins?, entries needed)
        of the worst plan."""
        self.join(filler)
        worst = None
        for plan in self.plans:
            need, cap = C.c_int64(), C.c_int64()
            _capi.check(self.lib.crender_plan_last_bin_usage(plan, filler._stream(), C.byref(need),
                                                             C.byref(cap)), "crender_plan_last_bin_usage")
            if need.value > cap.value and (worst is None or need.value > worst[1]):
                worst = (bool(self.lib.crender_plan_last_frame_direct(plan)), need.value)
        return worst

    def overflowed(self, filler):
        return self.overflow(filler) is not None

    def timing_begin(self, max_frames):
        _capi.check(self.lib.crender_pipeline_timing_begin(self.handle, int(max_frames)),
                    "crender_pipeline_timing_begin")

    def timing_end(self):
        """-> (frames, average ms between the events around each frame's launches on its stream)."""
        n, ms = C.c_int(), C.c_double()
        _capi.check(self.lib.crender_pipeline_timing_end(self.handle, C.byref(n), C.byref(ms)),
                    "crender_pipeline_timing_end")
        return n.value, ms.value


class AdvancedPixelBufferFiller:
    def __init__(self, h, w, fov=90.0, z_near=0.1, z_far=1000.0, n_threads=1, *,
                 device=None, tile=0, row_strip=None, track_winner=False, cache_inputs=False,
                 bin_capacity=0, direct_bins=True, pipeline=False, pipeline_depth=None,
                 presort=None, lookahead=None, raster_path=None):
        self._lib = _capi.load()                      # raises if the HIP library is missing
        self._ext = _torch_ext.load()                 # raises if the torch extension is not built
        if not torch.cuda.is_available():
            raise _capi.CrenderError("AdvancedPixelBufferFiller needs a ROCm GPU (no CPU fallback)")
        self.h, self.w = int(h), int(w)
        self.fov, self.z_near, self.z_far = float(fov), float(z_near), float(z_far)
        self.n_threads = n_threads                    # accepted for API parity, unused
        self.device = torch.device(device if device is not None else "cuda:0")
        self.tile = int(tile)
        self.y0, self.y1 = (0, self.h) if row_strip is None else (int(row_strip[0]), int(row_strip[1]))
        self.cache_inputs = cache_inputs
        self._bin_request = int(bin_capacity)   # 0 = let the library size the bin lists
        self._bin_floor = 0                     # what an overflow taught us this scene needs

        P = (C.c_float * 16)()
        _capi.check(self._lib.crender_projection_matrix(self.fov, self.z_near, self.z_far,
                                                        self.h, self.w, P), "crender_projection_matrix")
        self._P = P
        self.proj_mat = np.array(P[:], dtype=np.float32).reshape(4, 4)
        self._P_t = torch.from_numpy(self.proj_mat.reshape(16).copy())      # host tensor for the extension

        with torch.cuda.device(self.device):
            # same initial state as __cinit__ (.pyx:65-67)
            self.z_buffer = torch.full((self.h, self.w), 1e6, dtype=torch.float32, device=self.device)
            self.color_buffer = torch.zeros((self.h, self.w, 3), dtype=torch.float32, device=self.device)
            self.normals_buffer = torch.zeros((self.h, self.w, 3), dtype=torch.float32, device=self.device)
            self.winner_buffer = (torch.full((self.h, self.w), -1, dtype=torch.int32, device=self.device)
                                  if track_winner else None)
        self._plan = C.c_void_p()
        self._plan_max_T = -1
        self._plan_capacity = 0
        self._workspace = None
        self._inputs = None            # (tri, col, nrm) device tensors of the last frame
        self._inputs_private = False   # they are this filler's own copies (uploaded numpy arrays, or
                                       # the tile-coherent copy): nobody else can change their contents
        self._input_key = None
        self._last_flags = 0
        self._extra_flags = 0 if direct_bins else _capi.NO_DIRECT_BINS
        self._host = {}                # name -> numpy mirror handed out by a getter (view of _host_pin[name])
        self._host_pin = {}            # name -> pinned host tensor behind the mirror
        self._host_fresh = False       # mirrors equal the device buffers
        self._host_exposed = False     # a mirror was handed out and may have been edited
        self._stages = []              # numpy inputs' way up: up to two _Stage (pinned + device [3, T, 3, 3])
        self._inputs_stage = None      # the staging slot the resident inputs live in, if any
        self._model_ref = None         # weak reference to a generation-counting model behind the resident inputs,
        self._model_generation = None  # and the generation the resident (possibly sorted) copy was taken at
        self._sort_cache = None        # (key of caller-owned tensors, sorted inputs, order)
        self._pending = []             # frames launched whose bin lists have not been verified, oldest first
        self._redone = False
        # Tile-coherent copy of large models (crender_plan_set_triangle_order): None = from 2^18
        # triangles on, True / False = always / never.  Made once per upload; results do not change.
        self._presort = presort
        # swap chain: the launch that rasterizes a frame also bins the slot's next frame into a second
        # plan (crender_pipeline_set_lookahead).  None = for scenes that fit the direct bins.
        self._lookahead = lookahead
        # Tile size of the swap chain's plans; None = the chain's own choice (_chain_tile): with `tile`
        # left to the library, frames of a stream run on 32-pixel tiles whatever the frame size —
        # T-Rex 1024 x 1024: 9 % more frames per second than on the 16-pixel tiles a frame rendered
        # alone gets (a quarter of the workgroups).  (bench.py's A/B sets it.)
        self._pipeline_tile = None
        # Which raster kernel the plans' frames get (crender_plan_set_raster_path): None = each plan chooses
        # by the size classes its previous frames counted; 0 / 1 = general / pixel owners.
        # Speed only: every kernel renders every tile exactly.
        self._raster_path = raster_path
        self._order = None             # (orig_of, pos_of) int32 device tensors of the resident inputs
        self._plan_order = None        # what the single-stream plan currently holds
        self._fused_light = None       # (l0, l1, l2): illumination fused into cleared frames
        self._plan_light = None        # what the single-stream plan currently holds
        self._pipeline = bool(pipeline)  # render_frame(): overlap consecutive frames (see _FramePipeline)
        if not pipeline_depth:
            # measured on MI355X (scripts/ab_depth.sh): three frames in flight, or four for small
            # frames when the HIP runtime may use more than its default of 4 hardware queues
            # (GPU_MAX_HW_QUEUES >= 6, read by the runtime when it starts) — with 4 queues a
            # fourth stream shares a queue with another one and the frames serialise
            try:
                queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
            except ValueError:
                queues = 4
            pipeline_depth = 4 if (self.h * self.w <= 1024 * 1024 and queues >= 6) else 3
        self._pipeline_depth = max(1, min(8, int(pipeline_depth)))   # (1: frames one after another on ONE stream of the chain)
        self._pipe = None
        self._checking = False

    # ------------------------------------------------------------------ plumbing --
    def __del__(self):
        try:
            if getattr(self, "_pipe", None):
                self._pipe.close()
                self._pipe = None
            if getattr(self, "_plan", None):
                self._lib.crender_plan_destroy(self._plan)
                self._plan = C.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ensure_plan(self, T, capacity=None):
        capacity = self._bin_request if capacity is None else capacity
        if self._plan and T <= self._plan_max_T and capacity <= self._plan_capacity:
            return
        if self._plan:
            if self._pending and not self._checking:
                self._check_bins()         # (may replace the plan itself: start over)
                return self._ensure_plan(T, capacity)
            torch.cuda.current_stream(self.device).synchronize()
            self._lib.crender_plan_destroy(self._plan)
            self._plan = C.c_void_p()
        max_T = max(int(T), 1)
        nbytes = self._lib.crender_plan_workspace_bytes(self.h, self.w, self.y0, self.y1, max_T,
                                                        int(capacity), self.tile)
        if nbytes == 0:
            raise _capi.CrenderError("bad frame geometry for the tile rasterizer: "
                                     f"h={self.h} w={self.w} strip=({self.y0},{self.y1})")
        self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        plan = C.c_void_p()
        _capi.check(self._lib.crender_plan_create(C.byref(plan), self.h, self.w, self.y0, self.y1,
                                                  max_T, int(capacity), self.tile,
                                                  self._workspace.data_ptr(), nbytes, self._stream()),
                    "crender_plan_create")
        if self._raster_path is not None:
            _capi.check(self._lib.crender_plan_set_raster_path(plan, int(self._raster_path)),
                        "crender_plan_set_raster_path")
        self._plan = plan
        self._plan_max_T = max_T
        # a fresh plan has no triangle order and no light, whatever the address the allocator gave
        # it (a same-sized new after delete usually returns the old one)
        self._plan_order = self._plan_light = None
        need, cap = C.c_int64(), C.c_int64()
        _capi.check(self._lib.crender_plan_last_bin_usage(plan, self._stream(), C.byref(need),
                                                          C.byref(cap)), "crender_plan_last_bin_usage")
        self._plan_capacity = cap.value

    def debug_check(self):
        """Settles everything in flight, then checks the cross-frame state of every plan this filler owns
        (crender_plan_debug_check): raises CrenderError with the findings."""
        self.synchronize()
        with torch.cuda.device(self.device):
            for plan in ([self._plan] if self._plan else []) + (list(self._pipe.plans) if self._pipe is not None else []):
                _capi.plan_debug_check(plan, self._stream())

    def last_raster_paths(self):
        """Which raster kernel the most recent launch of every live plan was (crender_plan_last_raster_path:
        0 general, 1 pixel owners): the single-stream plan first, then the swap chain's."""
        plans = ([self._plan] if self._plan else []) + (list(self._pipe.plans) if self._pipe is not None else [])
        return [int(self._lib.crender_plan_last_raster_path(p)) for p in plans]

    def _chain_tile(self):
        """Tile size of the swap chain's plans: the caller's `tile` if one was given; else 32-pixel
        tiles (the library's own choice above 1024 x 1024, and the faster one for a STREAM of smaller
        frames too), 16 for frames too small to fill the chip with them."""
        if self._pipeline_tile is not None:
            return int(self._pipeline_tile)
        if self.tile:
            return self.tile
        return 32 if self.h * self.w >= 512 * 512 else 16

    def _join_pipe(self):
        """Pipelined frames run on the pipeline's streams: make the current stream wait for them
        before anything else reads or writes the buffers."""
        if self._pipe is not None and self._pipe.pending:
            self._pipe.join(self)

    def _push_host_edits(self):
        """Carry in-place edits of handed-out numpy views back to the device."""
        self._join_pipe()
        if not self._host_exposed:
            return
        for name, buf in (("z", self.z_buffer), ("color", self.color_buffer),
                          ("normals", self.normals_buffer)):
            if name in self._host_pin:
                buf.copy_(self._host_pin[name], non_blocking=True)    # (pinned: one DMA, stream-ordered)
        self._host_exposed = False

    def _win_ptr(self):
        return self.winner_buffer.data_ptr() if self.winner_buffer is not None else None

    def _launch(self, flags, inputs=None, private=False, generation=None, stage=None):
        self._join_pipe()
        if inputs is not None:
            self._inputs, self._order = self._tile_coherent(inputs, bool(private), generation)
            self._inputs_private = bool(private) or self._order is not None
            # (a tile-coherent copy is a tensor of its own: the staging slot is free again)
            self._inputs_stage = stage if self._order is None else None
        if not self._checking:
            if flags & _capi.FUSED_CLEAR:
                # whatever an earlier frame dropped is overwritten by this one
                self._pending.clear()
            elif self._pending:
                # This frame composites on top of the pending ones.  One of those that overflowed its
                # bin lists has to be redone from ITS OWN inputs, in its place in the sequence (a later
                # call must win equal depths): looked at now, without waiting, if its record has landed;
                # otherwise the check is put off — the pending frames keep their inputs alive, and
                # _settle replays the whole sequence in order if it finds an overflow later.  Inputs
                # somebody else may rewrite (the caller's device tensors) cannot be kept: wait for them.
                self._settle(block=not all(f.private for f in self._pending) or len(self._pending) >= 6)
        self._submit(flags)

    def _submit(self, flags):
        """Launch one frame of the resident inputs on the single-stream plan; the frame joins the
        pending list with its number on the plan (crender_plan_frame_ticket)."""
        tri, col, nrm = self._inputs
        T = tri.shape[0]
        self._ensure_plan(T)
        if self._plan_order != (self._plan.value, id(self._order)):
            o = self._order
            _capi.check(self._lib.crender_plan_set_triangle_order(
                self._plan, None if o is None else o[0].data_ptr(), None if o is None else o[1].data_ptr()),
                "crender_plan_set_triangle_order")
            _capi.check(self._lib.crender_plan_set_normal_z(self._plan, None if o is None else o[2].data_ptr()),
                        "crender_plan_set_normal_z")
            self._plan_order = (self._plan.value, id(o))
        flags &= ~_capi.FUSED_GURO
        if (flags & _capi.FUSED_CLEAR) and self._fused_light is not None:
            if self._plan_light != (self._plan.value, self._fused_light):
                _capi.check(self._lib.crender_plan_set_light(self._plan, (C.c_float * 3)(*self._fused_light)),
                            "crender_plan_set_light")
                self._plan_light = (self._plan.value, self._fused_light)
            flags |= _capi.FUSED_GURO
        # (the extension unwraps the tensors, takes torch's current stream of their device and
        # calls crender_render_model; it returns the frame's number on the plan)
        ticket = self._ext.render_model(self._plan.value, tri, col, nrm, self._P_t, self.z_buffer, self.color_buffer,
                                        self.normals_buffer, self.winner_buffer, flags | self._extra_flags)
        self._last_flags = flags
        self._host_fresh = False
        self._pending.append(_Frame(ticket, flags, self._inputs, self._order, self._inputs_private,
                                    self._fused_light, self._inputs_stage))

    def _tile_coherent(self, inputs, private, generation=None):
        """Large models are kept in HBM in tile-coherent order: sorted, once per upload, by the
        Morton code of the screen tile each triangle's centroid projects to, so that the raster
        kernel's gathers by list entry and by winning triangle read neighbouring records instead
        of 36-byte needles out of a gigabyte (10 M small triangles: 4.9 GB of HBM traffic per frame
        for 1.5 GB of algorithmic bytes before).  The kernels keep speaking the caller's indices
        (depth ties, winner plane) through the two index arrays returned with the sorted copies.

        Who gets sorted: by default (``presort=None``) arrays whose every change this filler gets to
        know — numpy inputs it uploaded itself, and the arrays of a model that counts its own
        changes (``DeviceModel.generation``) — from 2^18 triangles on.  Bare device tensors handed in
        by the caller are used in place as they are (``render_frame`` then sees what the caller
        writes into them between frames, as for small models).  ``presort=True`` sorts those too,
        which makes the resident copy a SNAPSHOT taken at every ``render_model`` / ``render_arrays``
        call: re-sorted each time, because nothing tells the filler whether a tensor was rewritten —
        torch's version counter misses every write through ``data_ptr()`` (a HIP kernel, ctypes).
        ``presort="static"`` is the caller's promise that only torch writes the tensors: the
        permutation and the sorted copy are then cached per (address, shape, torch version counter)
        and redone after a torch in-place write.  A generation-counting model is cached per
        (address, shape, generation) under any policy.  Cost of one sort at 10 M triangles: keys +
        radix sort + three gathers, about 1 GB of traffic, ~2 ms."""
        tri, col, nrm = inputs
        T = tri.shape[0]
        tracked = generation is not None
        want = bool(self._presort) if self._presort is not None else ((private or tracked) and T >= (1 << 18))
        if not want or T < 2 or T >= (1 << 31):
            return inputs, None
        key = None
        if not private:
            if tracked:
                key = ("generation", generation) + tuple((a.data_ptr(), tuple(a.shape)) for a in inputs)
            elif self._presort == "static":
                key = tuple((a.data_ptr(), tuple(a.shape), a._version) for a in inputs)
            if key is not None and self._sort_cache is not None and self._sort_cache[0] == key:
                return self._sort_cache[1], self._sort_cache[2]
        keys = torch.empty(T, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_tile_order_keys(tri.data_ptr(), T, self._P, self.w, self.h,
                                                          keys.data_ptr(), self._stream()),
                        "crender_tile_order_keys")
            perm = torch.sort(keys, stable=True).indices          # (device radix sort: upload-time plumbing)
            del keys
            sorted_inputs = tuple(a.index_select(0, perm) for a in (tri, col, nrm))
            orig_of = perm.to(torch.int32)
            pos_of = torch.empty_like(orig_of)
            pos_of[perm] = torch.arange(T, dtype=torch.int32, device=self.device)
            # the normals' z components apart (crender_plan_set_normal_z): the binning pass's back-face
            # test then reads 12 contiguous bytes per triangle, not every line of the normal array
            nz = sorted_inputs[2][:, :, 2].contiguous()
        order = (orig_of, pos_of, nz)
        self._sort_cache = None if key is None else (key, sorted_inputs, order)
        return sorted_inputs, order

    def _check_bins(self):
        """Synchronise the stream; if a pending frame overflowed its bin lists, grow them and redo it.
        Returns True if a frame was rendered again (the buffers changed since the call began)."""
        self._redone = False
        self._settle(block=True)
        return self._redone

    def _poll(self, frame):
        """None while the frame's usage record has not landed, else (entries needed, capacity)."""
        need, cap = C.c_int64(), C.c_int64()
        rc = self._lib.crender_plan_poll_bin_usage(self._plan, frame.ticket, C.byref(need), C.byref(cap))
        if rc == _capi.EBUSY:
            return None
        _capi.check(rc, "crender_plan_poll_bin_usage")
        return need.value, cap.value

    def _settle(self, block):
        """Verify the pending frames, oldest first, from the records their raster launches left in the
        plan's pinned host memory (crender_plan_poll_bin_usage: no copy, no event).  block=False:
        stop at the first frame whose record has not landed; block=True: synchronise the stream
        first (once), so that every record has.  A frame that overflowed its bin lists is rendered
        again — with every frame after it, in order: re-rendering is exact, the result is a
        per-pixel minimum over the prior value and all fragments in which a later frame wins equal
        depths, so fragments that already landed change nothing and the replayed sequence restores
        who wins a tie."""
        if self._checking:
            return
        self._checking = True
        try:
            synced = False
            if block:
                self._settle_pipe()
                torch.cuda.current_stream(self.device).synchronize()
                synced = True
            while self._pending:
                got = self._poll(self._pending[0])
                if got is None:
                    if not block:
                        return
                    if synced:      # (launched on another stream than today's current one)
                        torch.cuda.synchronize(self.device)
                        got = self._poll(self._pending[0])
                        if got is None:
                            raise _capi.CrenderError("a finished frame left no bin-usage record")
                    continue
                need, cap = got
                if need <= cap:
                    self._pending.pop(0)
                    continue
                # overflow: grow, then replay this frame and everything launched on top of it
                frames, self._pending = self._pending, []
                if self._lib.crender_plan_last_frame_direct(self._plan) and not (self._extra_flags & _capi.NO_DIRECT_BINS):
                    # this scene does not fit the small-scene direct bins: general path from now on
                    self._extra_flags |= _capi.NO_DIRECT_BINS
                else:
                    self._bin_floor = max(self._bin_floor, int(need * 1.25) + 1024)
                    self._ensure_plan(max(f.inputs[0].shape[0] for f in frames), capacity=self._bin_floor)
                keep = (self._inputs, self._order, self._inputs_private, self._fused_light, self._inputs_stage)
                for f in frames:
                    self._inputs, self._order, self._inputs_private, self._fused_light, self._inputs_stage = \
                        f.inputs, f.order, f.private, f.light, f.stage
                    self._submit(f.flags)
                self._inputs, self._order, self._inputs_private, self._fused_light, self._inputs_stage = keep
                self._redone = True
                block = True
                torch.cuda.current_stream(self.device).synchronize()
                synced = True
        finally:
            self._checking = False

    def _settle_pipe(self):
        if self._pipe is not None and self._pipe.n > 0:
            need = self._pipe.overflow(self)
            if need:
                # rare: the swap chain's bin lists were too small for this scene.  Remember the
                # larger size (the chain is rebuilt with it at the next pipelined frame) and redo
                # the frame on the plain path.
                direct, entries = need
                if direct:
                    self._extra_flags |= _capi.NO_DIRECT_BINS
                else:
                    self._bin_floor = max(self._bin_floor, int(entries * 1.25) + 1024)
                torch.cuda.synchronize(self.device)
                self._pipe.close()
                self._pipe = None
                self._redone = True
                self._pending.clear()
                self._submit(_capi.FUSED_CLEAR)
            else:
                self._pipe.n = 0

    # ------------------------------------------------------------- reference API --
    def get_size(self):
        return self.h, self.w

    def render_model(self, model, refresh=False, clear=False, refresh_views=True):
        """Project and rasterize ``model`` on top of the current buffers (.pyx:92-104);
        ``clear=True`` (extension) renders into freshly initialised buffers in the same pass.
        ``refresh_views=False`` (extension) leaves the numpy arrays handed out earlier stale until
        the next getter call — for callers that go on working on the device first (Renderer)."""
        src = (model._vertices_by_triangles, model._colors_by_triangles, model._normals_by_triangles)
        # a model that rewrites its arrays in place (DeviceModel: HIP kernels through raw pointers,
        # invisible to torch's version counter) counts its changes itself
        generation = getattr(model, "generation", None)
        key = tuple((id(a), getattr(a, "shape", None)) for a in src) + (generation,)
        private = not any(isinstance(a, torch.Tensor) for a in src)
        self._upload_stage = None
        if refresh or not self.cache_inputs or key != self._input_key:
            inputs = self._upload(src, ("model._vertices_by_triangles", "model._colors_by_triangles",
                                        "model._normals_by_triangles"), composite=not clear)
            self._input_key = key if self.cache_inputs else None
            self._input_refs = src         # keep ids alive while the key is cached
        else:
            inputs = None
        if clear:
            self._host_exposed = False
        else:
            self._push_host_edits()
        self._launch(_capi.FUSED_CLEAR if clear else 0, inputs, private=private, generation=generation,
                     stage=self._upload_stage)
        try:
            self._model_ref = weakref.ref(model) if generation is not None else None
        except TypeError:
            self._model_ref = None
        self._model_generation = generation
        if self._host and refresh_views:
            # arrays handed out earlier are views of the reference's own buffers there: they show
            # this render too
            self._refresh_mirrors()

    def _upload(self, src, names, composite):
        """The three model arrays as [T, 3, 3] float32 device tensors.  Device tensors are taken as
        they are; numpy arrays go through the filler's pinned staging buffer — three host copies
        into it (any strides), ONE asynchronous host-to-device copy — into device tensors the
        filler owns and reuses from call to call."""
        if any(isinstance(a, torch.Tensor) for a in src) or any(a is None for a in src):
            inputs = tuple(_as_device_f32(a, n, self.device) for a, n in zip(src, names))
            if not (inputs[0].shape == inputs[1].shape == inputs[2].shape):
                raise ValueError("vertex, colour and normal arrays must have the same shape")
            return inputs
        arrs = [_check_host_f32(a, n) for a, n in zip(src, names)]
        if not (arrs[0].shape == arrs[1].shape == arrs[2].shape):
            raise ValueError("vertex, colour and normal arrays must have the same shape")
        T = arrs[0].shape[0]
        self._join_pipe()
        st = self._free_stage(T, composite)
        for k in range(3):
            np.copyto(st.host[k], arrs[k])
        with torch.cuda.device(self.device):
            st.dev.copy_(st.pin, non_blocking=True)
            st.done.record(torch.cuda.current_stream(self.device))
        st.busy = True
        self._upload_stage = st
        return (st.dev[0], st.dev[1], st.dev[2])

    def _free_stage(self, T, composite):
        """A staging slot for T triangles that may be overwritten now: its last copy has left the
        host, and no pending frame this one composites on still needs its device side for a redo.
        (Frames in flight that merely READ the device side are ordered before the new copy by the
        stream.)  Another slot is made only when none is free — a caller that renders faster than the
        GPU — up to four for small models, two for large ones; with all taken, the pending frames are
        waited for (back-pressure: the one place where this path synchronises)."""
        if composite and self._pending:
            self._settle(block=False)                    # frames whose records have landed need nothing any more
        for attempt in range(2):
            held = {id(f.stage) for f in self._pending} if composite else set()
            for st in self._stages:
                if id(st) in held or (st.busy and not st.done.query()):
                    continue
                st.busy = False
                if st.cap < T:
                    self._stages[self._stages.index(st)] = st = _Stage(T, self.device)
                return st.shape(T)
            if len(self._stages) < (4 if T * 108 <= (8 << 20) else 2):
                self._stages.append(_Stage(T, self.device))
                return self._stages[-1].shape(T)
            self._check_bins()                           # synchronises: every copy has left, every frame is verified
        st = self._stages[0] = _Stage(T, self.device)    # (not reached: after a synchronisation every slot is free)
        return st.shape(T)

    # north_star wording; the reference's method is render_model
    render = render_model

    def get_normals_buffer(self):
        return self._mirror("normals", self.normals_buffer)

    def get_color_buffer(self):
        return self._mirror("color", self.color_buffer)

    def get_z_buffer(self):
        return self._mirror("z", self.z_buffer)

    # ------------------------------------------------------------------ extensions --
    def render_arrays(self, tri, col, nrm, clear=False):
        """``render_model`` on explicit [T,3,3] float32 arrays (numpy or torch, any device).
        ``clear=True`` renders into freshly initialised buffers in the same pass."""
        self._upload_stage = None
        inputs = self._upload((tri, col, nrm), ("tri", "col", "nrm"), composite=not clear)
        self._input_key = None
        if clear:
            self._host_exposed = False
        else:
            self._push_host_edits()
        self._launch(_capi.FUSED_CLEAR if clear else 0, inputs,
                     private=not any(isinstance(a, torch.Tensor) for a in (tri, col, nrm)),
                     stage=self._upload_stage)

    def set_fused_illumination(self, light_direction=None):
        """Fuse ``GuroIllumination(light_direction).draw_illumination`` into every frame that starts
        from cleared buffers (``render_arrays(..., clear=True)``, ``render_model(..., clear=True)``,
        ``render_frame``): colour is shaded as it is stored, instead of in a second pass over the
        colour and normal planes.  ``light_direction`` is the illumination object's own
        (flipped, normalised) vector; ``None`` switches the fusion off.  Frames that composite on
        older content are never fused (the reference shades the whole buffer again after every
        render)."""
        self._join_pipe()
        self._fused_light = None if light_direction is None else tuple(float(v) for v in light_direction)

    def render_frame(self, pipelined=None):
        """One benchmark frame: clear + project + rasterize the resident model
        (SURVEY.md section 8d 'one frame').  Inputs must have been set by a previous
        render_model / render_arrays call.  With ``pipeline=True`` (constructor) the filler is
        a swap chain of ``pipeline_depth`` (3 or 4 by default): consecutive frames render into
        rotating framebuffer sets on as many streams and overlap on the GPU; the buffer
        attributes and getters always refer to the most recently submitted frame.

        What "resident" means: device tensors handed in by the caller are read in place, every frame,
        as they are then.  A model of 2^18 triangles or more (or ``presort=True``) is rendered from a
        tile-coherent snapshot instead.  For a model that counts its rewrites (``DeviceModel.generation``
        — part of that class's contract: every method that rewrites the by-triangle arrays bumps it, and
        so must anybody who writes them through ``data_ptr()``: ``touch()``) a moved count is noticed
        here: the snapshot is retaken, the swap chain drops what it binned ahead.  Bare tensors under
        ``presort=True`` keep the snapshot of the last ``render_model`` / ``render_arrays`` call, and
        bare tensors rewritten in place under a swap chain need ``render_model`` again (the protocol of
        ``crender_pipeline_join``).

        numpy arrays handed out by the getters are NOT refreshed by this call (no PCIe traffic in the frame
        loop): they show the frame after the next getter call, and what is written into them before that
        is lost."""
        # The frame starts from cleared buffers: whatever the caller wrote into arrays the getters handed out
        # is void, and those arrays are stale from here until the next getter call refreshes them — they
        # must not be carried back over this frame by a later compositing render (they were, until round 5:
        # getter, render_frame, render_model lost the frame; test_fuzz_a_filler_through_a_random_session).
        self._host_exposed = False
        model = self._model_ref() if self._model_ref is not None else None
        if model is not None and model.generation != self._model_generation:
            # The resident model has rewritten its arrays since (DeviceModel.shift / rotate / scale count
            # their rewrites in `generation`).  Frames of the swap chain wait for the caller's stream
            # again (the rewrite was enqueued there) and what was binned ahead from the old contents is
            # dropped; a tile-coherent SNAPSHOT of the arrays is taken anew.  (The caller still must not
            # rewrite arrays that frames in flight are reading: join() first.)
            # The arrays are bound ANEW whether or not they are sorted: a model may have REPLACED one of them
            # (DeviceModel.set_uniform_color and the texture path make a new colour tensor and count it as a
            # rewrite) — until round 6 an unsorted resident model kept rendering the old tensor here.
            self._join_pipe()
            was_sorted = self._order is not None
            src = (model._vertices_by_triangles, model._colors_by_triangles, model._normals_by_triangles)
            inputs = self._upload(src, ("model._vertices_by_triangles", "model._colors_by_triangles",
                                        "model._normals_by_triangles"), composite=False)
            if was_sorted:
                self._inputs, self._order = self._tile_coherent(inputs, False, model.generation)
                self._inputs_private = self._order is not None
            else:
                self._inputs = inputs
                self._inputs_private = False
            self._inputs_stage = None
            self._input_key = None
            if self._pipe is not None:
                self._pipe._args = None          # every slot is bound again: a binding voids its look-ahead
            self._model_generation = model.generation
        use_pipe = self._pipeline if pipelined is None else (pipelined and self._pipeline)
        if not use_pipe:
            self._launch(_capi.FUSED_CLEAR)      # (joins the pipeline first if frames are pending)
            return
        pipe = self._pipe
        if pipe is None or pipe._args is None or pipe._args[0] is not self._inputs:
            # (first pipelined frame, or new inputs: the swap chain may have to grow)
            T = self._inputs[0].shape[0]
            if pipe is None or T > pipe.max_T:
                if pipe is not None:
                    torch.cuda.synchronize(self.device)
                    pipe.close()
                pipe = self._pipe = _FramePipeline(self, T, self._pipeline_depth)
        pipe.frame(self)
        self._host_fresh = False

    def render_projected_frame(self, proj):
        """One frame — clear + bin + rasterize — from ALREADY PROJECTED vertices (K2 alone,
        ``crender_raster``; .pyx:177-244) with the resident colours and normals: the receiving side
        of north_star's "projected vertices broadcast" layout (distributed.StripRenderer).  `proj`
        is a [T, 3, 3] float32 device tensor in the resident model's triangle order."""
        self._join_pipe()
        self._host_exposed = False         # (a cleared frame: see render_frame)
        tri, col, nrm = self._inputs
        if self._order is not None:
            raise ValueError("render_projected_frame needs the resident model in the caller's order (presort=False)")
        if tuple(proj.shape) != tuple(tri.shape) or proj.dtype != torch.float32 or not proj.is_contiguous():
            raise ValueError("proj must be a contiguous float32 tensor shaped like the resident vertices")
        T = tri.shape[0]
        self._ensure_plan(T)
        if self._plan_order != (self._plan.value, id(None)):
            _capi.check(self._lib.crender_plan_set_triangle_order(self._plan, None, None),
                        "crender_plan_set_triangle_order")
            _capi.check(self._lib.crender_plan_set_normal_z(self._plan, None), "crender_plan_set_normal_z")
            self._plan_order = (self._plan.value, id(None))
        flags = _capi.FUSED_CLEAR | self._extra_flags
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_raster(self._plan, proj.data_ptr(), col.data_ptr(), nrm.data_ptr(), T,
                                                 self.z_buffer.data_ptr(), self.color_buffer.data_ptr(),
                                                 self.normals_buffer.data_ptr(), self._win_ptr(), flags,
                                                 self._stream()), "crender_raster")
        self._last_flags = _capi.FUSED_CLEAR       # (a redo after a bin overflow projects the resident
        self._host_fresh = False                   #  vertices itself: the same pixels)
        self._pending.clear()
        self._pending.append(_Frame(self._lib.crender_plan_frame_ticket(self._plan), _capi.FUSED_CLEAR, self._inputs,
                                    None, self._inputs_private, self._fused_light, self._inputs_stage))

    def clear(self):
        """Back to the state __cinit__ leaves (.pyx:65-67)."""
        self._join_pipe()
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_clear(self.z_buffer.data_ptr(), self.color_buffer.data_ptr(),
                                                self.normals_buffer.data_ptr(), self._win_ptr(),
                                                self.h, self.w, self.y0, self.y1, self._stream()),
                        "crender_clear")
        self._pending.clear()          # (whatever those frames dropped is gone with the rest)
        self._host_fresh = False
        self._host_exposed = False

    def join(self):
        """Order the current stream after any pipelined frames still in flight (no host sync).
        Needed before other stream work touches the buffers, e.g. an all-gather of strips."""
        self._join_pipe()

    def synchronize(self):
        self._check_bins()

    def bin_usage(self):
        need, cap = C.c_int64(), C.c_int64()
        _capi.check(self._lib.crender_plan_last_bin_usage(self._plan, self._stream(), C.byref(need),
                                                          C.byref(cap)), "crender_plan_last_bin_usage")
        return need.value, cap.value

    def timing_begin(self, max_frames):
        """Record HIP events around the binning passes and the raster kernel of the next
        `max_frames` frames (measurement aid used by bench.py)."""
        _capi.check(self._lib.crender_plan_timing_begin(self._plan, int(max_frames)),
                    "crender_plan_timing_begin")

    def timing_end(self):
        """-> (frames, average ms of the binning passes, average ms of the raster kernel)."""
        n, b, r = C.c_int(), C.c_double(), C.c_double()
        _capi.check(self._lib.crender_plan_timing_end(self._plan, self._stream(), C.byref(n),
                                                      C.byref(b), C.byref(r)), "crender_plan_timing_end")
        return n.value, b.value, r.value

    def present_u8(self, flip_rows=True):
        """uint8 [H, W, 3] device tensor of the colour plane, rows flipped — what the
        reference's run.py:26 writes to disk (``image[::-1].astype('uint8')``)."""
        self._check_bins()
        self._push_host_edits()
        out = torch.empty((self.h, self.w, 3), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(self._lib.crender_present_u8(self.color_buffer.data_ptr(), out.data_ptr(),
                                                     self.h, self.w, 1 if flip_rows else 0,
                                                     self._stream()), "crender_present_u8")
        return out

    def get_z_tensor(self):
        self._check_bins()
        return self.z_buffer

    def get_color_tensor(self):
        self._check_bins()
        return self.color_buffer

    def get_normals_tensor(self):
        self._check_bins()
        return self.normals_buffer

    def get_winner_tensor(self):
        self._check_bins()
        return self.winner_buffer

    def _refresh_mirrors(self, only=None):
        """Bring the handed-out arrays up to date with the device buffers: one asynchronous copy per
        plane into its pinned buffer, issued BEFORE the synchronising bin-list check so that one wait
        covers both (a frame that has to be redone — rare — is copied again)."""
        names = [n for n in self._host_pin if only is None or n in only]
        while True:
            self._join_pipe()
            bufs = {"z": self.z_buffer, "color": self.color_buffer, "normals": self.normals_buffer}
            for n in names:
                self._host_pin[n].copy_(bufs[n], non_blocking=True)
            if not self._check_bins():              # synchronises the stream; True = the frame was redone
                break
        if only is None:
            self._host_fresh = True
            # The arrays the caller holds show the buffers again — and are the caller's to write into from
            # here on, getter call or not (they are the reference's buffers themselves, .pyx:246-253): the next
            # compositing render carries them back first.  (Until round 5 only a getter call raised this flag:
            # an edit made after a render_model, into arrays handed out before it, never reached the device.)
            if self._host:
                self._host_exposed = True

    def _mirror(self, name, buf):
        if not self._host_fresh:
            self._refresh_mirrors()
        if name not in self._host:
            with torch.cuda.device(self.device):
                self._host_pin[name] = torch.empty(tuple(buf.shape), dtype=buf.dtype, pin_memory=True)
            self._host[name] = self._host_pin[name].numpy()
            self._refresh_mirrors(only=(name,))
        self._host_exposed = True
        return self._host[name]
