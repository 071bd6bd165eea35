"""Row-strip sharding of the framebuffer across the GPUs of one node.

Rank g rasterizes rows [g*S, (g+1)*S) of the full frame (S = ceil(H / world)); every rank
holds all triangles.  Strips of a row-major [H, W, C] tensor are contiguous, so the one exchange
step — an all-gather of the finished strips — needs no packing: each plane is gathered in place
with ``all_gather_into_tensor`` (RCCL on ROCm).

Who projects (``project=``), north_star's two variants:
  "local"      (default) the model is replicated and every rank runs K1 itself, fused into its
               binning pass: 36 B/triangle of local traffic, nothing on the links;
  "broadcast"  rank 0 runs K1 (``crender_project``) and broadcasts the projected vertices, 36 B per
               triangle to every rank over xGMI, the others rasterize them (``crender_raster``).
Same pixels either way (the same device function projects); what differs is where 36·T bytes move.

What is exchanged is the caller's choice (``exchange=``), because on MI355X the exchange, not
the rasterization, bounds a sharded frame (one GPU renders 8192 x 8192 in 0.36 ms; 28 B/pixel to
every rank is 1.64 GB received per rank):
  "planes"   z, colour and normal planes, 28 B/pixel — what the reference's three getters expose;
  "color"    the colour plane only, 12 B/pixel — what ``Renderer.render`` returns;
  "present"  the presented image, uint8 BGR with rows flipped (reference run.py:26), 3 B/pixel —
             what the reference's script writes to disk.
``chunks > 1`` cuts every rank's strip into the SAME number of sub-strips (a function of H, world
and chunks alone), rendered one after another, each gathered on a second stream as soon as it is
finished while the next one is rasterized.  Every rank issues exactly that many collectives per
plane, whatever its own share of rows: a rank whose sub-strip is empty (ragged last strip, more
ranks than rows) contributes zero rows.

With a gloo group (CPU rehearsal of the multi-rank path on a box with fewer GPUs than ranks)
device tensors are staged through the host.
"""
from __future__ import annotations

import numpy as np
import weakref

import torch
import torch.distributed as dist


def strip_height(H: int, world: int) -> int:
    return (H + world - 1) // world


def strip_rows(H: int, world: int, rank: int):
    """Rows [y0, y1) owned by `rank`; equal strips of ceil(H/world) rows, the last ones
    clipped to the frame (possibly empty when world > H)."""
    s = strip_height(H, world)
    y0 = min(H, rank * s)
    return y0, min(H, y0 + s)


def chunk_count(H: int, world: int, chunks: int) -> int:
    """Sub-strips per rank: the same on every rank (no rank may issue fewer collectives)."""
    return max(1, min(int(chunks), strip_height(H, world)))


def chunk_height(H: int, world: int, chunks: int) -> int:
    n = chunk_count(H, world, chunks)
    return (strip_height(H, world) + n - 1) // n


def substrip_rows(H: int, world: int, rank: int, chunk: int, chunks: int):
    """Rows of sub-strip `chunk` (of ``chunk_count(H, world, chunks)`` equal blocks) of rank
    `rank`'s strip; possibly empty."""
    c = chunk_height(H, world, chunks)
    y0, y1 = strip_rows(H, world, rank)
    return min(y1, y0 + chunk * c), min(y1, y0 + (chunk + 1) * c)


def _is_gloo(group):
    try:
        return dist.get_backend(group) == "gloo"
    except Exception:
        return False


class Gatherer:
    """All-gathers of row blocks into full-frame tensors, with the staging tensors of the padded
    path allocated ONCE per (shape, dtype, block height) instead of per call."""

    def __init__(self, group=None):
        self.group = group
        self._staging = {}

    def _stage(self, p, block, world, device):
        key = (tuple(p.shape[1:]), p.dtype, block, world, str(device))
        st = self._staging.get(key)
        if st is None:
            pad = torch.zeros((block,) + tuple(p.shape[1:]), dtype=p.dtype, device=device)
            out = torch.empty((world * block,) + tuple(p.shape[1:]), dtype=p.dtype, device=device)
            st = self._staging[key] = (pad, out)
        return st

    def gather_blocks(self, p, rank, world, block, rows_of):
        """Rank r contributes ``p[a:b]`` with ``(a, b) = rows_of(r)`` (at most `block` rows,
        possibly none) and every rank ends up with all of them in place.  When the blocks tile `p`
        in rank order with equal heights the collective writes straight into `p`; otherwise
        (ragged last strip, sub-strips, a flipped image, or a gloo group with device tensors)
        blocks are padded to `block` rows and land through the staging tensors."""
        a, b = rows_of(rank)
        spans = [rows_of(r) for r in range(world)]
        via_host = p.is_cuda and _is_gloo(self.group)
        in_place = (not via_host and all(y1 - y0 == block for y0, y1 in spans)
                    and all(spans[r][0] == r * block for r in range(world)) and spans[-1][1] == p.shape[0])
        if in_place:
            dist.all_gather_into_tensor(p, p[a:b], group=self.group)
            return
        pad, out = self._stage(p, block, world, "cpu" if via_host else p.device)
        if b > a:
            pad[: b - a].copy_(p[a:b])
        dist.all_gather_into_tensor(out, pad, group=self.group)
        # equal, equally spaced blocks: ONE strided copy lands all of them
        step = spans[1][0] - spans[0][0] if world > 1 else 0
        regular = (world > 1 and all(y1 - y0 == block for y0, y1 in spans)
                   and all(spans[r][0] == spans[0][0] + r * step for r in range(world)) and step >= block
                   and spans[0][0] + (world - 1) * step + block <= p.shape[0])
        if regular:
            tail = tuple(p.shape[1:])
            row = p.stride(0)
            dst = p.as_strided((world, block) + tail, (step * row, row) + tuple(p.stride()[1:]),
                               p.storage_offset() + spans[0][0] * row)
            dst.copy_(out.view((world, block) + tail))
            return
        for r, (y0, y1) in enumerate(spans):
            if y1 > y0 and r != rank:
                p[y0:y1].copy_(out[r * block: r * block + (y1 - y0)])


# Staging of the free functions below, one Gatherer per process group for as long as the group object
# lives (weak keys: a destroyed group's staging goes with it, and a new group that happens to get the
# old one's address starts clean); the default group (None) has an entry of its own.  A Gatherer's
# staging tensors carry no stream ordering of their own: use one from ONE stream at a time
# (StripRenderer owns its Gatherer and exchanges on one communication stream).
_default_gatherers = weakref.WeakKeyDictionary()
_world_gatherer = []


def _gatherer(group):
    if group is None:
        if not _world_gatherer:
            _world_gatherer.append(Gatherer(None))
        return _world_gatherer[0]
    try:
        g = _default_gatherers.get(group)
        if g is None:
            g = _default_gatherers[group] = Gatherer(group)
        return g
    except TypeError:            # a group object that cannot be weakly referenced: no caching
        return Gatherer(group)


def gather_blocks(p, rank, world, block, rows_of, group=None):
    _gatherer(group).gather_blocks(p, rank, world, block, rows_of)


def all_gather_strips(planes, H: int, rank: int, world: int, group=None):
    """In-place all-gather of row strips.  `planes` are full-frame tensors [H, ...] whose
    rows strip_rows(H, world, rank) hold this rank's result; on return every rank holds
    every strip.  Frames whose height is a multiple of `world` are gathered with no staging
    copy; otherwise strips are padded to a common height."""
    if world == 1:
        return planes
    s = strip_height(H, world)
    for p in planes:
        assert p.shape[0] == H and p.is_contiguous()
        gather_blocks(p, rank, world, s, lambda r: strip_rows(H, world, r), group)
    return planes


def all_gather_substrips(planes, H: int, rank: int, world: int, chunk: int, chunks: int, group=None):
    """The same for sub-strip `chunk`: block `chunk` of every rank's strip is exchanged (the
    blocks are not adjacent in the frame, so they land through a staging tensor)."""
    if world == 1:
        return planes
    c = chunk_height(H, world, chunks)
    for p in planes:
        gather_blocks(p, rank, world, c, lambda r: substrip_rows(H, world, r, chunk, chunks), group)
    return planes


EXCHANGES = ("planes", "color", "present")
PROJECTIONS = ("local", "broadcast")
EXCHANGE_BYTES_PER_PIXEL = {"planes": 28, "color": 12, "present": 3}


def exchange_bytes_received(exchange: str, H: int, W: int, world: int, rank: int) -> int:
    """Bytes rank `rank` receives per frame under `exchange`: every other rank's rows of the payload."""
    y0, y1 = strip_rows(H, world, rank)
    return EXCHANGE_BYTES_PER_PIXEL[exchange] * (H - (y1 - y0)) * W


def gather_present(gatherer, image, H: int, rank: int, world: int, chunk=None, chunks: int = 1):
    """Exchange of the presented image (uint8 [H, W, 3], rows flipped as run.py:26 writes them): colour
    rows [y0, y1) of a rank's strip (or of its sub-strip `chunk`) are image rows [H - y1, H - y0), so
    the blocks come in DESCENDING row order and land through the staging tensors."""
    def rows_of(r):
        y0, y1 = strip_rows(H, world, r) if chunk is None else substrip_rows(H, world, r, chunk, chunks)
        return H - y1, H - y0
    block = strip_height(H, world) if chunk is None else chunk_height(H, world, chunks)
    gatherer.gather_blocks(image, rank, world, block, rows_of)


class StripRenderer:
    """One rank's share of a sharded frame: a filler restricted to its row strip plus the
    exchange of the finished strips (see the module docstring for ``exchange``, ``chunks`` and
    ``project``)."""

    def __init__(self, h, w, rank, world, fov=90.0, z_near=0.1, z_far=1000.0, device=None,
                 tile=0, group=None, exchange="planes", chunks=1, pipeline=False, project="local"):
        from .pixel_buffer_filler import AdvancedPixelBufferFiller
        if exchange not in EXCHANGES:
            raise ValueError(f"exchange must be one of {EXCHANGES}")
        if project not in PROJECTIONS:
            raise ValueError(f"project must be one of {PROJECTIONS}")
        self.rank, self.world, self.group = rank, world, group
        self.h, self.w = h, w
        self.exchange = exchange
        self.project = project
        y0, y1 = strip_rows(h, world, rank)
        self.empty = y0 >= y1        # more ranks than rows: this rank owns nothing, it only gathers
        self.chunks = chunk_count(h, world, chunks)          # identical on every rank
        dev = torch.device(device if device is not None else "cuda:0")
        self.device = dev
        self._gather = Gatherer(group)
        # one filler per non-empty sub-strip (chunk index -> filler or None); they share the
        # full-frame buffers of the first one.  A rank without rows keeps one whole-frame filler
        # for its buffers only.
        self.by_chunk = []
        first = None
        for k in range(self.chunks):
            a, b = substrip_rows(h, world, rank, k, self.chunks)
            if a >= b:
                self.by_chunk.append(None)
                continue
            f = AdvancedPixelBufferFiller(h, w, fov=fov, z_near=z_near, z_far=z_far, device=dev,
                                          tile=tile, row_strip=(a, b),
                                          pipeline=pipeline and self.chunks == 1 and project == "local",
                                          # (broadcast vertices come in the caller's triangle order)
                                          presort=False if project == "broadcast" else None)
            if first is None:
                first = f
            else:
                f.z_buffer, f.color_buffer, f.normals_buffer = first.z_buffer, first.color_buffer, first.normals_buffer
            self.by_chunk.append(f)
        if first is None:
            first = AdvancedPixelBufferFiller(h, w, fov=fov, z_near=z_near, z_far=z_far, device=dev, tile=tile)
        self.filler = first
        self.fillers = [f for f in self.by_chunk if f is not None]
        self.image = (torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
                      if exchange == "present" else None)
        self._comm = torch.cuda.Stream(device=dev) if self.chunks > 1 else None
        self._proj = None            # project="broadcast": the projected vertices every rank receives
        self._tri = None

    def set_model_arrays(self, tri, col, nrm):
        """Upload the (replicated) model and render a first frame of the strip; synchronises, so
        that a bin-list overflow is found and repaired BEFORE any strip is gathered."""
        for f in self.fillers:
            f.render_arrays(tri, col, nrm, clear=True)
            f.synchronize()
        if self.project == "broadcast":
            self._proj = torch.empty((len(tri), 3, 3), dtype=torch.float32, device=self.device)
            if self.rank == 0:       # K1's input, in the caller's triangle order
                self._tri = (tri if isinstance(tri, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(tri))) \
                    .to(self.device).contiguous()

    def _broadcast_projection(self):
        """north_star's variant: rank 0 runs K1, everybody receives its 36·T bytes."""
        from . import _capi
        import ctypes as C
        f = self.filler
        if self.rank == 0:
            with torch.cuda.device(self.device):
                _capi.check(f._lib.crender_project(self._tri.data_ptr(), self._proj.data_ptr(), self._tri.shape[0],
                                                   f._P, f.w, f.h,
                                                   C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                            "crender_project")
        if _is_gloo(self.group):
            host = self._proj.cpu()
            dist.broadcast(host, src=0, group=self.group)
            self._proj.copy_(host)
        else:
            dist.broadcast(self._proj, src=0, group=self.group)

    def _payload(self, f):
        """Tensors of this rank's rows to exchange, as full-frame tensors."""
        if self.exchange == "planes":
            return [f.z_buffer, f.color_buffer, f.normals_buffer]
        if self.exchange == "color":
            return [f.color_buffer]
        return [self.image]

    def _present_rows(self, f):
        """uint8 image rows of the filler's strip: the image is flipped (run.py:26), so strip rows
        [y0, y1) of the colour plane are image rows [h - y1, h - y0)."""
        from . import _capi
        import ctypes as C
        f.join()
        y0, y1 = f.y0, f.y1
        src = f.color_buffer[y0:y1]
        dst = self.image[self.h - y1: self.h - y0]
        with torch.cuda.device(self.device):
            _capi.check(f._lib.crender_present_u8(src.data_ptr(), dst.data_ptr(), y1 - y0, self.w, 1,
                                                  C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                        "crender_present_u8")

    def _render(self, f):
        if self.project == "broadcast":
            f.render_projected_frame(self._proj)
        else:
            f.render_frame()
        if self.exchange == "present":
            self._present_rows(f)

    def render_frame(self, gather=True):
        """One frame of this rank's strip and (``gather``) the exchange; returns the gathered
        tensors (views of the full-frame buffers every rank then holds)."""
        if self.project == "broadcast":
            self._broadcast_projection()
        if self.chunks == 1:
            f = self.by_chunk[0]
            if f is not None:
                self._render(f)
            if gather:
                if f is not None:
                    f.join()         # the collective runs on the current stream
                if self.exchange == "present":
                    self._gather_present()
                else:
                    for p in self._payload(self.filler):
                        self._gather.gather_blocks(p, self.rank, self.world, strip_height(self.h, self.world),
                                                   lambda r: strip_rows(self.h, self.world, r))
            return self._payload(self.filler)
        # sub-strips: rasterize chunk k + 1 while chunk k is exchanged on the second stream; every
        # rank takes part in every chunk's collective, with or without rows of its own
        cur = torch.cuda.current_stream(self.device)
        for k, f in enumerate(self.by_chunk):
            if f is not None:
                self._render(f)
            if gather:
                done = torch.cuda.Event()
                done.record(cur)
                self._comm.wait_event(done)
                with torch.cuda.stream(self._comm):
                    if self.exchange == "present":
                        self._gather_present(k)
                    else:
                        c = chunk_height(self.h, self.world, self.chunks)
                        for p in self._payload(self.filler):
                            self._gather.gather_blocks(
                                p, self.rank, self.world, c,
                                lambda r, k=k: substrip_rows(self.h, self.world, r, k, self.chunks))
        if gather:
            cur.wait_stream(self._comm)
        return self._payload(self.filler)

    def _gather_present(self, chunk=None):
        gather_present(self._gather, self.image, self.h, self.rank, self.world, chunk, self.chunks)
