"""Row-strip sharding of the framebuffer across the GPUs of one node.

Rank g rasterizes rows [g*S, (g+1)*S) of the full frame (S = ceil(H / world)); every rank
holds all triangles (the model is replicated; re-projecting 36 B/triangle locally is
cheaper than broadcasting it over xGMI).  Strips of a row-major [H, W, C] tensor are
contiguous, so the one exchange step — an all-gather of the finished strips — needs no
packing: each plane is gathered in place with ``all_gather_into_tensor`` (RCCL on ROCm).

What is exchanged is the caller's choice (``exchange=``), because on MI355X the exchange, not
the rasterization, bounds a sharded frame (one GPU renders 8192 x 8192 in 0.36 ms; 28 B/pixel to
every rank is 1.64 GB received per rank):
  "planes"   z, colour and normal planes, 28 B/pixel — what the reference's three getters expose;
  "color"    the colour plane only, 12 B/pixel — what ``Renderer.render`` returns;
  "present"  the presented image, uint8 BGR with rows flipped (reference run.py:26), 3 B/pixel —
             what the reference's script writes to disk.
``chunks > 1`` cuts a rank's strip into sub-strips rendered one after another, each gathered on a
second stream as soon as it is finished, while the next one is rasterized.

With a gloo group (CPU rehearsal of the multi-rank path on a box with fewer GPUs than ranks)
device tensors are staged through the host.
"""
from __future__ import annotations

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


def _is_gloo(group):
    try:
        return dist.get_backend(group) == "gloo"
    except Exception:
        return False


def gather_blocks(p, rank, world, block, rows_of, group=None):
    """All-gather of one block of rows per rank into the full tensor `p`: rank r contributes
    ``p[a:b]`` with ``(a, b) = rows_of(r)`` (at most `block` rows) and every rank ends up with all
    of them in place.  When the blocks tile `p` in rank order with equal heights the collective
    writes straight into `p`; otherwise (ragged last strip, sub-strips, a flipped image, or a gloo
    group with device tensors) blocks are padded to `block` rows and land through a staging tensor."""
    a, b = rows_of(rank)
    spans = [rows_of(r) for r in range(world)]
    via_host = p.is_cuda and _is_gloo(group)
    in_place = (not via_host and all(y1 - y0 == block for y0, y1 in spans)
                and all(spans[r][0] == r * block for r in range(world)) and spans[-1][1] == p.shape[0])
    if in_place:
        dist.all_gather_into_tensor(p, p[a:b], group=group)
        return
    pad = torch.zeros((block,) + tuple(p.shape[1:]), dtype=p.dtype, device="cpu" if via_host else p.device)
    if b > a:
        pad[: b - a] = p[a:b]
    out = torch.empty((block * world,) + tuple(p.shape[1:]), dtype=p.dtype, device=pad.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    for r, (y0, y1) in enumerate(spans):
        if y1 > y0 and r != rank:
            p[y0:y1].copy_(out[r * block: r * block + (y1 - y0)])


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


def substrip_rows(H: int, world: int, rank: int, chunk: int, chunks: int):
    """Rows of sub-strip `chunk` (of `chunks` equal blocks) of rank `rank`'s strip."""
    c = (strip_height(H, world) + chunks - 1) // chunks
    y0, y1 = strip_rows(H, world, rank)
    return min(y1, y0 + chunk * c), min(y1, y0 + (chunk + 1) * c)


def all_gather_substrips(planes, H: int, rank: int, world: int, chunk: int, chunks: int, group=None):
    """The same for sub-strip `chunk` of `chunks`: block `chunk` of every rank's strip is
    exchanged (the blocks are not adjacent in the frame, so they land through a staging tensor)."""
    if world == 1:
        return planes
    c = (strip_height(H, world) + chunks - 1) // chunks
    for p in planes:
        gather_blocks(p, rank, world, c, lambda r: substrip_rows(H, world, r, chunk, chunks), group)
    return planes


EXCHANGES = ("planes", "color", "present")


class StripRenderer:
    """One rank's share of a sharded frame: a filler restricted to its row strip plus the
    exchange of the finished strips (see the module docstring for ``exchange`` and ``chunks``)."""

    def __init__(self, h, w, rank, world, fov=90.0, z_near=0.1, z_far=1000.0, device=None,
                 tile=0, group=None, exchange="planes", chunks=1, pipeline=False):
        from .pixel_buffer_filler import AdvancedPixelBufferFiller
        if exchange not in EXCHANGES:
            raise ValueError(f"exchange must be one of {EXCHANGES}")
        self.rank, self.world, self.group = rank, world, group
        self.h, self.w = h, w
        self.exchange = exchange
        y0, y1 = strip_rows(h, world, rank)
        self.empty = y0 >= y1        # more ranks than rows: this rank owns nothing, it only gathers
        self.chunks = 1 if self.empty else max(1, min(int(chunks), y1 - y0))
        dev = torch.device(device if device is not None else "cuda:0")
        self.device = dev
        # sub-strip fillers share the full-frame buffers of the first one
        self.fillers = []
        for k in range(self.chunks):
            a, b = (0, h) if self.empty else substrip_rows(h, world, rank, k, self.chunks)
            if not self.empty and a >= b:
                continue
            f = AdvancedPixelBufferFiller(h, w, fov=fov, z_near=z_near, z_far=z_far, device=dev,
                                          tile=tile, row_strip=None if self.empty else (a, b),
                                          pipeline=pipeline and self.chunks == 1)
            if self.fillers:
                f0 = self.fillers[0]
                f.z_buffer, f.color_buffer, f.normals_buffer = f0.z_buffer, f0.color_buffer, f0.normals_buffer
            self.fillers.append(f)
            if self.empty:
                break
        self.filler = self.fillers[0]
        self.image = (torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
                      if exchange == "present" else None)
        self._comm = torch.cuda.Stream(device=dev) if self.chunks > 1 else None

    def set_model_arrays(self, tri, col, nrm):
        """Upload the (replicated) model and render a first frame of the strip; synchronises, so
        that a bin-list overflow is found and repaired BEFORE any strip is gathered."""
        if self.empty:
            return
        for f in self.fillers:
            f.render_arrays(tri, col, nrm, clear=True)
            f.synchronize()

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

    def render_frame(self, gather=True):
        """One frame of this rank's strip and (``gather``) the exchange; returns the gathered
        tensors (views of the full-frame buffers every rank then holds)."""
        if self.chunks == 1:
            f = self.filler
            if not self.empty:
                f.render_frame()
                if self.exchange == "present":
                    self._present_rows(f)
            if gather:
                if not self.empty:
                    f.join()         # the collective runs on the current stream
                if self.exchange == "present":
                    self._gather_present()
                else:
                    all_gather_strips(self._payload(f), self.h, self.rank, self.world, self.group)
            return self._payload(f)
        # sub-strips: rasterize chunk k + 1 while chunk k is exchanged on the second stream
        cur = torch.cuda.current_stream(self.device)
        for k, f in enumerate(self.fillers):
            f.render_frame()
            if self.exchange == "present":
                self._present_rows(f)
            if gather:
                done = torch.cuda.Event()
                done.record(cur)
                self._comm.wait_event(done)
                with torch.cuda.stream(self._comm):
                    if self.exchange == "present":
                        self._gather_present(k)
                    else:
                        all_gather_substrips(self._payload(f), self.h, self.rank, self.world, k,
                                             self.chunks, self.group)
        if gather:
            cur.wait_stream(self._comm)
        return self._payload(self.filler)

    def _gather_present(self, chunk=None):
        # the image is flipped (run.py:26): colour rows [y0, y1) are image rows [h - y1, h - y0)
        H, world = self.h, self.world

        def rows_of(r):
            y0, y1 = strip_rows(H, world, r) if chunk is None else substrip_rows(H, world, r, chunk, self.chunks)
            return H - y1, H - y0
        block = strip_height(H, world) if chunk is None else (strip_height(H, world) + self.chunks - 1) // self.chunks
        gather_blocks(self.image, self.rank, world, block, rows_of, self.group)
