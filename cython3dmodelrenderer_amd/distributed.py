"""Row-strip sharding of the framebuffer across the GPUs of one node.

Rank g rasterizes rows [g*S, (g+1)*S) of the full frame (S = ceil(H / world)); every rank
holds all triangles (the model is replicated; re-projecting 36 B/triangle locally is
cheaper than broadcasting it over xGMI).  Strips of a row-major [H, W, C] tensor are
contiguous, so the one exchange step — an all-gather of the finished strips — needs no
packing: each plane is gathered in place with ``all_gather_into_tensor`` (RCCL on ROCm,
gloo in the CPU tests).
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


def all_gather_strips(planes, H: int, rank: int, world: int, group=None):
    """In-place all-gather of row strips.  `planes` are full-frame tensors [H, ...] whose
    rows strip_rows(H, world, rank) hold this rank's result; on return every rank holds
    every strip.  Frames whose height is a multiple of `world` are gathered with no staging
    copy; otherwise strips are padded to a common height."""
    if world == 1:
        return planes
    s = strip_height(H, world)
    y0, y1 = strip_rows(H, world, rank)
    for p in planes:
        assert p.shape[0] == H and p.is_contiguous()
        if H % world == 0:
            dist.all_gather_into_tensor(p, p[y0:y1], group=group)
        else:
            pad = torch.zeros((s,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
            pad[: y1 - y0] = p[y0:y1]
            out = torch.empty((s * world,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
            dist.all_gather_into_tensor(out, pad, group=group)
            p.copy_(out[:H])
    return planes


class StripRenderer:
    """One rank's share of a sharded frame: a filler restricted to its row strip plus the
    all-gather of the three planes."""

    def __init__(self, h, w, rank, world, fov=90.0, z_near=0.1, z_far=1000.0, device=None,
                 tile=0, group=None):
        from .pixel_buffer_filler import AdvancedPixelBufferFiller
        self.rank, self.world, self.group = rank, world, group
        self.h = h
        y0, y1 = strip_rows(h, world, rank)
        self.empty = y0 >= y1        # more ranks than rows: this rank owns nothing, it only gathers
        self.filler = AdvancedPixelBufferFiller(h, w, fov=fov, z_near=z_near, z_far=z_far,
                                                device=device, tile=tile,
                                                row_strip=None if self.empty else (y0, y1))

    def set_model_arrays(self, tri, col, nrm):
        """Upload the (replicated) model and render a first frame of the strip; synchronises, so
        that a bin-list overflow is found and repaired BEFORE any strip is gathered."""
        if self.empty:
            return
        self.filler.render_arrays(tri, col, nrm, clear=True)
        self.filler.synchronize()

    def render_frame(self, gather=True):
        f = self.filler
        if not self.empty:
            f.render_frame()
        if gather:
            if not self.empty:
                f.join()         # the collective runs on the current stream
            all_gather_strips([f.z_buffer, f.color_buffer, f.normals_buffer], self.h, self.rank,
                              self.world, self.group)
