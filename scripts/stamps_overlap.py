#!/usr/bin/env python3
"""Diagnostic: in-kernel timeline of FRAMES IN FLIGHT.  Builds the -DCRENDER_STAMPS library, runs the
swap chain (T-Rex 1024^2 by default) and reads the stamps of the last frame of every slot — the
frames that were on the GPU together: when each frame's workgroups started and ended, how long
empty and covered tiles lived, the phases of the covered ones, and how many workgroups were
resident over time.  The stamped build is slower; read the shape, not the totals."""
import ctypes as C, os, subprocess, sys
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cython3dmodelrenderer_amd import _build
lib = "/tmp/libcrender_hip_stamps.so"
_build.compile_library(lib, ["-DCRENDER_STAMPS"] + os.environ.get("STAMPS_DEFS", "").split(), quiet=True)
_build.LIB_PATH = lib
import torch
from cython3dmodelrenderer_amd import _capi, scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
wl = sys.argv[1] if len(sys.argv) > 1 else "trex1024"
look = None if len(sys.argv) < 3 else sys.argv[2] == "on"
tri, col, nrm, (H, W), fov = scenes.scene(wl)
L = _capi.load()
f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True, lookahead=look)
f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
for _ in range(12): f.render_frame()
f.synchronize()
depth = f._pipe.depth
buf = torch.zeros(8 * 8192 * 16, dtype=torch.int64, device="cuda:0")
L.crender_debug_set_stamps.argtypes = [C.c_void_p]; L.crender_debug_set_stamps.restype = C.c_int
assert L.crender_debug_set_stamps(buf.data_ptr()) == 0
for _ in range(10 * depth + 1): f.render_frame()
torch.cuda.synchronize()
L.crender_debug_set_stamps(None)
s = buf.cpu().numpy().reshape(8, 8192, 16).astype(np.int64)
frames = []
for k in range(depth):
    a = s[k]; a = a[a[:, 0] != 0]
    frames.append(a)
t0 = min(a[:, 0].min() for a in frames)
print(f"{wl}: swap chain of {depth}, look-ahead {f._pipe.lookahead}; times in us since the earliest workgroup start among the {depth} last frames")
order = sorted(range(depth), key=lambda k: frames[k][:, 0].min())
allw = []
for k in order:
    a = frames[k]
    st, en, n = (a[:, 0] - t0) / 100.0, (a[:, 3] - t0) / 100.0, a[:, 4]
    cov = a[:, 1] != 0
    life = en - st
    print(f"slot {k}: {len(a)} workgroups ({int(cov.sum())} covered); first start {st.min():7.2f}, last start {st.max():7.2f}, last end {en.max():7.2f}  (span {en.max()-st.min():.2f})")
    print("   empty tiles   life p50 %.2f p90 %.2f max %.2f | start p50 %.2f" % (*np.percentile(life[~cov], [50, 90, 100]), np.percentile(st[~cov], 50)))
    c = a[cov]
    cs, ce = (c[:, 0] - t0) / 100.0, (c[:, 3] - t0) / 100.0
    rd, ld, qd, sw = ((c[:, j] - c[:, 0]) / 100.0 for j in (1, 5, 6, 2))
    print("   covered tiles life p50 %.2f p90 %.2f max %.2f | start p50 %.2f p90 %.2f | end p50 %.2f p90 %.2f" % (
        *np.percentile(ce - cs, [50, 90, 100]), *np.percentile(cs, [50, 90]), *np.percentile(ce, [50, 90])))
    print("   covered phases (us since the tile's start, p50 / p90): list known %.2f / %.2f, records landed %.2f / %.2f, queue built %.2f / %.2f, swept %.2f / %.2f, end %.2f / %.2f" % (
        *np.percentile(rd, [50, 90]), *np.percentile(ld, [50, 90]), *np.percentile(qd, [50, 90]), *np.percentile(sw, [50, 90]), *np.percentile(ce - cs, [50, 90])))
    allw.append(np.stack([st, en, cov.astype(float)], 1))
w = np.concatenate(allw)
hi = w[:, 1].max()
print("resident workgroups over time (all / covered), every 2 us:")
for t in np.arange(0.0, hi, 2.0):
    live = (w[:, 0] <= t) & (w[:, 1] > t)
    print(f"   t={t:6.1f}: {int(live.sum()):5d} / {int((live & (w[:, 2] > 0)).sum()):5d}")
ends = sorted(float(((a[:, 3] - t0) / 100.0).max()) for a in frames)
print("frame completion times:", " ".join(f"{e:.1f}" for e in ends))
