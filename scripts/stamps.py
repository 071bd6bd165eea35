#!/usr/bin/env python3
"""Diagnostic: builds a -DCRENDER_STAMPS copy of the library, renders one workload and prints
per-tile phase durations of k_raster (s_memrealtime, 10 ns ticks, printed in ns).  Never used for timing
claims: the stamped build is slower; only the SHARES are read."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cython3dmodelrenderer_amd import _build
dbg_lib = "/tmp/libcrender_hip_stamps.so"
_build.compile_library(dbg_lib, ["-DCRENDER_STAMPS"] + os.environ.get("STAMPS_DEFS", "").split(), quiet=True)
_build.LIB_PATH = dbg_lib
import torch
from cython3dmodelrenderer_amd import _capi, scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
wl = sys.argv[1] if len(sys.argv) > 1 else "trex1024"; tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tri, col, nrm, (H, W), fov = scenes.scene(wl)
if os.environ.get("STAMPS_TRI"):       # keep a slice of the model: "first:count"
    a, b = (int(v) for v in os.environ["STAMPS_TRI"].split(":"))
    tri, col, nrm = tri[a:a + b], col[a:a + b], nrm[a:a + b]
L = _capi.load()
f = AdvancedPixelBufferFiller(H, W, fov=fov, tile=tile)
f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
ts = tile or (16 if H * W <= 1024 * 1024 else 32)   # pick_tile() of abi.hip
nt = ((W + ts - 1) // ts) * ((H + ts - 1) // ts)
nhelp = 3 * min(nt // 8, 128) if ts == 16 else 0      # make_layout(): helper workgroups lead the grid
nwg = nt + nhelp
buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda:0")
L.crender_debug_set_stamps.argtypes = [C.c_void_p]; L.crender_debug_set_stamps.restype = C.c_int
for _ in range(3): f.render_frame()
f.synchronize()
assert L.crender_debug_set_stamps(buf.data_ptr()) == 0
f.render_frame(); f.synchronize()
L.crender_debug_set_stamps(None)
s = buf.cpu().numpy().reshape(nwg, 16).astype(np.int64)
ran = s[:, 0] != 0
print(f"workgroups {nwg} ({nhelp} helper slots, {int(ran[:nhelp].sum())} used); quadrant workgroups {int((s[:, 9] > 0).sum())}")
s = s[ran]; nt = len(s)
# stamps are s_memrealtime ticks (100 MHz, device-wide); shown in ns since the first tile's start
for k in (0, 1, 2, 3, 5, 6):
    s[:, k] *= 10
base = np.full(nt, s[:, 0].min(), np.int64)
empty = s[:, 1] == 0            # empty tiles take the fast path and only stamp start / end
s[empty, 1] = s[empty, 0]; s[empty, 2] = s[empty, 0]
start, ready, swept, end, n = s[:, 0] - base, s[:, 1] - base, s[:, 2] - base, s[:, 3] - base, s[:, 4]
wk = s[:, 11] != 0
if wk.any():
    mhz = (s[wk, 11] - s[wk, 10]) / np.maximum(s[wk, 3] - s[wk, 0], 1) * 1000.0
    print("shader clock of working workgroups (s_memtime / s_memrealtime): p10 %.0f p50 %.0f p90 %.0f MHz" % tuple(np.percentile(mhz, [10, 50, 90])))
print("XCDs:", len(set(s[:, 7].tolist())))
print(f"{wl} tile={ts} tiles={nt} kernel span {end.max()} ns")
print("start (since its XCD's first tile): p50 %d p90 %d max %d | active tiles p50 %d p90 %d max %d" % (
    tuple(np.percentile(start, [50, 90, 100])) + tuple(np.percentile(start[n > 0], [50, 90, 100]))))
print("end   (since its XCD's first tile): p50 %d p90 %d max %d" % tuple(np.percentile(end, [50, 90, 100])))
for name, d in (("init(start->ready)", ready - start), ("sweeps(ready->swept)", swept - ready), ("resolve(swept->end)", end - swept), ("total", end - start)):
    act = n > 0
    print(f"{name:22s} all: p50 {np.percentile(d,50):8.0f} p90 {np.percentile(d,90):8.0f} max {d.max():8d} | active tiles: p50 {np.percentile(d[act],50) if act.any() else 0:8.0f} p90 {np.percentile(d[act],90) if act.any() else 0:8.0f} max {d[act].max() if act.any() else 0:8d}")
ld, qd = s[:, 5] - base, s[:, 6] - base
act = n > 0
for name, d in (("  ready->loads landed", ld - ready), ("  loads->queue built", qd - ld), ("  queue->swept (1st batch sweep + rest)", swept - qd)):
    print(f"{name:40s} active tiles: p50 {np.percentile(d[act],50):8.0f} p90 {np.percentile(d[act],90):8.0f} max {d[act].max():8d}")
order = np.argsort(-(end - start))[:12]
print("slowest tiles: tile list_len start init sweeps resolve end xcc/hwid")
for i in order:
    print(i, n[i], start[i], ready[i] - start[i], swept[i] - ready[i], end[i] - swept[i], end[i], hex(s[i, 5]))
last = np.argsort(-end)[:8]
print("last-finishing tiles:", [(int(i), int(n[i]), int(start[i]), int(end[i])) for i in last])
# sweeps duration vs list length
for lo, hi in ((1, 8), (8, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 100000)):
    m = (n >= lo) & (n < hi)
    if m.any(): print(f"list [{lo},{hi}): tiles {m.sum():5d} sweeps p50 {np.percentile((swept-ready)[m],50):8.0f} max {(swept-ready)[m].max():8d}  resolve p50 {np.percentile((end-swept)[m],50):8.0f}")
# timeline: tiles in flight per slice of the kernel span (active = has a list, empty = fast path)
span = int(end.max()); nb = 24; edges = np.linspace(0, span, nb + 1)
print("timeline (slice start ns: active tiles in flight, empty tiles in flight, tiles started)")
for b in range(nb):
    lo, hi = edges[b], edges[b + 1]
    infl = (start < hi) & (end > lo)
    print("%8d: active %5d empty %5d started %5d" % (lo, (infl & act).sum(), (infl & ~act).sum(),
                                                      ((start >= lo) & (start < hi)).sum()))
