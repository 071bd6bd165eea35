#!/usr/bin/env python3
"""Diagnostic: in-kernel timeline of ONE k_frame<32,true> launch as the swap chain of depth 1 dispatches
it — the kernel bench.py's `roofline` prices on the headline workload: frames one after another on one
stream, 32-pixel plans, every covered tile split in four 16 x 16 quadrants (one workgroup each), dispatch
in the previous frame's order, the next frame's binning wavefronts in the launch's first workgroups.
Builds the -DCRENDER_STAMPS library, runs the chain, reads the stamps of the last frame: when workgroups
started and ended, the phases of the covered quadrants (list length known -> records landed -> batch
queue built -> swept -> resolved and stored), and how many workgroups were resident over time.
The stamped build is slower (one s_memrealtime + one store per phase and workgroup): read the shape.
  python scripts/stamps_kframe.py [workload] > gpurun_out/stamps_kframe32_<workload>.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cython3dmodelrenderer_amd import _build
lib = "/tmp/libcrender_hip_stamps.so"
_build.compile_library(lib, ["-DCRENDER_STAMPS"] + os.environ.get("STAMPS_DEFS", "").split(), quiet=True)
_build.LIB_PATH = lib
import torch
from cython3dmodelrenderer_amd import _capi, scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller

wl = sys.argv[1] if len(sys.argv) > 1 else "trex1024"
tri, col, nrm, (H, W), fov = scenes.scene(wl)
L = _capi.load()
f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True, pipeline_depth=1, lookahead=True)
f.render_arrays(tri, col, nrm, clear=True)
f.synchronize()
for _ in range(12):
    f.render_frame()
f.synchronize()
assert f._pipe.lookahead and f._pipe.depth == 1
buf = torch.zeros(8 * 8192 * 16, dtype=torch.int64, device="cuda:0")
L.crender_debug_set_stamps.argtypes = [C.c_void_p]
L.crender_debug_set_stamps.restype = C.c_int
assert L.crender_debug_set_stamps(buf.data_ptr()) == 0
spans = []
for rep in range(5):
    # a burst of frames back to back (a frame right after a synchronisation runs on a GPU whose clocks have
    # dropped: 57-76 us); every frame stamps the same words, the LAST frame's stay
    for _ in range(30):
        f.render_frame()
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(8, 8192, 16).astype(np.int64)[0]
    a = s[s[:, 0] != 0]
    spans.append((a[:, 3].max() - a[:, 0].min()) / 100.0)
L.crender_debug_set_stamps(None)
print(f"{wl}: k_frame<{f._pipe.tile},true>, chain of depth 1 with look-ahead, stamped build; span of the stamped workgroups (first "
      f"start -> last end) in the last frame of five bursts of 30: " + " ".join(f"{v:.2f}" for v in spans) + " us")
print("(workgroups that clear groups of empty tiles and the binning wavefronts of the next frame do not stamp)")
t0 = a[:, 0].min()
st, en, n = (a[:, 0] - t0) / 100.0, (a[:, 3] - t0) / 100.0, a[:, 4]
cov = a[:, 1] != 0
quad = a[:, 9]
print(f"workgroups that stamped: {len(a)} ({int(cov.sum())} with a list: {int((quad[cov] > 0).sum())} quadrants of "
      f"{len(set(a[cov, 8].tolist()))} covered tiles); XCDs {len(set(a[:, 7].tolist()))}")
life = en - st
if (~cov).any():
    print("workgroups without a list (clears): start p50 %.2f p90 %.2f max %.2f | life p50 %.2f p90 %.2f max %.2f | end max %.2f" % (
        *np.percentile(st[~cov], [50, 90, 100]), *np.percentile(life[~cov], [50, 90, 100]), en[~cov].max()))
c = a[cov]
cs, ce = (c[:, 0] - t0) / 100.0, (c[:, 3] - t0) / 100.0
rd, ld, qd, sw = ((c[:, j] - c[:, 0]) / 100.0 for j in (1, 5, 6, 2))
print("covered quadrants: start p50 %.2f p90 %.2f max %.2f | end p50 %.2f p90 %.2f max %.2f | life p50 %.2f p90 %.2f max %.2f" % (
    *np.percentile(cs, [50, 90, 100]), *np.percentile(ce, [50, 90, 100]), *np.percentile(ce - cs, [50, 90, 100])))
print("phases of a covered quadrant, us since ITS start (p50 / p90 / max):")
for name, d in (("list length known", rd), ("first batch's records landed", ld), ("batch queue built", qd),
                ("swept (all batches)", sw), ("resolved and stored (end)", ce - cs)):
    print(f"   {name:32s} {np.percentile(d, 50):6.2f} / {np.percentile(d, 90):6.2f} / {d.max():6.2f}")
print("phase DURATIONS (p50 / p90): counter load %.2f / %.2f, record loads %.2f / %.2f, queue %.2f / %.2f, sweeps %.2f / %.2f, resolve %.2f / %.2f" % (
    *np.percentile(rd, [50, 90]), *np.percentile(ld - rd, [50, 90]), *np.percentile(qd - ld, [50, 90]),
    *np.percentile(sw - qd, [50, 90]), *np.percentile((ce - cs) - sw, [50, 90])))
nl = c[:, 4]
for lo, hi in ((1, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 100000)):
    m = (nl >= lo) & (nl < hi)
    if m.any():
        print(f"   list [{lo},{hi}): {int(m.sum()):4d} quadrant workgroups, sweeps p50 {np.percentile((sw - qd)[m], 50):5.2f} max {(sw - qd)[m].max():5.2f}, "
              f"resolve p50 {np.percentile(((ce - cs) - sw)[m], 50):5.2f}, life p50 {np.percentile((ce - cs)[m], 50):5.2f} max {(ce - cs)[m].max():5.2f}")
# word 12: the FIRST batch's work as the sweep counts it — items of two pixels (run-wise sweep) or, with bit 32, 16-pixel blocks
it = c[:, 12] & 0xFFFFFFFF
blk = (c[:, 12] >> 32) != 0
own = (c[:, 12] == 0)
print(f"first batch's sweep: {int((~blk & ~own).sum())} workgroups run-wise (two-pixel items), {int(blk.sum())} by culled blocks, {int(own.sum())} pixel owners / pixel path")
for lo, hi in ((1, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 100000)):
    m = (~blk & ~own) & (it >= lo) & (it < hi)
    if m.any():
        print(f"   first-batch items [{lo},{hi}): {int(m.sum()):4d} workgroups, sweeps p50 {np.percentile((sw - qd)[m], 50):5.2f} max {(sw - qd)[m].max():5.2f} us")
# words 13 / 14 / 15: inside the first batch's sweep phase — per-record constants in LDS, key plane initialised,
# thread 0's own run walked (the workgroup's other wavefronts may still be walking: "swept" is behind the barrier)
rw_ = (~blk & ~own) & (c[:, 13] != 0) & (c[:, 14] != 0) & (c[:, 15] != 0)
if rw_.any():
    p13, p14, p15 = ((c[rw_, j] - c[rw_, 0]) / 100.0 for j in (13, 14, 15))
    q6, s2 = qd[rw_], sw[rw_]
    one = rw_ & (c[:, 4] <= 256)
    print("inside the sweep phase of run-wise workgroups, durations p50 / p90 (us): constants of the records -> LDS %.2f / %.2f, "
          "key plane %.2f / %.2f, thread 0's run %.2f / %.2f, rest (other wavefronts, later batches, barrier) %.2f / %.2f" % (
              *np.percentile(p13 - q6, [50, 90]), *np.percentile(p14 - p13, [50, 90]), *np.percentile(p15 - p14, [50, 90]),
              *np.percentile(s2 - p15, [50, 90])))
    chunkn = np.ceil(it[rw_] / 256.0)
    for k in (1, 2, 3, 4, 5, 6):
        m = chunkn == k
        if m.any():
            print(f"   {k} item(s) per thread: {int(m.sum()):4d} workgroups, thread 0's run p50 {np.percentile((p15 - p14)[m], 50):5.2f} p90 {np.percentile((p15 - p14)[m], 90):5.2f} us")
last = np.argsort(-ce)[:10]
print("last workgroups to end (tile, quadrant, list, first batch's items or blocks, start, known, landed, queued, swept, end):")
for i in last:
    print("   tile %4d q%d list %4d %s %5d: start %.2f +%.2f +%.2f +%.2f +%.2f end %.2f" % (
        c[i, 8], c[i, 9] - 1, c[i, 4], "blocks" if blk[i] else "items", it[i], cs[i], rd[i], ld[i], qd[i], sw[i], ce[i]))
print("resident workgroups over time (all / covered), every 1 us:")
hi = en.max()
for t in np.arange(0.0, hi, 1.0):
    live = (st <= t) & (en > t)
    print(f"   t={t:5.1f}: {int(live.sum()):5d} / {int((live & cov).sum()):5d}")
