#!/usr/bin/env python3
"""Diagnostic: exactly the driver's headline run shape — the swap chain, W frames, a synchronisation, K frames, a
synchronisation — and nothing after it, so that a rocprofv3 kernel trace of this process ends with the K timed
launches (scripts/timeline.py <dir> K+2).  Prints the wall time of the K frames as bench.py takes it.
  python scripts/k20_probe.py [K] [W] [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Wm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wl = sys.argv[3] if len(sys.argv) > 3 else "trex1024"
tri, col, nrm, (H, W), fov = scenes.scene(wl)
f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True)
f.render_arrays(tri, col, nrm, clear=True)
f.synchronize()
f.render_frame(); f.synchronize()
for rep in range(3):
    for _ in range(Wm):
        f.render_frame()
    f.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stamps = []
    for _ in range(K):
        f.render_frame()
        stamps.append(time.perf_counter())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{wl} K={K} W={Wm} rep {rep}: submit loop {1e6 * (t1 - t0):.1f} us, with the closing synchronisation {1e6 * (t2 - t0):.1f} us "
          f"= {1e6 * (t2 - t0) / K:.2f} us per frame; submits at " + " ".join(f"{1e6 * (s - t0):.0f}" for s in stamps))
