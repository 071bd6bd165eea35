#!/usr/bin/env python3
"""Diagnostic: phase timestamps of k_setup's first batch per workgroup (-DCRENDER_STAMPS build)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cython3dmodelrenderer_amd import _build
dbg_lib = "/tmp/libcrender_hip_stamps.so"
_build.compile_library(dbg_lib, ["-DCRENDER_STAMPS"], quiet=True)
_build.LIB_PATH = dbg_lib
import torch
from cython3dmodelrenderer_amd import _capi, scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
wl = sys.argv[1] if len(sys.argv) > 1 else "trex1024"
tri, col, nrm, (H, W), fov = scenes.scene(wl)
L = _capi.load()
f = AdvancedPixelBufferFiller(H, W, fov=fov)
f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
nb = 4096
buf = torch.zeros(nb * 8, dtype=torch.int64, device="cuda:0")
L.crender_debug_set_setup_stamps.argtypes = [C.c_void_p]; L.crender_debug_set_setup_stamps.restype = C.c_int
for _ in range(3): f.render_frame(pipelined=False)
f.synchronize()
assert L.crender_debug_set_setup_stamps(buf.data_ptr()) == 0
f.render_frame(pipelined=False); f.synchronize()
L.crender_debug_set_setup_stamps(None)
s = buf.cpu().numpy().reshape(nb, 8).astype(np.int64) * 10      # ns
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
names = ["start", "inputs staged", "projected, ranges known", "pass A (LDS counts)", "pass B (global atomics)", "entries issued"]
print(f"{wl}: {len(s)} workgroups; ns since the first workgroup's start (p50 / max)")
for k, nm in enumerate(names):
    ok = s[:, k] > 0
    if ok.any():
        v = s[ok, k] - t0
        print(f"  {nm:28s} p50 {np.percentile(v, 50):7.0f}  max {v.max():7d}  ({int(ok.sum())} workgroups)")
