"""Where the time of the drop-in call goes: AdvancedPixelBufferFiller.render_model(model) with the model
arrays in host numpy memory (the call cy/renderer.py:47 makes), on T-Rex 1024^2, cube 256^2 and
bunny 4096^2 — per call: the whole call, and inside it the staging-slot choice (with its
non-blocking look at the pending frames' bin-usage records), the three host copies into pinned
memory, the one host-to-device copy, the launch through the torch extension.  Also: how many calls
had to synchronise (back-pressure), and the same loop with a wait for the GPU after every call.
Run on the GPU box: python scripts/api_call_cost.py > gpurun_out/api_call_cost.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
from cython3dmodelrenderer_amd.pixel_buffer_filler import advanced_pixel_buffer_filler as mod


class M:
    def __init__(self, t, c, n):
        self._vertices_by_triangles, self._colors_by_triangles, self._normals_by_triangles = t, c, n


def wrap(obj, name, acc):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            acc[name + "#"] = acc.get(name + "#", 0) + 1
    setattr(obj, name, timed)
    return fn


for workload in ("trex1024", "cube256", "bunny4096"):
    tri, col, nrm, (H, W), fov = scenes.scene(workload)
    m = M(tri, col, nrm)
    f = AdvancedPixelBufferFiller(H, W, fov=fov)
    for _ in range(5):
        f.render_model(m)
    f.synchronize()
    K = 300 if workload != "bunny4096" else 100
    # ---- the whole call, calls that only enqueue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        f.render_model(m)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{workload}: render_model(model) x {K}: issue {1e6 * (t1 - t0) / K:7.1f} us/call, with the final wait "
          f"{1e6 * (t2 - t0) / K:7.1f} us/call; pending at the end {len(f._pending)}, staging slots {len(f._stages)}")
    # ---- with a wait after every call
    t0 = time.perf_counter()
    for _ in range(K):
        f.render_model(m)
        torch.cuda.synchronize()
    print(f"{workload}: render_model(model) + wait for the GPU, each call: {1e6 * (time.perf_counter() - t0) / K:7.1f} us/call")
    # ---- breakdown (the wrappers cost ~0.3 us each)
    acc = {}
    for name in ("_free_stage", "_settle", "_submit", "_join_pipe", "_push_host_edits", "_tile_coherent"):
        wrap(f, name, acc)
    real_copyto = np.copyto
    wrap(np, "copyto", acc)
    syncs = {"n": 0}
    real_sync = torch.cuda.Stream.synchronize

    def counting_sync(self):
        syncs["n"] += 1
        return real_sync(self)
    torch.cuda.Stream.synchronize = counting_sync
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        f.render_model(m)
    total = time.perf_counter() - t0
    torch.cuda.synchronize()
    torch.cuda.Stream.synchronize = real_sync
    np.copyto = real_copyto
    print(f"{workload}: breakdown over {K} calls, us/call (whole call {1e6 * total / K:7.1f}; stream synchronisations: {syncs['n']})")
    for name in ("_free_stage", "_settle", "copyto", "_submit", "_tile_coherent", "_join_pipe", "_push_host_edits"):
        if name in acc:
            print(f"    {name:18s} {1e6 * acc[name] / K:8.2f}   ({acc[name + '#'] / K:.1f} calls per render)")
    del f

# ---- Renderer.render(model) in its four modes, with the time inside render_model and inside the getter
from cython3dmodelrenderer_amd import Renderer
from cython3dmodelrenderer_amd.illumination import GuroIllumination
tri, col, nrm, (H, W), fov = scenes.scene("trex1024")
m = M(tri, col, nrm)
light = GuroIllumination([0, 0, 1])
for name, mode in (("default", None), ("on_device", True), ("fused", "fused")):
    f = AdvancedPixelBufferFiller(H, W, fov=fov)
    r = Renderer(f, light, None, H, W, on_device=mode)
    for _ in range(5):
        r.render(m)
    acc = {}
    for nm in ("render_model", "get_color_tensor", "get_color_buffer", "_settle", "_free_stage", "_submit", "_poll", "synchronize"):
        wrap(f, nm, acc)
    busy = {"n": 0}
    real_poll = f._lib.crender_plan_poll_bin_usage

    def poll(*a):
        rc = real_poll(*a)
        busy["n"] += rc == 4
        return rc
    f._lib.crender_plan_poll_bin_usage = poll
    K = 100
    t0 = time.perf_counter()
    for _ in range(K):
        r.render(m)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    f._lib.crender_plan_poll_bin_usage = real_poll
    print(f"Renderer.render, on_device={mode!r}: {1e6 * total / K:7.1f} us/call; polls that found no record: {busy['n']}")
    for nm in ("render_model", "_free_stage", "_submit", "get_color_tensor", "get_color_buffer", "synchronize", "_settle", "_poll"):
        if nm in acc:
            print(f"    {nm:18s} {1e6 * acc[nm] / K:8.2f}   ({acc[nm + '#'] / K:.1f} calls per render)")
