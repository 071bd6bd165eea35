#!/usr/bin/env python3
"""Diagnostic: the raster launch of one workload, launch by launch — HIP events around each `draw` of the
same binned frame, N launches in a row (default: T-Rex 8192^2, whose launch is store-bound and whose time
is bimodal between boxes and within a process).  Prints the series in chunks, so that one sees whether
the two modes alternate launch by launch, drift with time (clocks, power) or belong to the allocation
(the same series again with the three planes re-allocated at other addresses).
  python scripts/raster_series.py [workload] [launches] > gpurun_out/raster_series_<workload>.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cython3dmodelrenderer_amd import lowlevel as ll, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "trex8192"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
tri, col, nrm, (H, W), fov = scenes.scene(wl)
dev = "cuda:0"
t_tri, t_col, t_nrm = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (tri, col, nrm))
P = ll.projection_matrix(fov, 0.1, 1000.0, H, W)      # (the filler's defaults)
T = tri.shape[0]


def series(tag, pad_bytes):
    # (pad: the planes of this framebuffer set start `pad_bytes` further apart than torch would place them)
    fb = ll.FrameBuffers(H, W, device=dev, winner=False)
    pads = []
    if pad_bytes:
        pads.append(torch.empty(pad_bytes, dtype=torch.uint8, device=dev))
        fb.color = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        pads.append(torch.empty(pad_bytes, dtype=torch.uint8, device=dev))
        fb.normals = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    plan = ll.Plan(H, W, T, device=dev)
    ll.prepare(plan, t_tri, t_nrm, P)
    for _ in range(5):
        ll.draw(plan, t_col, t_nrm, T, fb, clear=True)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    for a, b in ev:
        a.record()
        ll.draw(plan, t_col, t_nrm, T, fb, clear=True)
        b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    addr = [getattr(fb, n).data_ptr() for n in ("z", "color", "normals") if hasattr(fb, n)]
    print(f"{wl} {tag}: {N} raster launches back to back; planes at " + " ".join(hex(a) for a in addr))
    print("   min %.4f p10 %.4f p50 %.4f p90 %.4f max %.4f ms" % (ms.min(), *np.percentile(ms, [10, 50, 90]), ms.max()))
    for i in range(0, N, 25):
        print("   " + " ".join(f"{v * 1e3:4.0f}" for v in ms[i:i + 25]))
    del fb, plan, pads


if os.environ.get("ONE"):
    # the three planes of a set carved from ONE allocation, `pad` bytes between them: is the launch's time then the
    # same for every set, and which pad is the fast one?
    plan = ll.Plan(H, W, T, device=dev)
    ll.prepare(plan, t_tri, t_nrm, P)
    keep = []
    for pad in [int(v) for v in os.environ["ONE"].split(",")]:
        row = []
        for k in range(3):
            nz, nc = H * W * 4, H * W * 12
            big = torch.empty(nz + nc + nc + 2 * pad + 4096, dtype=torch.uint8, device=dev)
            keep.append(big)
            fb = ll.FrameBuffers(8, 8, device=dev, winner=False)
            fb.h, fb.w = H, W
            o1 = nz + pad; o2 = o1 + nc + pad
            fb.z = big[0:nz].view(torch.float32).view(H, W)
            fb.color = big[o1:o1 + nc].view(torch.float32).view(H, W, 3)
            fb.normals = big[o2:o2 + nc].view(torch.float32).view(H, W, 3)
            for _ in range(3):
                ll.draw(plan, t_col, t_nrm, T, fb, clear=True)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(20):
                ll.draw(plan, t_col, t_nrm, T, fb, clear=True)
            b.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(b) / 20)
        print(f"{wl}: planes of a set in ONE allocation, {pad} B between them: ms per raster launch, three such sets: " + " ".join(f"{v:.4f}" for v in row))
    sys.exit(0)
if os.environ.get("SETS"):
    # K framebuffer sets alive at once (distinct memory), the same plan: does the launch's time belong to the SET?
    K = int(os.environ["SETS"])
    plan = ll.Plan(H, W, T, device=dev)
    ll.prepare(plan, t_tri, t_nrm, P)
    sets = [ll.FrameBuffers(H, W, device=dev, winner=False) for _ in range(K)]
    for rnd in range(3):
        row = []
        for fb in sets:
            for _ in range(3):
                ll.draw(plan, t_col, t_nrm, T, fb, clear=True)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(20):
                ll.draw(plan, t_col, t_nrm, T, fb, clear=True)
            b.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(b) / 20)
        print(f"{wl} round {rnd}: ms per raster launch into each of {K} framebuffer sets alive at once: " + " ".join(f"{v:.4f}" for v in row))
    print("   z planes at " + " ".join(hex(fb.z.data_ptr()) for fb in sets))
    print("   colour planes at " + " ".join(hex(fb.color.data_ptr()) for fb in sets))
    # and a pure fill of each set (the fused clear alone: no triangles) — the store pattern without the coverage work
    e = torch.zeros((0, 3, 3), dtype=torch.float32, device=dev)
    plan0 = ll.Plan(H, W, 1, device=dev)
    row = []
    for fb in sets:
        ll.render_model(plan0, e, e, e, P, fb, clear=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            ll.render_model(plan0, e, e, e, P, fb, clear=True)
        b.record()
        torch.cuda.synchronize()
        row.append(a.elapsed_time(b) / 20)
    print(f"{wl}: ms per EMPTY frame (fused clear of every tile) into each set: " + " ".join(f"{v:.4f}" for v in row))
    sys.exit(0)
series("first allocation", 0)
series("second allocation, same sizes", 0)
series("planes 1 MiB + 4 KiB further apart", (1 << 20) + 4096)
series("planes 37 MiB further apart", 37 << 20)
