#!/usr/bin/env python3
"""Diagnostic (round 6, the review's item 6): WHERE a framebuffer set lies decides how long T-Rex 8192^2's
store-bound raster launch takes into it (0.30 .. 0.39 ms; profiles/r05/raster_sets_trex8192.txt).  This probe
times the fused clear of an EMPTY frame (every tile's 28 KB of stores, nothing else) into K sets made one of
several ways, one way per process:

  python scripts/placement_probe.py torch  [K]   K sets as torch tensors, one after another (the product's way)
  python scripts/placement_probe.py arena  [K]   ONE allocation taken before anything else on the device, K sets carved from it
  python scripts/placement_probe.py late   [K]   the same arena, taken AFTER the plan's workspace and the model
  python scripts/placement_probe.py reuse  [K]   torch sets; then all freed and made again (the caching allocator hands the blocks back)
  python scripts/placement_probe.py pmc    [K]   torch sets, 6 clears into each in turn and nothing else: run under
                                                 rocprofv3 --kernel-trace --pmc ... (scripts/r6_placement.sh correlates)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H = W = 8192
dev = "cuda:0"
NZ, NC = H * W * 4, H * W * 12
SET = NZ + 2 * NC

arena = None
if mode == "arena":
    arena = torch.empty(K * SET + (4 << 20), dtype=torch.uint8, device=dev)      # the process's first device allocation

from cython3dmodelrenderer_amd import lowlevel as ll          # noqa: E402

P = ll.projection_matrix(45.0, 0.1, 1000.0, H, W)
e = torch.zeros((0, 3, 3), dtype=torch.float32, device=dev)
plan0 = ll.Plan(H, W, 13814, device=dev)                      # (the workspace of a T-Rex plan: ~0.6 GB, as in the product)
if mode == "late":
    arena = torch.empty(K * SET + (4 << 20), dtype=torch.uint8, device=dev)


def carve(k):
    fb = ll.FrameBuffers(8, 8, device=dev, winner=False)
    fb.h, fb.w = H, W
    base = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - arena.data_ptr() + k * SET
    fb.z = arena[base:base + NZ].view(torch.float32).view(H, W)
    fb.color = arena[base + NZ:base + NZ + NC].view(torch.float32).view(H, W, 3)
    fb.normals = arena[base + NZ + NC:base + SET].view(torch.float32).view(H, W, 3)
    return fb


def make_sets():
    if arena is not None:
        return [carve(k) for k in range(K)]
    return [ll.FrameBuffers(H, W, device=dev, winner=False) for _ in range(K)]


def clear_ms(fb, n=20):
    for _ in range(3):
        ll.render_model(plan0, e, e, e, P, fb, clear=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        ll.render_model(plan0, e, e, e, P, fb, clear=True)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


sets = make_sets()
if mode == "pmc":
    for rnd in range(6):
        for fb in sets:
            ll.render_model(plan0, e, e, e, P, fb, clear=True)
    torch.cuda.synchronize()
    print("z planes at " + " ".join(hex(fb.z.data_ptr()) for fb in sets))
    sys.exit(0)
for rnd in range(2):
    print(f"{mode} round {rnd}: ms per empty frame into each of {K} sets: " + " ".join(f"{clear_ms(fb):.4f}" for fb in sets))
print("   z planes at " + " ".join(hex(fb.z.data_ptr()) for fb in sets))
print("   colour planes at " + " ".join(hex(fb.color.data_ptr()) for fb in sets))
if mode == "reuse":
    del sets
    torch.cuda.synchronize()
    sets = make_sets()
    print(f"{mode} after free + allocate again: " + " ".join(f"{clear_ms(fb):.4f}" for fb in sets))
    print("   z planes at " + " ".join(hex(fb.z.data_ptr()) for fb in sets))
    del sets
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                  # the blocks go back to the driver
    sets = make_sets()
    print(f"{mode} after empty_cache + allocate again: " + " ".join(f"{clear_ms(fb):.4f}" for fb in sets))
    print("   z planes at " + " ".join(hex(fb.z.data_ptr()) for fb in sets))
