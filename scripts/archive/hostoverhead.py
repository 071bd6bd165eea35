"""Host issue time vs GPU time per frame (T-Rex 1024^2), plain and pipelined."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
tri, col, nrm, (H, W), fov = scenes.scene("trex1024")
for pipe in (False, True):
    f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=pipe)
    f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
    for _ in range(50): f.render_frame()
    torch.cuda.synchronize()
    for K in (1000,):
        t0 = time.perf_counter()
        for _ in range(K): f.render_frame()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"pipeline={pipe} K={K}: issue {1e6*(t1-t0)/K:.1f} us/frame, total {1e6*(t2-t0)/K:.1f} us/frame")
