"""Swap-chain depth sweep (frames in flight) per workload; run with GPU_MAX_HW_QUEUES=8 to see
depths beyond 3 (the runtime's default of 4 hardware queues per process caps the overlap)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
depths = [int(d) for d in (sys.argv[1].split(",") if len(sys.argv) > 1 else "2,3,4".split(","))]
for wl, K in (("trex1024", 2000), ("bunny4096", 200), ("trex8192", 100)):
    tri, col, nrm, (H, W), fov = scenes.scene(wl)
    for depth in depths:
        f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True, pipeline_depth=depth)
        f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
        for _ in range(20): f.render_frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): f.render_frame()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{wl} depth={depth}: issue {1e6*(t1-t0)/K:.1f} us/frame, total {1e6*(t2-t0)/K:.1f} us/frame")
        del f
