"""Diagnostic: bench.py's api_call sequence of Renderer modes with per-call times inside the timed loop."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cython3dmodelrenderer_amd import scenes, Renderer
from cython3dmodelrenderer_amd.illumination import GuroIllumination
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
tri, col, nrm, (H, W), fov = scenes.scene(sys.argv[1] if len(sys.argv) > 1 else "trex1024")
device = torch.device("cuda", 0)
m = bench._Model(tri, col, nrm)
light = GuroIllumination([0, 0, 1])
def dsync(): torch.cuda.synchronize(device)
def timed(call, sync, budget):
    call(); call(); sync()
    t0 = time.perf_counter(); call(); sync(); one = time.perf_counter() - t0
    n = int(max(3, min(200, budget / max(one, 1e-5))))
    ts = []
    t0 = time.perf_counter()
    for _ in range(n):
        t1 = time.perf_counter(); call(); ts.append(time.perf_counter() - t1)
    sync()
    total = time.perf_counter() - t0
    ts = np.array(ts) * 1e3
    return total / n * 1e3, n, ts
f = AdvancedPixelBufferFiller(H, W, fov=fov, device=device)
print("render_model %.3f" % timed(lambda: f.render_model(m), dsync, 3.0)[0])
print("render_model+color %.3f" % timed(lambda: (f.render_model(m), f.get_color_buffer()), dsync, 3.0)[0])
f3 = AdvancedPixelBufferFiller(H, W, fov=fov, device=device)
print("render_model+3 %.3f" % timed(lambda: (f3.render_model(m), f3.get_color_buffer(), f3.get_normals_buffer(), f3.get_z_buffer()), dsync, 3.0)[0])
for name, mode in (("numpy_illumination", False), ("default", None), ("on_device", True), ("fused", "fused")):
    r = Renderer(AdvancedPixelBufferFiller(H, W, fov=fov, device=device), light, None, H, W, on_device=mode)
    ms, n, ts = timed(lambda: r.render(m), dsync, 3.0)
    print(f"{name:20s} {ms:.3f} ms over {n} calls; per-call mean {ts.mean():.3f}, median {np.median(ts):.3f}, slowest " +
          " ".join(f"{i}:{ts[i]:.2f}" for i in np.argsort(-ts)[:5]), "| gc counts", gc.get_count())
