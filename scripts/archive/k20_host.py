"""Where a 20-frame burst spends its time: host time per submit, and the wait for the GPU at the
end.  (bench.py --steps 20 --warmup 5, T-Rex 1024^2.)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
tri, col, nrm, (H, W), fov = scenes.scene("trex1024")
f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True)
f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
f.render_frame(); f.synchronize()
for rep in range(4):
    for _ in range(5): f.render_frame()
    f.synchronize(); torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(20):
        f.render_frame(); ts.append(time.perf_counter())
    torch.cuda.synchronize(); te = time.perf_counter()
    d = [1e6 * (b - a) for a, b in zip(ts, ts[1:])]
    print("issue us per frame:", " ".join(f"{x:.1f}" for x in d))
    print(f"  issue total {1e6*(ts[-1]-ts[0]):.1f} us, drain {1e6*(te-ts[-1]):.1f} us, per frame {1e6*(te-ts[0])/20:.2f} us")
