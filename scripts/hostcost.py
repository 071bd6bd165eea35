"""Where the host time of one pipelined frame goes (T-Rex 1024^2): the Python method, the swap
chain's frame(), the bare two-argument ctypes call.  scripts/ubench/frame_issue.cpp is the same
loop without Python (5.1 us per frame: two kernel launches)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
tri, col, nrm, (H, W), fov = scenes.scene("trex1024")
f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True)
f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
for _ in range(50): f.render_frame()
torch.cuda.synchronize()
K = 3000
def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:44s} issue {1e6*(t1-t0)/K:6.2f} us/frame   total {1e6*(t2-t0)/K:6.2f} us/frame")
timed("filler.render_frame()", f.render_frame)
pipe = f._pipe
timed("pipe.frame(filler)", lambda: pipe.frame(f))
lib = pipe.lib
stream = torch.cuda.current_stream().cuda_stream
timed("ctypes crender_pipeline_submit(handle, stream)", lambda: lib.crender_pipeline_submit(pipe.handle, stream))
ext = f._ext
h, idx = pipe.handle.value, pipe._index
timed("crender_torch.pipeline_submit(handle, device)", lambda: ext.pipeline_submit(h, idx))
