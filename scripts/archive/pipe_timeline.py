import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
tri, col, nrm, (H, W), fov = scenes.scene("trex1024")
f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True)
f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
for _ in range(40): f.render_frame()
torch.cuda.synchronize()
