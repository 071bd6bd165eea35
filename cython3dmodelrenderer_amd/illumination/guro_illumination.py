"""Gouraud-style ("Guro") directional shading, SURVEY.md section 8f row f1
(reference: crender/cy/illumination/guro_illumination.py:6-27):

    colour *= clip(n . l / (|n| + 1e-6), 0, 1)      per pixel, float32, in place

``draw_illumination`` is the numpy form on host arrays (same call sequence as the
reference, so the same bits); ``draw_illumination_device`` runs the HIP kernel
``crender_guro_illumination`` on the filler's device buffers; ``fuse_into`` hands the light
to the filler, whose cleared frames then shade each pixel as the raster kernel stores it
(CRENDER_FUSED_GURO: no second pass over the colour and normal planes).
"""
import ctypes as C

import numpy as np

from .illumination_drawer import IlluminationDrawer


class GuroIllumination(IlluminationDrawer):
    def __init__(self, light_direction=(0, 0, 1)):
        # the light vector is flipped so that it opposes front-facing normals
        flipped = -np.asarray(light_direction, dtype="float32")
        self.light_direction = flipped / np.linalg.norm(flipped)

    def draw_illumination(self, color_buffer, n_buffer):
        cos = np.sum(n_buffer * self.light_direction, axis=-1, keepdims=True)
        length = np.linalg.norm(n_buffer, axis=-1, keepdims=True)
        color_buffer *= np.clip(cos / (length + 1e-6), 0, 1)

    def draw_illumination_device(self, filler):
        from .. import _capi
        lib = _capi.load()
        light = (C.c_float * 3)(*[float(v) for v in self.light_direction])
        filler._push_host_edits()
        filler.synchronize()      # the frame must be complete (bin lists may grow and redo it)
        _capi.check(lib.crender_guro_illumination(
            filler.color_buffer.data_ptr(), filler.normals_buffer.data_ptr(), light,
            filler.h, filler.w, filler.y0, filler.y1, filler._stream()), "crender_guro_illumination")
        filler._host_fresh = False
        return True

    def fuse_into(self, filler):
        filler.set_fused_illumination(self.light_direction)
        return True
