"""Caller of the hot path — mirror of the reference's ``crender.cy.Renderer``
(reference: crender/cy/renderer.py:9-52): optional model fit, ``render_model``,
illumination, return the colour buffer."""
import numpy as np


class Renderer:
    def __init__(self, pixel_buffer_filler, illumination, triangle_iterator_type=None,
                 image_height=512, image_width=512, use_tqdm=True, on_device=False):
        self.pixel_buffer_filler = pixel_buffer_filler
        self.illumination = illumination
        self.triangle_iterator_type = triangle_iterator_type   # stored, unused (as in Version C)
        self.im_h = image_height
        self.im_w = image_width
        self.use_tqdm = use_tqdm
        # on_device=True keeps shading on the GPU and returns the colour TENSOR; the default
        # reproduces the reference's data flow through writable numpy buffers.
        # on_device="fused": one model per frame — the frame starts from cleared buffers and the
        # raster kernel shades each pixel as it stores it (no illumination pass at all).
        self.on_device = on_device

    def render(self, model, normalize_model=False, random_colors=True):
        if normalize_model:
            # fit the model into the image (reference: renderer.py:41-46)
            centre = (self.im_h // 2, self.im_w // 2)
            span = min(centre)
            model.scale(span / model.get_max_span())
            model.shift(-model.get_mean_vertex() + [centre[0], centre[1], -span])
        filler = self.pixel_buffer_filler
        if self.on_device == "fused" and getattr(self.illumination, "fuse_into", None):
            self.illumination.fuse_into(filler)
            filler.render_model(model, clear=True)
            return filler.get_color_tensor()
        filler.render_model(model)
        if self.on_device and self.illumination.draw_illumination_device(filler):
            return filler.get_color_tensor()
        self.illumination.draw_illumination(filler.get_color_buffer(), filler.get_normals_buffer())
        return filler.get_color_buffer()

    def reset_buffers(self):
        pass   # a no-op in the reference too (renderer.py:51-52): renders composite
