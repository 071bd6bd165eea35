"""Caller of the hot path — mirror of the reference's ``crender.cy.Renderer``
(reference: crender/cy/renderer.py:9-52): optional model fit, ``render_model``,
illumination, return the colour buffer."""
import numpy as np


def _device_form_is_the_same_shading(illumination):
    """True if the class that defines ``draw_illumination_device`` is the one that defines
    ``draw_illumination`` (or derives from it): the device form then belongs to the host form in
    force.  A subclass overriding only ``draw_illumination`` makes the inherited device form stale."""
    host_at = dev_at = None
    for i, cls in enumerate(type(illumination).__mro__):
        if host_at is None and "draw_illumination" in vars(cls):
            host_at = i
        if dev_at is None and "draw_illumination_device" in vars(cls):
            dev_at = i
    return dev_at is not None and host_at is not None and dev_at <= host_at


class Renderer:
    def __init__(self, pixel_buffer_filler, illumination, triangle_iterator_type=None,
                 image_height=512, image_width=512, use_tqdm=True, on_device=None):
        self.pixel_buffer_filler = pixel_buffer_filler
        self.illumination = illumination
        self.triangle_iterator_type = triangle_iterator_type   # stored, unused (as in Version C)
        self.im_h = image_height
        self.im_w = image_width
        self.use_tqdm = use_tqdm
        # on_device=None (default): same call sequence and same return value as the reference — the
        #   filler's writable numpy colour buffer — but when the filler keeps its buffers on the GPU
        #   and the illumination has a device form, the shading runs there, so that only the colour
        #   plane crosses PCIe (the normals are never handed out);
        # on_device=False: the reference's data flow to the letter, numpy illumination on the
        #   filler's host views of colour and normals;
        # on_device=True: shading on the GPU, returns the colour TENSOR (nothing crosses PCIe);
        # on_device="fused": one model per frame — the frame starts from cleared buffers and the
        #   raster kernel shades each pixel as it stores it (no illumination pass at all).
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
        device_form = getattr(self.illumination, "draw_illumination_device", None)
        if self.on_device is None and not _device_form_is_the_same_shading(self.illumination):
            # a subclass that overrides draw_illumination alone — the reference's only hook
            # (renderer.py:47-49) — gets ITS shading, on the host views, not the parent's device form
            device_form = None
        if self.on_device is not False and device_form is not None and hasattr(filler, "get_color_tensor"):
            # (the views handed out so far are refreshed by the getter below, after the shading)
            filler.render_model(model, refresh_views=False)
            if device_form(filler):
                return filler.get_color_tensor() if self.on_device is True else filler.get_color_buffer()
        else:
            filler.render_model(model)
        self.illumination.draw_illumination(filler.get_color_buffer(), filler.get_normals_buffer())
        return filler.get_color_buffer()

    def reset_buffers(self):
        pass   # a no-op in the reference too (renderer.py:51-52): renders composite
