"""MI355X-native drop-in for the Version-C rasterizer of oKatanaaa/Cython3DModelRenderer.

One hot path, re-built for gfx950: ``AdvancedPixelBufferFiller.render_model(model)``
(reference: crender/cy/pixel_buffer_filler/advanced_pixel_buffer_filler.pyx:92).
Usage mirrors the reference's ``crender.cy`` package::

    from cython3dmodelrenderer_amd import Renderer
    from cython3dmodelrenderer_amd.data_structures import Model
    from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
    from cython3dmodelrenderer_amd.illumination import GuroIllumination
"""
from .renderer import Renderer  # noqa: F401

__version__ = "0.1.0"
