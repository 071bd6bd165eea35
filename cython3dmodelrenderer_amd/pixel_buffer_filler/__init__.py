"""The hot path: ``AdvancedPixelBufferFiller`` backed by the HIP library (the reference's
``crender.cy.pixel_buffer_filler``; its wireframe filler is out of scope)."""
from . import advanced_pixel_buffer_filler as _filler

AdvancedPixelBufferFiller = _filler.AdvancedPixelBufferFiller

__all__ = ["AdvancedPixelBufferFiller"]
