from .advanced_pixel_buffer_filler import AdvancedPixelBufferFiller  # noqa: F401
