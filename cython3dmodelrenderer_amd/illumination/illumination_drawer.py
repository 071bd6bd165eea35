"""Post-process interface that follows the rasterizer in ``Renderer.render``
(reference: crender/cy/illumination/illumination_drawer.py:5-13)."""


class IlluminationDrawer:
    def draw_illumination(self, color_buffer, n_buffer):
        """Shade ``color_buffer`` in place using the per-pixel normals ``n_buffer``."""
        raise NotImplementedError

    # Optional fast path: shade the filler's device buffers without a host round trip.
    # Return False to make Renderer fall back to draw_illumination on numpy mirrors.
    def draw_illumination_device(self, filler):
        return False


class NoIllumination(IlluminationDrawer):
    def draw_illumination(self, color_buffer, n_buffer):
        pass

    def draw_illumination_device(self, filler):
        return True
