"""Input side of the rasterizer: the triangle-mesh container whose three ``*_by_triangles``
arrays ``AdvancedPixelBufferFiller.render_model`` reads (the reference's
``crender.cy.data_structures``; its ``Buffer`` wrapper is not needed here: framebuffers are
torch tensors / numpy mirrors)."""
from . import model as _model

Model = _model.Model


def __getattr__(name):            # DeviceModel pulls in torch: imported on first use
    if name == "DeviceModel":
        from .device_model import DeviceModel
        return DeviceModel
    raise AttributeError(name)


__all__ = ["Model", "DeviceModel"]
