"""Shading step that follows the rasterizer in ``Renderer.render`` — host (numpy) and device
(HIP) forms.  Mirrors the names of the reference's ``crender.cy.illumination`` package."""
from . import guro_illumination as _guro
from . import illumination_drawer as _drawer

GuroIllumination = _guro.GuroIllumination
IlluminationDrawer = _drawer.IlluminationDrawer
NoIllumination = _drawer.NoIllumination

__all__ = ["GuroIllumination", "IlluminationDrawer", "NoIllumination"]
