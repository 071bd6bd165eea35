from .illumination_drawer import IlluminationDrawer, NoIllumination  # noqa: F401
from .guro_illumination import GuroIllumination  # noqa: F401
