from .model import Model  # noqa: F401
