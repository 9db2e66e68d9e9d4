"""Test-only shim of the torchvision surface the reference imports (see ../README.md)."""
__version__ = "0.25.0+shim"
from . import ops, transforms, datasets  # noqa: F401
