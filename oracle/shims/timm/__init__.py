"""Test-only shim of the parts of timm the reference imports (see ../README.md)."""
__version__ = "1.0.17"
from . import layers, models  # noqa: F401
