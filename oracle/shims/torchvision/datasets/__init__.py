from . import vision  # noqa: F401
