"""Test-only shim of timm.models."""
from . import layers, vision_transformer, _builder, helpers  # noqa: F401


def register_model(fn):
    return fn
