"""EfficientSAM3 image hot path, MI355X-native (HIP kernels behind a C ABI).

Public surface mirrors the reference (sam3/sam3/__init__.py:3-21):
``build_efficientsam3_image_model``, ``Sam3Processor``; see INTEGRATION.md.
Importing this package does not load the HIP library; building a model does, and fails
loudly if it is missing (there is no CPU fallback).
"""
from .model_builder import build_efficientsam3_image_model, build_sam3_image_model  # noqa: F401
from .sam3_image_processor import Sam3Processor  # noqa: F401

__all__ = ["build_efficientsam3_image_model", "build_sam3_image_model", "Sam3Processor"]
__version__ = "0.1.0"
