"""``sam3.model_builder`` facade (reference: sam3/sam3/model_builder.py:643-750,944-1053)."""
from efficientsam3_amd.model_builder import build_efficientsam3_image_model, build_sam3_image_model  # noqa: F401
