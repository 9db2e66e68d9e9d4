"""``sam3.model.sam3_image_processor`` facade (reference: sam3/sam3/model/sam3_image_processor.py)."""
from efficientsam3_amd.sam3_image_processor import Sam3Processor  # noqa: F401
