"""``sam3.model.box_ops`` of the import facade (efficientsam3_image_predictor_example.py:29)."""
from efficientsam3_amd.box_ops import (box_area, box_cxcywh_to_xywh, box_cxcywh_to_xyxy, box_iou,  # noqa: F401
                                       box_xywh_to_cxcywh, box_xywh_to_xyxy, box_xyxy_to_cxcywh, box_xyxy_to_xywh)
