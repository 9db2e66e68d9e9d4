"""Import facade: put ``<repo>/compat`` on PYTHONPATH and the reference's callers run unchanged on
the MI355X engine --

    from sam3 import build_efficientsam3_image_model            (eval/eval_coco.py:16)
    from sam3.model.sam3_image_processor import Sam3Processor    (eval/eval_coco.py:17)
    from sam3.device import get_device                           (eval/eval_coco.py:18)
    from sam3.model_builder import build_sam3_image_model, build_efficientsam3_image_model

(SURVEY.md §8(b) "Import surface callers rely on").  Everything is a re-export of
``efficientsam3_amd``; modules of the reference that are outside the hot path are not provided.
"""
from efficientsam3_amd import build_efficientsam3_image_model, build_sam3_image_model  # noqa: F401

__all__ = ["build_efficientsam3_image_model", "build_sam3_image_model"]
