"""Import facade: put ``<repo>/compat`` on PYTHONPATH and the reference's callers run unchanged on
the MI355X engine --

    from sam3 import build_efficientsam3_image_model            (eval/eval_coco.py:16)
    from sam3.model.sam3_image_processor import Sam3Processor    (eval/eval_coco.py:17)
    from sam3.device import get_device                           (eval/eval_coco.py:18)
    from sam3.model_builder import build_sam3_image_model, build_efficientsam3_image_model
    from sam3.model.box_ops import box_xywh_to_cxcywh            (efficientsam3_image_predictor_example.py:29)
    from sam3.model.tokenizer_ve import SimpleTokenizer          (stage1/model.py, the tokenizer the text path uses)
    import sam3.sam3.<...>                                       (stage1/model.py:8-27 uses BOTH spellings)

(SURVEY.md §8(b) "Import surface callers rely on").  The hot path is a re-export of ``efficientsam3_amd``.

Everything the hot path does NOT replace (``sam3.visualization_utils``, ``sam3.backbones.*``, ``sam3.train`` ...) is
forwarded to a reference checkout when one is named: set ``ESAM3_REFERENCE_SAM3`` to the reference's INNER package
directory (``<reference>/sam3/sam3``) and those imports resolve there, while the names above keep resolving here
(this directory comes first on the package's search path).  Without it they raise ImportError as usual.
"""
import os as _os
import sys as _sys

from efficientsam3_amd import build_efficientsam3_image_model, build_sam3_image_model  # noqa: F401

_ref = _os.environ.get("ESAM3_REFERENCE_SAM3")
if _ref and _os.path.isdir(_ref):
    __path__.append(_ref)  # noqa: F821  (a package's own search path)

# the reference's outer shim makes ``sam3.sam3`` the inner package (sam3/__init__.py:8-20); both spellings work here too
_sys.modules.setdefault(__name__ + ".sam3", _sys.modules[__name__])

__all__ = ["build_efficientsam3_image_model", "build_sam3_image_model"]
