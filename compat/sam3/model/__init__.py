"""``sam3.model`` of the import facade; modules not provided here are looked up in the reference checkout named by
ESAM3_REFERENCE_SAM3 (see compat/sam3/__init__.py)."""
import os as _os

_ref = _os.environ.get("ESAM3_REFERENCE_SAM3")
if _ref and _os.path.isdir(_os.path.join(_ref, "model")):
    __path__.append(_os.path.join(_ref, "model"))  # noqa: F821
