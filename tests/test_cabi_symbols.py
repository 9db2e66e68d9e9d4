"""CPU: the C-ABI library is built, loads with ctypes, and exports every function that
include/esam3.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from efficientsam3_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "esam3.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(esam3_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.load()


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    lib = _lib.load()
    cfg = _lib.Config(dtype=1, backbone=0, model_name=b"b1", device=0, interactive=1)
    h = ctypes.c_void_p()
    assert lib.esam3_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"HIP device" in lib.esam3_last_error()
    import pytest
    from efficientsam3_amd import build_efficientsam3_image_model
    with pytest.raises(Exception):
        build_efficientsam3_image_model(enable_inst_interactivity=True, model_name="b1")
