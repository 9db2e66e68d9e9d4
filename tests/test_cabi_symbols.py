"""CPU: the C-ABI library is built, loads with ctypes, and exports every function that
include/esam3.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

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


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/esam3.h must compile as C99 (no C++ in the signatures) and the library must
    link from a C translation unit."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    src = tmp_path / "t.c"
    src.write_text('#include "esam3.h"\n'
                   "int probe(void) { esam3_config c; esam3_ground_in g; esam3_ground_out o; (void)c; (void)g; (void)o;\n"
                   "  return esam3_last_error() == 0 && esam3_rle_scratch_bytes(1, 8, 8, 16) > 0; }\n")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o",
                        str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
