"""Helpers shared by the GPU parity tests: call single operators through the C ABI."""
import ctypes as C

import numpy as np
import torch

from efficientsam3_amd import _lib

ACT = {None: 0, "relu": 1, "gelu": 2, "hswish": 3, "sigmoid": 4}
DT = {"f32": (0, torch.float32), "bf16": (1, torch.bfloat16)}
# tolerance of one operator vs its fp32 torch reference: (atol, rtol)
TOL = {"f32": (2e-4, 2e-4), "bf16": (6e-2, 3e-2)}


def lib():
    return _lib.load()


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def H(a):
    """host fp32 numpy -> void*"""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def np32(t):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().float().numpy())


def to_dev_nhwc(x_nchw: torch.Tensor, tdt) -> torch.Tensor:
    return x_nchw.permute(0, 2, 3, 1).contiguous().to("cuda", tdt)


def from_dev_nhwc(y_nhwc: torch.Tensor) -> torch.Tensor:
    return y_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def check(rc, what):
    _lib.check(rc, what)


def assert_close(got: torch.Tensor, ref: torch.Tensor, mode: str, what: str = "", scale=1.0):
    atol, rtol = TOL[mode]
    atol, rtol = atol * scale, rtol * scale
    diff = (got.double() - ref.double()).abs()
    bound = atol + rtol * ref.double().abs()
    bad = diff > bound
    assert not bad.any(), (f"{what} [{mode}]: {int(bad.sum())}/{bad.numel()} elements out of tolerance; "
                           f"max abs err {float(diff.max()):.3e}, ref max {float(ref.abs().max()):.3e}")


# ---- golden prompt cases (tests/golden/manifest.json, written by oracle/gen_golden.py) ----------
def case_kwargs(case):
    """manifest 'kw' -> predict_inst keyword arguments ("mask_input" is a synth.mask_logits seed)."""
    from efficientsam3_amd import synth
    kw = {}
    for k, v in case["kw"].items():
        if k == "mask_input":
            kw[k] = synth.mask_logits(seed=v)
        elif isinstance(v, list):
            kw[k] = np.asarray(v, dtype=np.int32 if k == "point_labels" else np.float32)
        else:
            kw[k] = v
    return kw


def case_image_chw_u8(case):
    """None for cases on the shared 1008x1008 image 0; else the case's own CHW uint8 image."""
    from efficientsam3_amd import synth
    spec = case.get("image")
    if spec is None:
        return None
    assert spec["kind"] == "smooth_crop"
    h, w = case["hw"]
    img = np.ascontiguousarray(synth.smooth_image_u8(seed=spec["seed"], size=max(h, w))[:h, :w])
    return torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0)))
