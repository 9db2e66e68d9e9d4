"""Development aid: hole filling (csrc/kernels_decoder.hip) on 32 low-res masks of the kind a decode produces (a few large blobs with
small holes + noise speckle), through the dev library with ESAM3_OP_REPEAT=N ([op_timed] lines); ESAM3_CC_OLD=1 selects the
round-4 grid-wide union-find.     ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 python tools/cc_bench.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientsam3_amd import _lib  # noqa: E402

lib = C.CDLL(os.environ["ESAM3_DEV_LIB"]) if os.environ.get("ESAM3_DEV_LIB") else _lib.load()
rng = np.random.default_rng(3)
n, H, W = 32, 288, 288
yy, xx = np.mgrid[0:H, 0:W]
m = -4.0 * np.ones((n, H, W), np.float32)
for i in range(n):
    for _ in range(3):
        cy, cx, r = rng.integers(40, H - 40), rng.integers(40, W - 40), rng.integers(30, 90)
        m[i][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 5.0
    for _ in range(40):   # small holes / islands
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(1, 6)
        m[i][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] *= -1.0
m += rng.normal(0.0, 0.5, m.shape).astype(np.float32)
m_d = torch.from_numpy(m).to("cuda")
o_d = torch.empty_like(m_d)
sys.stderr.write(f"fill_holes {n} x {H} x {W} ({'old' if os.environ.get('ESAM3_CC_OLD') else 'tiled'}): ")
sys.stderr.flush()
rc = lib.esam3_op_fill_holes(C.c_void_p(m_d.data_ptr()), C.c_void_p(o_d.data_ptr()), n, H, W, C.c_float(0.0), C.c_float(256.0), None)
torch.cuda.synchronize()
if rc:
    sys.stderr.write(f"rc {rc}\n")
from scipy import ndimage  # noqa: E402
ref = m.copy()
st = np.ones((3, 3), dtype=np.int32)
for i in range(n):
    lab, k = ndimage.label(m[i] <= 0.0, structure=st)
    if k:
        a = np.bincount(lab.ravel(), minlength=k + 1)
        ref[i][(lab > 0) & (a[lab] <= 256)] = 10.0
print("fill_holes bit-exact vs scipy:", bool(np.array_equal(o_d.cpu().numpy(), ref)))
