"""Development aid: the exact-lattice MBConv cases of tests/test_lattice_gpu.py through the DEV library (ESAM3_DEV_LIB), so that kernel
variants selected by dev-build environment switches (ESAM3_MB3B_SMALL=1, ESAM3_MB3B_64=1, ESAM3_MB3_GENERIC=1 ...) can be checked bit for
bit before they become the default path.

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_MB3B_SMALL=1 python tools/mbconv_variant_check.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_lattice_gpu as L  # noqa: E402
import util as U  # noqa: E402

lib = C.CDLL(os.environ["ESAM3_DEV_LIB"])
CASES = [(2, 40, 40, 16, 64, 32, 2, 0), (1, 30, 26, 32, 128, 32, 1, 1), (1, 41, 23, 32, 128, 64, 2, 0), (2, 33, 50, 64, 256, 64, 1, 1),
         (3, 64, 64, 16, 64, 32, 2, 0), (4, 126, 126, 64, 256, 64, 1, 1), (5, 100, 132, 32, 128, 32, 1, 1), (5, 101, 131, 32, 128, 64, 2, 0),
         (3, 250, 250, 16, 64, 32, 2, 0), (2, 21, 19, 64, 256, 128, 2, 0), (1, 31, 29, 128, 512, 256, 2, 0)]
bad_total = 0
for (B, H, W, Cin, Cmid, Cout, stride, res) in CASES:
    x, w1, b1, wd, bd, w2, b2, ref = L.mbconv_lattice(B, H, W, Cin, Cmid, Cout, stride, res)
    x_d = U.to_dev_nhwc(x, torch.bfloat16)
    OH, OW = ref.shape[-2:]
    out = torch.full((B, OH, OW, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = lib.esam3_op_mbconv3(U.P(x_d), U.H(U.np32(w1[:, :, None, None])), U.H(U.np32(b1)), U.H(U.np32(wd)), U.H(U.np32(bd)),
                              U.H(U.np32(w2[:, :, None, None])), U.H(U.np32(b2)), U.P(out), B, H, W, Cin, Cmid, Cout, stride, res, None)
    got = U.from_dev_nhwc(out)
    bad = int((got != ref.float()).sum()) if rc == 0 else -1
    bad_total += bad != 0
    print(f"{Cin}->{Cmid}->{Cout} s{stride} B{B} {H}x{W}: rc {rc}, {bad} of {got.numel()} elements differ")
print("ALL EXACT" if bad_total == 0 else f"{bad_total} CASES DIFFER")
