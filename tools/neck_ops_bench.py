"""Development aid: the neck-side elementwise operators at the real EV-M shapes, B = 32, through the dev library
(ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=N: N timed launches per op call, printed as "[op_timed] ...").

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 python tools/neck_ops_bench.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientsam3_amd import _lib  # noqa: E402

lib = C.CDLL(os.environ["ESAM3_DEV_LIB"]) if os.environ.get("ESAM3_DEV_LIB") else _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
H = lambda a: a.ctypes.data_as(C.c_void_p)
ACT_NONE, ACT_GELU = 0, 2
# name: C, taps, act, pad     (student head output 32 x 32 -> 72 x 72, the first layer of each neck level computed on the small map)
RS = {"level0 convT 1024->512 +GELU": (512, 4, ACT_GELU, 1), "level0 no act": (512, 4, ACT_NONE, 1),
      "level1 convT 1024->256": (256, 4, ACT_NONE, 1), "level2 1x1 ->256": (256, 1, ACT_NONE, 1)}

if __name__ == "__main__":
    B, IH, OH = 32, 32, 72
    for name, (Cc, taps, act, pad) in RS.items():
        x = torch.randn(B, IH, IH, taps * Cc, generator=torch.Generator().manual_seed(1)).to("cuda", torch.bfloat16)
        s = 2 if taps == 4 else 1
        out = torch.zeros((B, s * OH + 2 * pad, s * OH + 2 * pad, Cc), dtype=torch.bfloat16, device="cuda")
        bias = (np.arange(Cc) % 7 * 0.01).astype(np.float32)
        gb = out.numel() * 2 / 1e9
        sys.stderr.write(f"resize_shuffle {name} ({gb:.3f} GB written): ")
        sys.stderr.flush()
        rc = lib.esam3_op_resize_shuffle(1, P(x), H(bias), P(out), B, IH, IH, OH, OH, Cc, taps, act, pad, None)
        if rc:
            sys.stderr.write(f"rc {rc}\n")
    torch.cuda.synchronize()
