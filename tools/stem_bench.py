"""Development aid: the fused EfficientViT input stem at the headline shape (B = 32, 1008^2) through esam3_op_stem_dsconv;
ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=N prints the average of N launches per variant ("[op_timed] stem_dsconv").

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 python tools/stem_bench.py [variant ...]
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


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).numpy().astype(np.float32)


if __name__ == "__main__":
    B, S = 32, 1008
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(1)).to("cuda")
    w0, b0 = rnd(16, 3, 3, 3, seed=2, scale=27 ** -0.5), rnd(16, seed=3, scale=0.1)
    wd, bd = rnd(16, 1, 3, 3, seed=4, scale=1 / 3), rnd(16, seed=5, scale=0.1)
    wp, bp = rnd(16, 16, seed=6, scale=0.25), rnd(16, seed=7, scale=0.1)
    out = torch.empty((B, S // 2, S // 2, 16), dtype=torch.bfloat16, device="cuda")
    for v in [int(a) for a in sys.argv[1:]] or [0, 1]:
        sys.stderr.write(f"variant {v}: ")
        sys.stderr.flush()
        rc = lib.esam3_op_stem_dsconv(1, P(x), H(w0), H(b0), H(wd), H(bd), H(wp), H(bp), P(out), B, S, S, v, None)
        if rc:
            sys.stderr.write(f"rc {rc}\n")
    torch.cuda.synchronize()
