"""Development aid: the fused EfficientViT operators (csrc/evit_fused.hip) at the real EV-M shapes, B = 32, through the dev
library (ESAM3_DEV_LIB=build_dev/libesam3_dev.so, built with -DESAM3_DEV): ESAM3_OP_REPEAT=N makes every op call time N
launches with HIP events (printed by the library as "[op_timed] ..."), ESAM3_MB3_ABL=mask drops phases of mbconv3
(1: no x loads, 2: no expand phase, 4: no depthwise phase, 8: no project phase, 16: no output stores).

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 python tools/evit_fused_bench.py [which ...]
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
MB = {  # name: B, H, W, Cin, Cmid, Cout, stride, res
    "s0.0": (32, 504, 504, 16, 64, 32, 2, 0), "s0.1": (32, 252, 252, 32, 128, 32, 1, 1),
    "s1.0": (32, 252, 252, 32, 128, 64, 2, 0), "s1.1": (32, 126, 126, 64, 256, 64, 1, 1),
    "s2.0": (32, 126, 126, 64, 256, 128, 2, 0), "s2.loc": (32, 63, 63, 128, 512, 128, 1, 1),
    "s3.0": (32, 63, 63, 128, 512, 256, 2, 0), "s3.loc": (32, 32, 32, 256, 1024, 256, 1, 1),
    # TinyViT-5M / 11M layer 0 (res = 3: shortcut + the GELU variant, tiny_vit.py:73-108); "tv.hs" = the same shape with Hardswish
    "tv.0": (32, 252, 252, 64, 256, 64, 1, 3), "tv.hs": (32, 252, 252, 64, 256, 64, 1, 1),
}
MLA = {"s2.ctx": (32, 63, 63, 128), "s3.ctx": (32, 32, 32, 256)}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).numpy().astype(np.float32)


def run_mb(name, v2=False):
    B, Hh, W, Cin, Cmid, Cout, st, res = MB[name]
    x = torch.randn(B, Hh, W, Cin, generator=torch.Generator().manual_seed(1)).to("cuda", torch.bfloat16)
    w1, b1 = rnd(Cmid, Cin, seed=2, scale=(2.0 / Cin) ** 0.5), rnd(Cmid, seed=3, scale=0.1)
    wd, bd = rnd(Cmid, 9, seed=4, scale=0.4), rnd(Cmid, seed=5, scale=0.1)
    w2, b2 = rnd(Cout, Cmid, seed=6, scale=Cmid ** -0.5), rnd(Cout, seed=7, scale=0.1)
    out = torch.empty((B, (Hh + st - 1) // st, (W + st - 1) // st, Cout), dtype=torch.bfloat16, device="cuda")
    sys.stderr.write(f"{name} {'v2 ' if v2 else ''}{Cin}->{Cmid}->{Cout} s{st} @{Hh}: ")
    sys.stderr.flush()
    if v2:
        rc = lib.esam3_op_mbconv_fused(1, P(x), H(w1), H(b1), H(wd), H(bd), H(w2), H(b2), P(out), B, Hh, W, Cin, Cmid, Cout, st, res, None)
    else:
        rc = lib.esam3_op_mbconv3(P(x), H(w1), H(b1), H(wd), H(bd), H(w2), H(b2), P(out), B, Hh, W, Cin, Cmid, Cout, st, res, None)
    if rc:
        sys.stderr.write(f"rc {rc}\n")


def run_mla(name):
    B, Hh, W, Cc = MLA[name]
    x = torch.randn(B, Hh, W, Cc, generator=torch.Generator().manual_seed(1)).to("cuda", torch.bfloat16)
    wq, wd5 = rnd(3 * Cc, Cc, seed=2, scale=Cc ** -0.5), rnd(3 * Cc, 25, seed=3, scale=0.2)
    wg, wp, bp = rnd(3 * Cc, 16, seed=4, scale=0.25), rnd(Cc, 2 * Cc, seed=5, scale=(2 * Cc) ** -0.5), rnd(Cc, seed=6, scale=0.1)
    out = torch.empty((B, Hh, W, Cc), dtype=torch.bfloat16, device="cuda")
    sys.stderr.write(f"{name} lite_mla_block C={Cc} @{Hh}: ")
    sys.stderr.flush()
    rc = lib.esam3_op_lite_mla_block(P(x), H(wq), H(wd5), H(wg), H(wp), H(bp), P(out), B, Hh, W, Cc, None)
    if rc:
        sys.stderr.write(f"rc {rc}\n")


if __name__ == "__main__":
    which = sys.argv[1:] or list(MB) + list(MLA)
    for w in which:
        if w.endswith(":v2"):
            run_mb(w[:-3], v2=True)
        elif w in MB:
            run_mb(w)
        else:
            run_mla(w)
    torch.cuda.synchronize()
