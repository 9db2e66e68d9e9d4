"""Development aid: the RepViT / TinyViT patch-embedding stem (esam3_op_stem, bf16) through the development library: the persistent kernel of
round 6 against the one-tile-per-workgroup kernel it replaces (ESAM3_STEM_OLD=1), bit for bit, with rough timings (the op uploads its weights
and synchronises per call: compare the two columns, not the absolute values).

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so python tools/stem_ab.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.environ["ESAM3_DEV_LIB"])
P = lambda t: C.c_void_p(t.data_ptr())
H = lambda a: a.ctypes.data_as(C.c_void_p)


def run(x, w, b, cout, old, reps=10):
    os.environ["ESAM3_STEM_OLD"] = "1" if old else "0"
    B, _, S, _ = x.shape
    out = torch.full((B, (S + 1) // 2, (S + 1) // 2, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    assert lib.esam3_op_stem(1, P(x), H(w), H(b), P(out), B, S, S, cout, 2, None) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.esam3_op_stem(1, P(x), H(w), H(b), P(out), B, S, S, cout, 2, None)
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3


for (B, S, cout) in [(32, 1008, 32), (32, 1008, 24), (5, 1001, 16), (3, 640, 48), (2, 1008, 64)]:
    if cout % 16:
        continue
    g = torch.Generator().manual_seed(cout + S)
    x = torch.randn(B, 3, S, S, generator=g).cuda()
    w = (torch.randn(cout, 3, 3, 3, generator=g) / 27 ** 0.5).numpy().astype(np.float32)
    b = (torch.randn(cout, generator=g) * 0.1).numpy().astype(np.float32)
    o_old, t_old = run(x, w, b, cout, True)
    o_new, t_new = run(x, w, b, cout, False)
    same = bool(torch.equal(o_old.view(torch.int16), o_new.view(torch.int16)))
    print(f"B {B} {S}^2 -> {cout} ch: {'bit-identical' if same else 'DIFFERS'}; one tile per workgroup {t_old:.3f} ms, persistent {t_new:.3f} ms per op call")
