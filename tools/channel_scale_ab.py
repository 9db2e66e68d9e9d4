"""Development aid: esam3_channel_scale's per-channel fast kernel (round 6) against the general kernel (reached with per-image multipliers that
repeat the same row), bit for bit, fp32 and bf16, with and without the add / bias terms."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientsam3_amd import _lib  # noqa: E402

lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
ok = True
for dt, tdt in ((0, torch.float32), (1, torch.bfloat16)):
    for (B, HW, Cc) in ((4, 63 * 63, 128), (2, 126 * 126, 64), (3, 1000, 256), (2, 504 * 504, 16)):
        g = torch.Generator().manual_seed(B + Cc)
        x = torch.randn(B, HW, Cc, generator=g).to("cuda", tdt)
        a = torch.randn(B, HW, Cc, generator=g).to("cuda", tdt)
        mul = torch.randn(Cc, generator=g).cuda()
        bias = torch.randn(Cc, generator=g).cuda()
        for use_add in (False, True):
            for use_bias in (False, True):
                o1 = torch.empty_like(x)
                o2 = torch.empty_like(x)
                rc1 = lib.esam3_channel_scale(dt, P(x), P(mul), 0, C.c_float(0.25), P(bias) if use_bias else None, 0, C.c_float(0.5),
                                              P(a) if use_add else None, P(o1), B, HW, Cc, None)
                mulb = mul[None].expand(B, Cc).contiguous()
                rc2 = lib.esam3_channel_scale(dt, P(x), P(mulb), 1, C.c_float(0.25), P(bias) if use_bias else None, 0, C.c_float(0.5),
                                              P(a) if use_add else None, P(o2), B, HW, Cc, None)
                torch.cuda.synchronize()
                same = rc1 == 0 and rc2 == 0 and bool(torch.equal(o1.view(torch.uint8), o2.view(torch.uint8)))
                ok &= same
                if not same:
                    print("DIFFERS", dt, B, HW, Cc, use_add, use_bias)
print("channel_scale fast kernel: " + ("bit-identical to the general kernel on all cases" if ok else "MISMATCH"))
