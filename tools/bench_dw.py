"""Development aid: time the depthwise-conv launches of the EV-M / RV-M / TV-M backbones (B = 32).

    ESAM3_DEV_LIB=build_dev/libesam3_x.so python tools/bench_dw.py [act_flags]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientsam3_amd import _lib  # noqa: E402

SHAPES = [  # name, B, H, W, C, k, stride
    ("EV-M stage3 MBConv dw3 512@63", 32, 63, 63, 512, 3, 1),
    ("EV-M stage4 MBConv dw3 1024@32", 32, 32, 32, 1024, 3, 1),
    ("EV-M stage3 LiteMLA dw5 384@63", 32, 63, 63, 384, 5, 1),
    ("EV-M stage4 LiteMLA dw5 768@32", 32, 32, 32, 768, 5, 1),
    ("TV-M layers.0 dw3 256@252", 32, 252, 252, 256, 3, 1),
]
lib = C.CDLL(os.environ.get("ESAM3_DEV_LIB") or _lib.LIB_PATH)
lib.esam3_bench_dwconv.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_float)]
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
print("lib:", os.environ.get("ESAM3_DEV_LIB") or "product", "dev flags", flags, flush=True)
for name, B, H, W, Cc, k, st in SHAPES:
    ms = C.c_float()
    rc = lib.esam3_bench_dwconv(1, B, H, W, Cc, k, st, 3 | (flags << 8), 20, C.byref(ms))
    mb = 2 * B * H * W * Cc * 2 / 2 ** 20
    print(f"{name:34s} rc={rc} {ms.value * 1e3:8.1f} us  {mb:7.1f} MB in+out  {mb * 2 ** 20 / ms.value / 1e9:6.2f} TB/s", flush=True)
