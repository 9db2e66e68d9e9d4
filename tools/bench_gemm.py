"""Development aid: time the implicit-GEMM kernels on the neck / head / ViT-H shapes.

    python tools/bench_gemm.py [substring]          # ESAM3_GEMM256_CLASSIC=1 selects the two-barrier kernel (A/B; needs the
                                                    # dev build: make -C efficientsam3_amd/csrc dev; ESAM3_DEV_LIB=build_dev/libesam3_dev.so)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientsam3_amd import _lib  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, N, ksize, convt
    ("neck L0 3x3 256->256 @288", 32, 288, 288, 256, 256, 3, 0),
    ("neck L1 3x3 256->256 @144", 32, 144, 144, 256, 256, 3, 0),
    ("head.3 3x3 1024->1024 @32", 32, 32, 32, 1024, 1024, 3, 0),
    ("neck L0 up-conv 512->4x256 @144", 32, 144, 144, 512, 1024, 2, 1),
    ("neck L0 convT0 1024->512 @72", 32, 72, 72, 1024, 2048, 1, 1),
    ("neck L0 convT1' 512->256 @144", 32, 144, 144, 512, 1024, 1, 1),
    ("neck L1 convT' 1024->256 @72", 32, 72, 72, 1024, 1024, 1, 1),
    ("neck L2 1x1 1024->256 @72", 32, 72, 72, 1024, 256, 1, 0),
    ("sam2 L0 3x3 256->32 @288", 32, 288, 288, 256, 32, 3, 0),
    ("sam2 L1 3x3 256->64 @144", 32, 144, 144, 256, 64, 3, 0),
    ("ViT-H qkv 1024->3072 (B=8)", 8, 72, 72, 1024, 3072, 1, 0),
    ("ViT-H proj 1024->1024", 8, 72, 72, 1024, 1024, 1, 0),
    ("ViT-H fc1 1024->4736", 8, 72, 72, 1024, 4736, 1, 0),
    ("ViT-H fc2 4736->1024", 8, 72, 72, 4736, 1024, 1, 0),
]
lib = C.CDLL(os.environ.get("ESAM3_DEV_LIB") or _lib.LIB_PATH)  # ESAM3_DEV_LIB: an ablation build (tools/dev_variants.sh)
lib.esam3_bench_gemm.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_float)]
only = sys.argv[1] if len(sys.argv) > 1 else None
print("lib:", os.environ.get("ESAM3_DEV_LIB") or "product", flush=True)
print("kernel:", "gemm256 (classic)" if os.environ.get("ESAM3_GEMM256_CLASSIC", "0") not in ("", "0") else "gemm256p", flush=True)
for name, B, H, W, Cin, N, ks, ct in SHAPES:
    if only and not any(o in name for o in only.split(',')):
        continue
    ms = C.c_float()
    rc = lib.esam3_bench_gemm(1, B, H, W, Cin, N, ks, ct, int(os.environ.get("ESAM3_BENCH_ITERS", "10")), C.byref(ms))
    fl = 2.0 * B * H * W * N * Cin * ks * ks
    print(f"{name:34s} rc={rc} {ms.value:8.3f} ms  {fl / ms.value / 1e9:8.1f} TF/s", flush=True)
    if hasattr(lib, "esam3_dev_read_trace"):  # trace build: cycle stamps of the last launch (workgroups 0-7, waves 0 / 4)
        n = 8 * 2 * 16 * 16
        buf = (C.c_ulonglong * n)()
        lib.esam3_dev_read_trace(buf, n)
        names = ["tile start", "K tile 0 done", "steady loop done", "K loop done", "group barrier", "addr set-up",
                 "block 0 packed (bias arrived)", "vmcnt(0) (next K tile 1 landed)", "stores 0", "stores 1", "stores 2", "stores 3"]
        for blk in (0, 3):
            for wv in (0, 1):
                for tile in (1, 2):
                    base = ((blk * 2 + wv) * 16 + tile) * 16
                    t0 = buf[base]
                    row = [buf[base + i] - t0 for i in range(12)]
                    nxt = buf[base + 16] - t0
                    print(f"  trace wg{blk} wave{wv * 4} tile{tile}: " + ", ".join(f"{nm}={v}" for nm, v in zip(names[1:], row[1:])) + f", next tile start={nxt}")
