"""Yardstick for the GEMM kernels: the library GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the shapes the
engine's own `gemm256_kernel` runs, timed with HIP events.  Not part of the product path or of bench.py; the
numbers go to DESIGN.md next to the engine's per-launch figures (bench.py's profile output).

    python tools/gemm_yardstick.py            # on an MI355X
"""
import json

import torch

SHAPES = [  # (name, M, N, K)
    ("neck L0 3x3 as GEMM (B=32: M=32*288^2, K=9*256)", 32 * 288 * 288, 256, 2304),
    ("neck L1 3x3 (M=32*144^2)", 32 * 144 * 144, 256, 2304),
    ("head 3x3 1024->1024 (M=32*32^2, K=9216)", 32 * 32 * 32, 1024, 9216),
    ("ViT-H qkv (B=8: M=8*5184)", 8 * 5184, 3072, 1024),
    ("ViT-H proj", 8 * 5184, 1024, 1024),
    ("ViT-H fc1", 8 * 5184, 4736, 1024),
    ("ViT-H fc2", 8 * 5184, 1024, 4736),
    ("square 8192", 8192, 8192, 8192),
]


def main():
    dev = torch.device("cuda")
    out = []
    for name, m, n, k in SHAPES:
        a = torch.randn((m, k), dtype=torch.bfloat16, device=dev)
        w = torch.randn((n, k), dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            torch.matmul(a, w.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            torch.matmul(a, w.t())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
        out.append({"shape": name, "M": m, "N": n, "K": k, "ms": round(ms, 4), "tflops": round(tf, 1)})
        print(f"{tf:8.1f} TF/s  {ms:8.3f} ms  {name}")
        del a, w
    print(json.dumps(out))


if __name__ == "__main__":
    main()
