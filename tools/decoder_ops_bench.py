"""Development aid: the fused decoder operators at the benchmark shapes (32 prompts, 72 x 72 image tokens, point + box prompt)
through the dev library (ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=N: N timed launches, printed as "[op_timed] ...").

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so ESAM3_OP_REPEAT=20 python tools/decoder_ops_bench.py
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
    Bp, Pn, T = 32, 5184, 10
    x = torch.randn(Bp, Pn, 256, generator=torch.Generator().manual_seed(1)).to("cuda", torch.bfloat16)
    out = torch.empty_like(x)
    args = [rnd(128, 256, seed=2, scale=1 / 16), rnd(128, seed=3, scale=0.1), rnd(Pn, 128, seed=4, scale=0.5), rnd(256, 128, seed=5, scale=128 ** -0.5),
            rnd(256, seed=6, scale=0.1), np.ones(256, np.float32), rnd(256, seed=8, scale=0.1), rnd(Bp, T, 128, seed=9), rnd(Bp, T, 128, seed=10)]
    sys.stderr.write(f"i2t_block Bp={Bp} P={Pn} T={T} ({2 * x.numel() * 2 / 1e6:.0f} MB in + out): ")
    sys.stderr.flush()
    rc = lib.esam3_op_i2t_block(P(x), *[H(a) for a in args], P(out), Bp, Pn, T, None)
    if rc:
        sys.stderr.write(f"rc {rc}\n")
    torch.cuda.synchronize()
    # token -> image attention: the VALU kernel (few_keys 0) and the matrix-core kernel on the merged [k | v] rows (3)
    q = torch.randn(Bp, T, 128, generator=torch.Generator().manual_seed(2)).to("cuda", torch.bfloat16)
    kv = torch.randn(Bp, Pn, 256, generator=torch.Generator().manual_seed(3)).to("cuda", torch.bfloat16)
    k, v = kv[..., :128].contiguous(), kv[..., 128:].contiguous()
    o = torch.empty(Bp, T, 128, dtype=torch.bfloat16, device="cuda")
    sys.stderr.write("attn_t2i VALU: "); sys.stderr.flush()
    lib.esam3_op_attention(1, P(q), P(k), P(v), P(o), Bp, T, Pn, 8, 16, 0, None)
    sys.stderr.write("attn_t2i MFMA: "); sys.stderr.flush()
    lib.esam3_op_attention(1, P(q), P(kv), None, P(o), Bp, T, Pn, 8, 16, 3, None)
    torch.cuda.synchronize()

    # the merged k | v projection: weights-resident row kernel
    xr = torch.randn(Bp * Pn, 256, generator=torch.Generator().manual_seed(5)).to("cuda", torch.bfloat16)
    orow = torch.empty_like(xr)
    sys.stderr.write("rowlin256 (kv projection, 170 MB in + out): "); sys.stderr.flush()
    lib.esam3_op_rowlin256(P(xr), H(rnd(256, 256, seed=6, scale=1 / 16)), H(rnd(256, seed=7, scale=0.1)), H(rnd(Pn, 256, seed=8, scale=0.5)), Pn, P(orow),
                           C.c_int64(Bp * Pn), None)
    torch.cuda.synchronize()
    # hole filling on the low-res logits of 32 prompts (realistic masks: smooth blobs, a few small holes)
    yy, xx = np.mgrid[0:288, 0:288]
    rng = np.random.default_rng(3)
    m = np.full((Bp, 288, 288), -4.0, np.float32)
    for i in range(Bp):
        cy, cx, r = rng.integers(80, 200), rng.integers(80, 200), rng.integers(30, 90)
        m[i][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 4.0
        for _ in range(12):
            hy, hx, hr = rng.integers(0, 288), rng.integers(0, 288), rng.integers(1, 7)
            m[i][(yy - hy) ** 2 + (xx - hx) ** 2 <= hr * hr] *= -1.0
    m_d = torch.from_numpy(m).to("cuda")
    o_d = torch.empty_like(m_d)
    sys.stderr.write("fill_holes 32 x 288^2: "); sys.stderr.flush()
    lib.esam3_op_fill_holes(P(m_d), P(o_d), Bp, 288, 288, C.c_float(0.0), C.c_float(256.0), None)
    torch.cuda.synchronize()
