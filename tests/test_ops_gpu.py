"""GPU parity of every HIP operator (through the C ABI) against plain PyTorch fp32 on CPU.

Each kernel of the hot path is exercised in both activation modes: "f32" (validation,
exact-f32 MFMA; tight tolerance) and "bf16" (throughput; tolerance = bf16 input/output
rounding).  Shapes include ragged tails (M, N, K not multiples of the tile) and the edge
cases the graphs rely on (K < one K-tile, N in {1, 4}, stride-2 odd sizes, batch gather).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util as U

pytestmark = pytest.mark.gpu
MODES = ["f32", "bf16"]


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _q(x, mode):
    """quantise an input the way the device sees it"""
    return x.to(torch.bfloat16).float() if mode == "bf16" else x


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("M,N,K,act,res", [
    (8, 256, 256, None, False), (300, 128, 256, "relu", True), (1000, 2048, 256, "relu", False),
    (77, 256, 2048, None, True), (5, 4, 256, "sigmoid", False), (3, 1, 256, None, False),
    (513, 32, 16, "hswish", False), (129, 64, 24, "gelu", True), (64, 384, 128, None, False),
    (288, 256, 256, "relu", True), (2000, 64, 512, "gelu", False),  # skinny_gemm_kernel: token rows x (256 | 512) K
])
def test_linear(mode, M, N, K, act, res):
    d, tdt = U.DT[mode]
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2) / K ** 0.5, _rand(N, seed=3) * 0.1
    r = _rand(M, N, seed=4) if res else None
    aq, wq = _q(a, mode), _q(w, mode)
    ref = F.linear(aq, wq, b)
    ref = {None: lambda t: t, "relu": F.relu, "gelu": F.gelu, "hswish": F.hardswish,
           "sigmoid": torch.sigmoid}[act](ref)
    if res:
        ref = ref + _q(r, mode)
    a_d = a.to("cuda", tdt)
    r_d = r.to("cuda", tdt) if res else None
    out = torch.empty((M, N), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_linear(d, U.P(a_d), U.H(U.np32(w)), U.H(U.np32(b)), U.P(r_d), U.P(out),
                                    M, N, K, U.ACT[act], None), "op_linear")
    U.assert_close(out.float().cpu(), ref, mode, f"linear {M}x{N}x{K}")


def test_gelu_epilogue_is_erf_gelu_to_1e6():
    """The GELU every epilogue uses (esam3_common.h gelu_fast: max(x, 0) + u P(u) exp(-x^2 / 2), one transcendental) against
    torch's erf GELU on a dense sweep of [-12, 12] plus tiny, huge and signed-zero inputs, through an fp32 identity Linear (the
    product by the identity is exact): |error| <= 1e-6 absolute and, for |x| < 1, <= 1e-5 relative."""
    K = 64
    x = torch.cat([torch.linspace(-12, 12, 64 * 2048 - 64 * 2), torch.logspace(-30, 0, 64), -torch.logspace(-30, 0, 64)]).float()
    x[:8] = torch.tensor([0.0, -0.0, 1e4, -1e4, 3e38, -3e38, 5.9, -5.9])
    a = x.view(-1, K).contiguous()
    M = a.shape[0]
    w = np.ascontiguousarray(np.eye(K, dtype=np.float32))
    b = np.zeros(K, dtype=np.float32)
    out = torch.empty((M, K), dtype=torch.float32, device="cuda")
    U.check(U.lib().esam3_op_linear(0, U.P(a.to("cuda")), U.H(w), U.H(b), None, U.P(out), M, K, K, U.ACT["gelu"], None), "op_linear")
    got = out.cpu().double().view(-1)
    ref = F.gelu(x.double())
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    assert float(err[x.abs() < 100].max()) <= 1e-6, float(err[x.abs() < 100].max())
    big = x.abs() >= 100
    assert torch.equal(got[big], torch.where(x[big] > 0, x[big], torch.zeros_like(x[big])).double())   # exactly relu far out
    small = (x.abs() < 1) & (x != 0)
    assert float((err[small] / ref[small].abs()).max()) <= 1e-5


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,act,res", [
    (2, 20, 24, 256, 256, 3, None, False), (1, 9, 7, 64, 32, 3, "gelu", False),
    (1, 33, 31, 16, 64, 1, "hswish", False), (2, 12, 12, 1024, 256, 1, None, True),
    (1, 16, 16, 1024, 1024, 3, None, False), (1, 5, 5, 8, 8, 3, None, True),
])
def test_conv2d(mode, B, H, W, Cin, Cout, ks, act, res):
    d, tdt = U.DT[mode]
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, ks, ks, seed=2) / (Cin * ks * ks) ** 0.5
    b = _rand(Cout, seed=3) * 0.1
    r = _rand(B, Cout, H, W, seed=4) if res else None
    ref = F.conv2d(_q(x, mode), _q(w, mode), b, padding=ks // 2)
    ref = {None: lambda t: t, "gelu": F.gelu, "hswish": F.hardswish}[act](ref)
    if res:
        ref = ref + _q(r, mode)
    x_d = U.to_dev_nhwc(x, tdt)
    r_d = U.to_dev_nhwc(r, tdt) if res else None
    out = torch.empty((B, H, W, Cout), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_conv2d(d, U.P(x_d), U.H(U.np32(w)), U.H(U.np32(b)), U.P(r_d), U.P(out),
                                    B, H, W, Cin, Cout, ks, U.ACT[act], None), "op_conv2d")
    U.assert_close(U.from_dev_nhwc(out), ref, mode, f"conv{ks}x{ks} {Cin}->{Cout}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,act", [
    (2, 24, 24, 32, 64, None),   # RepViT / TinyViT patch-embed shape
    (1, 17, 13, 16, 32, "gelu"),  # odd sizes: ceil(H/2) outputs, bottom/right windows clipped
    (2, 10, 12, 64, 128, None), (1, 8, 8, 24, 48, None),
])
def test_conv3x3_stride2(mode, B, H, W, Cin, Cout, act):
    d, tdt = U.DT[mode]
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2) / (Cin * 9) ** 0.5
    b = _rand(Cout, seed=3) * 0.1
    ref = F.conv2d(_q(x, mode), _q(w, mode), b, stride=2, padding=1)
    ref = F.gelu(ref) if act == "gelu" else ref
    x_d = U.to_dev_nhwc(x, tdt)
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, Cout), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_conv3x3_s2(d, U.P(x_d), U.H(U.np32(w)), U.H(U.np32(b)), U.P(out), B, H, W, Cin, Cout,
                                        U.ACT[act], None), "op_conv3x3_s2")
    U.assert_close(U.from_dev_nhwc(out), ref, mode, f"conv3x3 s2 {Cin}->{Cout}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,heads,ws", [(2, 14, 14, 4, 7), (1, 18, 18, 2, 14), (2, 32, 32, 14, 7), (1, 63, 63, 8, 14),
                                            (3, 7, 7, 2, 7)])
def test_window_attention(mode, B, H, W, heads, ws):
    """TinyViT window attention incl. the zero-padded border windows whose padded tokens act as
    keys with the constant qkv(LayerNorm(0))."""
    d, tdt = U.DT[mode]
    C = heads * 32
    qkv = _q(_rand(B, H, W, 3 * C, seed=1), mode)
    pad_qkv = _q(_rand(3 * C, seed=2), mode)
    bias = _rand(heads, ws * ws, seed=3) * 0.5
    pad = (ws - H % ws) % ws
    full = pad_qkv.expand(B, H + pad, W + pad, 3 * C).clone()
    full[:, :H, :W] = qkv
    nw = (H + pad) // ws
    win = full.view(B, nw, ws, nw, ws, 3 * C).transpose(2, 3).reshape(B * nw * nw, ws * ws, heads, 96)
    q, k, v = [t.permute(0, 2, 1, 3) for t in win.split([32, 32, 32], dim=3)]
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    pts = torch.stack([ys.reshape(-1), xs.reshape(-1)], 1)
    dd = (pts[:, None] - pts[None]).abs()
    attn = (q @ k.transpose(-2, -1)) * 32 ** -0.5 + bias[:, dd[..., 0] * ws + dd[..., 1]]
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(B * nw * nw, ws * ws, C)
    ref = o.view(B, nw, nw, ws, ws, C).transpose(2, 3).reshape(B, H + pad, W + pad, C)[:, :H, :W]
    x_d = qkv.to(tdt).to("cuda").contiguous()
    out = torch.empty((B, H, W, C), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_window_attention(d, U.P(x_d), U.H(U.np32(pad_qkv)), U.H(U.np32(bias)), U.P(out), B, H, W,
                                              heads, ws, None), "op_window_attention")
    U.assert_close(out.float().cpu(), ref, mode, f"window attention ws={ws} heads={heads}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,ws,heads", [(1, 24, 24, 24, 2), (2, 48, 48, 24, 3), (1, 16, 16, 16, 1), (1, 72, 72, 72, 2),
                                            (2, 16, 16, 8, 2), (1, 10, 10, 5, 1)])
@pytest.mark.parametrize("rope", [False, True])
def test_vit_attention_window(mode, B, H, W, ws, heads, rope):
    """ViT-H softmax attention over ws x ws windows (ws = H: global).  bf16 with ws*ws % 64 == 0 runs the MFMA
    flash-attention kernel (P is rounded to bf16 before P.V, like the reference's bf16 SDPA); the other cases
    run the fp32 VALU kernel."""
    d, tdt = U.DT[mode]
    D = heads * 64
    qkv = _q(_rand(B, H, W, 3 * D, seed=1), mode)
    nw = H // ws
    t = qkv.view(B, nw, ws, nw, ws, 3, heads, 64).permute(5, 0, 1, 3, 6, 2, 4, 7).reshape(3, B * nw * nw, heads, ws * ws, 64)
    cs = None
    if rope:  # rotate q and k of every window token by its (cos, sin) row; bf16: rounded back like the engine
        ang = torch.rand(ws * ws, 32, generator=torch.Generator().manual_seed(7)) * 6.0
        cs = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous()
        def rot(u):
            a, b = u[..., 0::2], u[..., 1::2]
            return _q(torch.stack([a * cs[:, :, 0] - b * cs[:, :, 1], a * cs[:, :, 1] + b * cs[:, :, 0]], dim=-1).flatten(-2), mode)
        t = torch.stack([rot(t[0]), rot(t[1]), t[2]])
    o = F.scaled_dot_product_attention(t[0], t[1], t[2])  # [Bw, heads, N, 64]
    ref = o.view(B, nw, nw, heads, ws, ws, 64).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, H, W, D)
    x_d = qkv.to(tdt).to("cuda").contiguous()
    out = torch.empty((B, H, W, D), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_attn_window(d, U.P(x_d), U.H(U.np32(cs)) if rope else None, U.P(out), B, H, W, ws, heads,
                                         None), "op_attn_window")
    U.assert_close(out.float().cpu(), ref, mode, f"vit attention ws={ws} heads={heads}")


@pytest.mark.parametrize("mode", MODES)
def test_vit_rope(mode):
    d, tdt = U.DT[mode]
    B, H, W, ws, heads = 2, 12, 12, 6, 2
    D = heads * 64
    qkv = _q(_rand(B, H, W, 3 * D, seed=1), mode)
    ang = torch.rand(ws * ws, 32, generator=torch.Generator().manual_seed(2)) * 6.0
    cs = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous()
    ref = qkv.clone().view(B, H, W, 3, heads, 32, 2)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pos = (ys % ws) * ws + (xs % ws)
    c, s_ = torch.cos(ang)[pos][None, :, :, None], torch.sin(ang)[pos][None, :, :, None]
    for which in (0, 1):
        a, b = ref[:, :, :, which, :, :, 0].clone(), ref[:, :, :, which, :, :, 1].clone()
        ref[:, :, :, which, :, :, 0] = a * c - b * s_
        ref[:, :, :, which, :, :, 1] = a * s_ + b * c
    x_d = qkv.to(tdt).to("cuda").contiguous()
    U.check(U.lib().esam3_op_vit_rope(d, U.P(x_d), U.H(U.np32(cs)), B * H * W, H, W, ws, heads, None), "op_vit_rope")
    U.assert_close(x_d.float().cpu().view(B, H, W, 3, heads, 32, 2), _q(ref, mode) if mode == "bf16" else ref, mode, "vit rope")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,C,R", [(2, 63, 63, 256, 64), (3, 20, 31, 64, 16), (1, 32, 32, 512, 128), (2, 7, 5, 48, 16)])
def test_squeeze_excite(mode, B, H, W, C, R):
    d, tdt = U.DT[mode]
    x = _rand(B, C, H, W, seed=1)
    w1, b1 = _rand(R, C, seed=2) / C ** 0.5, _rand(R, seed=3) * 0.1
    w2, b2 = _rand(C, R, seed=4) / R ** 0.5, _rand(C, seed=5) * 0.5
    xq = _q(x, mode)
    g = xq.mean((2, 3), keepdim=True)
    g = F.relu(F.conv2d(g, w1[:, :, None, None], b1))
    g = torch.sigmoid(F.conv2d(g, w2[:, :, None, None], b2))
    ref = xq * g
    x_d = U.to_dev_nhwc(x, tdt)
    U.check(U.lib().esam3_op_squeeze_excite(d, U.P(x_d), U.H(U.np32(w1)), U.H(U.np32(b1)), U.H(U.np32(w2)),
                                            U.H(U.np32(b2)), B, H * W, C, R, None), "op_squeeze_excite")
    U.assert_close(U.from_dev_nhwc(x_d), ref, mode, f"squeeze-excite C={C}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,out_pad", [
    (2, 24, 24, 256, 256, 0),   # 256x256 LDS-DMA kernel, M = 1152 (ragged last tile)
    (1, 16, 16, 1024, 1024, 0),  # head.3 shape (16x16 patch tiles)
    (2, 32, 48, 256, 256, 0),   # several 16x16 patch tiles per image
    (1, 32, 32, 128, 256, 1),   # patch tiles + padded output
    (3, 19, 17, 256, 256, 1),   # ragged M + padded output
    (2, 12, 12, 256, 32, 0),    # composed 3x3 o conv_s0 shape -> 128x32 kernel with in_pad
    (1, 9, 9, 64, 64, 1),       # small-K register-staged path with in_pad
    (2, 32, 48, 256, 32, 0),    # bf16: conv3x3_narrow (16x16 patches, halo in LDS), N = 32, 8 chunks
    (1, 16, 16, 64, 64, 0),     # bf16: conv3x3_narrow, one tile, N = 64
    (3, 48, 32, 256, 64, 0),    # bf16: conv3x3_narrow, N = 64
    (2, 32, 96, 64, 32, 0),     # bf16: conv3x3_narrow, width a multiple of 2 and 3 patches (multi-patch workgroups)
    (5, 144, 144, 96, 32, 0),   # bf16: conv3x3_narrow, more tiles than resident workgroups (persistent loop), 3 chunks
])
def test_conv3x3_padded(mode, B, H, W, Cin, Cout, out_pad):
    """3x3 conv reading a zero-bordered NHWC input (no bounds checks; serves the DMA kernel)."""
    d, tdt = U.DT[mode]
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2) / (Cin * 9) ** 0.5
    b = _rand(Cout, seed=3) * 0.1
    ref = F.conv2d(_q(x, mode), _q(w, mode), b, padding=1)
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous().to("cuda", tdt)  # [B,H+2,W+2,C]
    if out_pad:
        out = torch.full((B, H + 2, W + 2, Cout), 7.0, dtype=tdt, device="cuda")
    else:
        out = torch.empty((B, H, W, Cout), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_conv3x3_padded(d, U.P(xp), U.H(U.np32(w)), U.H(U.np32(b)), U.P(out), B, H, W, Cin,
                                            Cout, 0, out_pad, None), "op_conv3x3_padded")
    got = out.float().cpu()
    if out_pad:
        assert float(got[:, 0].abs().max()) == 0 and float(got[:, -1].abs().max()) == 0
        assert float(got[:, :, 0].abs().max()) == 0 and float(got[:, :, -1].abs().max()) == 0
        got = got[:, 1:-1, 1:-1]
    U.assert_close(got.permute(0, 3, 1, 2), ref, mode, f"conv3x3 padded {Cin}->{Cout}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,act,res,after", [
    (2, 9, 9, 1024, 512, "gelu", False, 1), (1, 12, 10, 512, 256, None, False, 1),
    (2, 8, 8, 256, 64, None, True, 1), (2, 10, 10, 64, 32, "gelu", True, 0),
])
def test_conv_transpose(mode, B, H, W, Cin, Cout, act, res, after):
    d, tdt = U.DT[mode]
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cin, Cout, 2, 2, seed=2) / Cin ** 0.5
    b = _rand(Cout, seed=3) * 0.1
    r = _rand(B, Cout, 2 * H, 2 * W, seed=4) if res else None
    y = F.conv_transpose2d(_q(x, mode), _q(w, mode), b, stride=2)
    fn = {None: lambda t: t, "gelu": F.gelu}[act]
    if res and not after:
        ref = fn(y + _q(r, mode))
    else:
        ref = fn(y) + (_q(r, mode) if res else 0)
    x_d = U.to_dev_nhwc(x, tdt)
    r_d = U.to_dev_nhwc(r, tdt) if res else None
    out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_conv_transpose2x2(d, U.P(x_d), U.H(U.np32(w)), U.H(U.np32(b)), U.P(r_d), U.P(out),
                                               B, H, W, Cin, Cout, U.ACT[act], after, None), "op_convT")
    U.assert_close(U.from_dev_nhwc(out), ref, mode, f"convT {Cin}->{Cout}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,C,ks,stride,act", [
    (2, 17, 19, 64, 3, 1, "hswish"), (1, 18, 18, 128, 3, 2, "hswish"), (1, 15, 13, 16, 3, 2, None),
    (2, 11, 11, 384, 5, 1, None), (1, 63, 63, 256, 3, 2, "hswish"),
    # LDS-tiled kernel (stride 1, C % 64 == 0, >= 256 pixels): ragged tiles in both directions, several channel blocks
    (2, 63, 63, 384, 5, 1, None), (1, 32, 32, 1024, 3, 1, "hswish"), (3, 20, 17, 64, 3, 1, "hswish"), (1, 16, 16, 128, 5, 1, None),
])
def test_dwconv(mode, B, H, W, C, ks, stride, act):
    d, tdt = U.DT[mode]
    x = _rand(B, C, H, W, seed=1)
    w = _rand(C, 1, ks, ks, seed=2) / ks
    b = _rand(C, seed=3) * 0.1
    ref = F.conv2d(_q(x, mode), w, b, stride=stride, padding=ks // 2, groups=C)
    ref = F.hardswish(ref) if act else ref
    x_d = U.to_dev_nhwc(x, tdt)
    OH, OW = ref.shape[-2:]
    out = torch.empty((B, OH, OW, C), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_dwconv(d, U.P(x_d), U.H(U.np32(w)), U.H(U.np32(b)), U.P(out), B, H, W, C, ks,
                                    stride, U.ACT[act], None), "op_dwconv")
    U.assert_close(U.from_dev_nhwc(out), ref, mode, f"dw{ks} s{stride}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W,Cin,Cmid,Cout,stride,res", [
    (2, 40, 40, 16, 64, 32, 2, 0),     # stages.0.op_list.0 shape family
    (1, 30, 26, 32, 128, 32, 1, 1),    # residual MBConv, ragged tiles
    (2, 21, 19, 64, 256, 128, 2, 0),   # odd input size, stride 2 (126 -> 63 style)
    (1, 16, 16, 128, 512, 128, 1, 1),  # EfficientViTBlock local module (stage 3)
    (1, 9, 9, 256, 1024, 256, 1, 1),   # stage 4 local module: 16 channel chunks, Cout 256
    (1, 17, 17, 24, 96, 48, 2, 0),     # EfficientViT-B2 widths (Cin not a power of two)
    (2, 33, 50, 64, 256, 64, 1, 1),    # stages.1.op_list.1/2 shape (v2 kernel, ragged 8x16 tiles)
    (1, 41, 23, 32, 128, 64, 2, 0),    # stages.1.op_list.0 shape (v2, stride 2, odd sizes)
    (3, 64, 64, 16, 64, 32, 2, 0),     # several full tiles per image
])
def test_mbconv_fused(mode, B, H, W, Cin, Cmid, Cout, stride, res):
    """Fused expand -> dw3x3 -> project kernel vs the three-layer PyTorch reference
    (efficientvit/nn/ops.py:315-367 with Hardswish, BN folded into weights/biases)."""
    d, tdt = U.DT[mode]
    x = _rand(B, Cin, H, W, seed=1)
    w1, b1 = _rand(Cmid, Cin, 1, 1, seed=2) * (2.0 / Cin) ** 0.5, _rand(Cmid, seed=3) * 0.1
    wd, bd = _rand(Cmid, 1, 3, 3, seed=4) * 0.4, _rand(Cmid, seed=5) * 0.1
    w2, b2 = _rand(Cout, Cmid, 1, 1, seed=6) / Cmid ** 0.5, _rand(Cout, seed=7) * 0.1
    xq = _q(x, mode)
    m = F.hardswish(F.conv2d(xq, _q(w1, mode), b1))
    m = _q(m, mode)  # the kernel keeps the expanded tile in the activation dtype
    m = F.hardswish(F.conv2d(m, wd, bd, stride=stride, padding=1, groups=Cmid))
    m = _q(m, mode)
    ref = F.conv2d(m, _q(w2, mode), b2)
    if res:
        ref = ref + xq
    x_d = U.to_dev_nhwc(x, tdt)
    OH, OW = ref.shape[-2:]
    out = torch.empty((B, OH, OW, Cout), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_mbconv_fused(d, U.P(x_d), U.H(U.np32(w1)), U.H(U.np32(b1)), U.H(U.np32(wd)),
                                          U.H(U.np32(bd)), U.H(U.np32(w2)), U.H(U.np32(b2)), U.P(out), B, H, W,
                                          Cin, Cmid, Cout, stride, res, None), "op_mbconv_fused")
    U.assert_close(U.from_dev_nhwc(out), ref, mode, f"mbconv {Cin}->{Cmid}->{Cout} s{stride}", scale=2.0)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("Cout", [16, 8, 24])
def test_stem(mode, Cout):
    d, tdt = U.DT[mode]
    B, H, W = 2, 38, 42
    x = _rand(B, 3, H, W, seed=1)
    w = _rand(Cout, 3, 3, 3, seed=2) / 27 ** 0.5
    b = _rand(Cout, seed=3) * 0.1
    ref = F.hardswish(F.conv2d(x, w, b, stride=2, padding=1))
    x_d = x.to("cuda")
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, Cout), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_stem(d, U.P(x_d), U.H(U.np32(w)), U.H(U.np32(b)), U.P(out), B, H, W, Cout,
                                  U.ACT["hswish"], None), "op_stem")
    U.assert_close(U.from_dev_nhwc(out), ref, mode, "stem")


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 41, 37, 64, 128), (5, 126, 126, 64, 128), (2, 63, 63, 128, 256), (9, 40, 48, 128, 256)])
def test_patch_merging_fused(B, H, W, Cin, Cout):
    """TinyViT PatchMerging (tiny_vit.py:128-154: conv1 + BN -> GELU -> depthwise 3x3 stride 2 + BN -> GELU -> conv3 + BN) as ONE launch of the
    8-wave MBConv kernel (esam3_op_mbconv3 with residual = 4) against the same chain of torch functions with the bf16 rounding points of the
    layer-by-layer engine path (every stored tensor is bf16)."""
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(_rand(B, Cin, H, W, seed=1))
    w1, b1 = bf(_rand(Cout, Cin, seed=2) / Cin ** 0.5), _rand(Cout, seed=3) * 0.1
    wd, bd = _rand(Cout, 1, 3, 3, seed=4) / 3.0, _rand(Cout, seed=5) * 0.1
    w2, b2 = bf(_rand(Cout, Cout, seed=6) / Cout ** 0.5), _rand(Cout, seed=7) * 0.1
    wdq = bf(wd)    # the kernel multiplies bf16 taps on the matrix cores
    a = bf(F.gelu(F.conv2d(x, w1[:, :, None, None], b1)))
    d = bf(F.gelu(F.conv2d(a, wdq, bd, stride=2, padding=1, groups=Cout)))
    ref = F.conv2d(d, w2[:, :, None, None], b2)
    x_d = U.to_dev_nhwc(x, torch.bfloat16)
    OH, OW = ref.shape[-2:]
    out = torch.full((B, OH, OW, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    U.check(U.lib().esam3_op_mbconv3(U.P(x_d), U.H(U.np32(w1)), U.H(U.np32(b1)), U.H(U.np32(wd)), U.H(U.np32(bd)), U.H(U.np32(w2)),
                                     U.H(U.np32(b2)), U.P(out), B, H, W, Cin, Cout, Cout, 2, 4, None), "op_mbconv3 (PatchMerging)")
    got = U.from_dev_nhwc(out)
    assert torch.isfinite(got).all()
    assert _rel_l2(got, ref) < 6e-3, _rel_l2(got, ref)
    U.assert_close(got, ref, "bf16", f"PatchMerging {Cin}->{Cout}", scale=0.7)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,H,W", [(2, 38, 42), (3, 150, 134), (1, 64, 64)])
def test_stem_dsconv(mode, B, H, W):
    """The fused EfficientViT input stem (backbone.py:48-70: stem conv + Hardswish -> DSConv residual block) against torch;
    in bf16 the persistent kernel of round 6 must equal the one-tile-per-workgroup kernel bit for bit."""
    d, tdt = U.DT[mode]
    C = 16
    x = _rand(B, 3, H, W, seed=1)
    w0, b0 = _rand(C, 3, 3, 3, seed=2) / 27 ** 0.5, _rand(C, seed=3) * 0.1
    wd, bd = _rand(C, 1, 3, 3, seed=4) / 3.0, _rand(C, seed=5) * 0.1
    wp, bp = _rand(C, C, seed=6) / 4.0, _rand(C, seed=7) * 0.1
    y0 = F.hardswish(F.conv2d(x, w0, b0, stride=2, padding=1))
    y1 = F.hardswish(F.conv2d(y0, wd, bd, padding=1, groups=C))
    ref = y0 + F.conv2d(y1, wp[:, :, None, None], bp)
    x_d = x.to("cuda")
    OH, OW = (H + 1) // 2, (W + 1) // 2
    outs = []
    for variant in (0, 1):
        out = torch.full((B, OH, OW, C), float("nan"), dtype=tdt, device="cuda")
        U.check(U.lib().esam3_op_stem_dsconv(d, U.P(x_d), U.H(U.np32(w0)), U.H(U.np32(b0)), U.H(U.np32(wd)), U.H(U.np32(bd)),
                                             U.H(U.np32(wp)), U.H(U.np32(bp)), U.P(out), B, H, W, variant, None), "op_stem_dsconv")
        U.assert_close(U.from_dev_nhwc(out), ref, mode, f"stem_dsconv variant {variant}", scale=2.0)
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int16 if mode == "bf16" else torch.int32), outs[1].view(torch.int16 if mode == "bf16" else torch.int32))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,N,heads,dim", [(2, 700, 8, 16), (1, 3969, 8, 16), (3, 1024, 16, 16), (1, 500, 6, 32)])
def test_lite_mla(mode, B, N, heads, dim):
    """ops.py:584-621 on a [B, 2*heads groups x (q|k|v) x dim, N] tensor."""
    d, tdt = U.DT[mode]
    groups = 2 * heads
    ms = _rand(B, groups * 3 * dim, N, seed=1)
    msq = _q(ms, mode)
    t = msq.reshape(B, groups, 3 * dim, N)
    q, k, v = F.relu(t[:, :, :dim]), F.relu(t[:, :, dim:2 * dim]), t[:, :, 2 * dim:]
    v1 = F.pad(v, (0, 0, 0, 1), value=1.0)
    out = torch.matmul(torch.matmul(v1, k.transpose(-1, -2)), q)
    ref = (out[:, :, :-1] / (out[:, :, -1:] + 1e-15)).reshape(B, groups * dim, N)
    ms_d = ms.permute(0, 2, 1).contiguous().to("cuda", tdt)  # [B][N][C]
    o_d = torch.empty((B, N, groups * dim), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_lite_mla(d, U.P(ms_d), U.P(o_d), B, N, groups, dim, None), "op_lite_mla")
    U.assert_close(o_d.float().cpu().permute(0, 2, 1), ref, mode, "lite_mla")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("C,gs", [(384, 16), (768, 16), (576, 32)])
def test_grouped_pw(mode, C, gs):
    d, tdt = U.DT[mode]
    rows = 999
    x = _rand(1, C, rows, 1, seed=1)
    w = _rand(C, gs, 1, 1, seed=2) / gs ** 0.5
    ref = F.conv2d(_q(x, mode), w, None, groups=C // gs)
    x_d = x[0, :, :, 0].t().contiguous().to("cuda", tdt)
    o_d = torch.empty((rows, C), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_grouped_pw(d, U.P(x_d), U.H(U.np32(w)), U.P(o_d), rows, C, gs, None), "op_grouped_pw")
    U.assert_close(o_d.float().cpu().t(), ref[0, :, :, 0], mode, "grouped_pw")


@pytest.mark.parametrize("mode", MODES)
def test_resize_bilinear(mode):
    d, tdt = U.DT[mode]
    x = _rand(2, 64, 32, 32, seed=1)
    ref = F.interpolate(_q(x, mode), size=(72, 72), mode="bilinear", align_corners=False)
    o_d = torch.empty((2, 72, 72, 64), dtype=tdt, device="cuda")
    x_d = U.to_dev_nhwc(x, tdt)
    U.check(U.lib().esam3_op_resize_bilinear(d, U.P(x_d), U.P(o_d), 2, 32, 32, 72, 72, 64, None),
            "op_resize")
    U.assert_close(U.from_dev_nhwc(o_d), ref, mode, "resize")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("rows,C,eps,act,res", [(1000, 256, 1e-5, None, True), (777, 64, 1e-6, "gelu", False),
                                               (9, 256, 1e-5, None, False),
                                               # vectorised kernel: 3 chunks per lane (ViT-H width), 20 chunks on 32 lanes,
                                               # many row groups per wave, one chunk per row; C % 8 != 0 -> scalar kernel
                                               (300, 1280, 1e-6, None, True), (77, 160, 1e-5, None, False),
                                               (300000, 64, 1e-6, None, False), (50, 8, 1e-5, "gelu", True),
                                               (40, 2048, 1e-5, None, False), (33, 20, 1e-5, None, True)])
def test_layernorm(mode, rows, C, eps, act, res):
    d, tdt = U.DT[mode]
    x, r = _rand(rows, C, seed=1) * 2 + 0.3, _rand(rows, C, seed=2)
    g, b = _rand(C, seed=3) * 0.2 + 1, _rand(C, seed=4) * 0.1
    xin = _q(x, mode) + (_q(r, mode) if res else 0)
    ref = F.layer_norm(xin, (C,), g, b, eps)
    ref = F.gelu(ref) if act else ref
    o_d = torch.empty((rows, C), dtype=tdt, device="cuda")
    x_d, r_d = x.to("cuda", tdt), r.to("cuda", tdt)  # keep alive: U.P() only takes the address
    U.check(U.lib().esam3_op_layernorm(d, U.P(x_d), U.P(r_d) if res else None,
                                       U.H(U.np32(g)), U.H(U.np32(b)), U.P(o_d), rows, C, eps, U.ACT[act], None),
            "op_layernorm")
    U.assert_close(o_d.float().cpu(), ref, mode, "layernorm")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,Nq,Nk,heads,hd,few", [(2, 8, 5184, 8, 16, 0), (3, 10, 10, 8, 32, 0), (1, 23, 700, 8, 16, 0),
                                                 (2, 5184, 9, 8, 16, 1), (1, 1000, 1, 8, 16, 1),
                                                 # few keys: <= 8 / <= 16 keys in registers, ragged query count; > 16 -> generic
                                                 (3, 5001, 7, 8, 16, 1), (2, 777, 13, 8, 16, 1), (1, 300, 20, 8, 16, 1),
                                                 # the LDS-tiled token -> image kernel: full batch, ragged key count, one query
                                                 (32, 10, 5184, 8, 16, 0), (1, 16, 2001, 8, 16, 0), (3, 1, 1100, 8, 16, 0)])
def test_attention(mode, B, Nq, Nk, heads, hd, few):
    d, tdt = U.DT[mode]
    D = heads * hd
    q, k, v = _rand(B, Nq, D, seed=1), _rand(B, Nk, D, seed=2), _rand(B, Nk, D, seed=3)

    def split(t):
        return _q(t, mode).reshape(t.shape[0], t.shape[1], heads, hd).transpose(1, 2)

    ref = F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, Nq, D)
    o_d = torch.empty((B, Nq, D), dtype=tdt, device="cuda")
    q_d, k_d, v_d = q.to("cuda", tdt), k.to("cuda", tdt), v.to("cuda", tdt)  # keep alive
    U.check(U.lib().esam3_op_attention(d, U.P(q_d), U.P(k_d), U.P(v_d), U.P(o_d), B, Nq, Nk, heads, hd, few, None),
            "op_attention")
    U.assert_close(o_d.float().cpu(), ref, mode, "attention")


def _cpu_fill_holes(m, thr, max_area):
    from scipy import ndimage
    out = m.copy()
    st = np.ones((3, 3), dtype=np.int32)
    for i in range(m.shape[0]):
        bg = m[i] <= thr
        lab, n = ndimage.label(bg, structure=st)
        if n:
            areas = np.bincount(lab.ravel(), minlength=n + 1)
            out[i][(lab > 0) & (areas[lab] <= max_area)] = thr + 10.0
    return out


@pytest.mark.parametrize("kind", ["noise", "blobs", "all_bg", "all_fg", "checker"])
def test_fill_holes(kind):
    """Hole filling is integer work: bit-exact vs scipy 8-connected labelling."""
    rng = np.random.default_rng(7)
    n, H, W = 4, 288, 288
    if kind == "noise":
        m = rng.normal(0.3, 1.0, (n, H, W)).astype(np.float32)
    elif kind == "blobs":
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.ones((n, H, W), np.float32)
        for i in range(n):
            for _ in range(60):
                cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(1, 14)
                m[i][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = -1.0
    elif kind == "all_bg":
        m = -np.ones((n, H, W), np.float32)
    elif kind == "all_fg":
        m = np.ones((n, H, W), np.float32)
    else:
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.where(((yy // 3) + (xx // 5)) % 2 == 0, 1.0, -1.0).astype(np.float32)[None].repeat(n, 0)
    ref = _cpu_fill_holes(m, 0.0, 256.0)
    m_d = torch.from_numpy(m).to("cuda")
    o_d = torch.empty_like(m_d)
    U.check(U.lib().esam3_op_fill_holes(U.P(m_d), U.P(o_d), n, H, W, 0.0, 256.0, None), "op_fill_holes")
    assert np.array_equal(o_d.cpu().numpy(), ref), kind


def test_upsample_masks():
    m = _rand(3, 1, 288, 288, seed=5)
    for (oh, ow) in [(1008, 1008), (600, 800), (333, 517)]:
        ref = F.interpolate(m, (oh, ow), mode="bilinear", align_corners=False)[:, 0]
        f_d = torch.empty((3, oh, ow), dtype=torch.float32, device="cuda")
        u_d = torch.empty((3, oh, ow), dtype=torch.uint8, device="cuda")
        m_d = m.to("cuda")
        U.check(U.lib().esam3_op_upsample_masks(U.P(m_d), U.P(f_d), U.P(u_d), 3, 288, 288, oh, ow, 0.0, None),
                "op_upsample")
        got = f_d.cpu()
        assert float((got - ref).abs().max()) < 2e-5
        mism = (u_d.cpu().bool() != (ref > 0)) & ((ref.abs() > 1e-5))
        assert not mism.any()


@pytest.mark.parametrize("path,mode,B,Nq,Hk,Wk,masked,biased", [
    (0, "f32", 2, 37, 6, 8, False, True),      # VALU core, bias (decoder image cross-attention)
    (0, "f32", 2, 37, 1, 19, True, False),     # VALU core, key mask (text cross-attention)
    (0, "bf16", 2, 150, 6, 8, False, True),
    (1, "bf16", 2, 320, 8, 32, False, False),  # MFMA flash kernel (fusion-encoder self-attention)
    (1, "bf16", 1, 200, 8, 16, False, False),  # query remainder
    (2, "bf16", 2, 201, 12, 24, False, True),  # split-K MFMA: bias + presence row without bias
    (2, "bf16", 2, 1, 9, 16, False, False),    # one query (geometry CLS)
    (2, "bf16", 2, 45, 1, 139, True, False),   # key remainder + key mask
    (2, "bf16", 1, 33, 10, 20, True, True),
])
def test_mha_core(path, mode, B, Nq, Hk, Wk, masked, biased):
    """nn.MultiheadAttention core (8 heads x 32) of the PCS encoder / decoder on its three kernels against torch
    SDPA with the additive mask built the way the reference does (decoder.py:333-415 bias, key padding mask)."""
    d, tdt = U.DT[mode]
    heads, Nk, D = 8, Hk * Wk, 256
    q = _q(_rand(B, Nq, D, seed=1), mode)
    kv = _q(_rand(2, B, Nk, D, seed=2), mode)
    by = _rand(B, heads, Nq, Hk, seed=3) * 2.0
    bx = _rand(B, heads, Nq, Wk, seed=4) * 2.0
    km = torch.zeros(B, Nk, dtype=torch.bool)
    if masked:
        km = torch.rand(B, Nk, generator=torch.Generator().manual_seed(5)) < 0.3
        km[:, 0] = False
    add = torch.zeros(B, heads, Nq, Nk)
    if biased:
        add = (by[..., :, None] + bx[..., None, :]).reshape(B, heads, Nq, Nk).clone()
        add[:, :, 0] = 0.0                                           # bias_q0 = 1: the presence token row
    add = add.masked_fill(km[:, None, None, :], float("-inf"))
    sp = lambda t, n: t.view(B, n, heads, 32).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, Nq), sp(kv[0], Nk), sp(kv[1], Nk), attn_mask=add)
    ref = ref.transpose(1, 2).reshape(B, Nq, D)
    q_d, kv_d = q.to(tdt).cuda().contiguous(), kv.to(tdt).cuda().contiguous()
    by_d, bx_d, km_d = by.cuda().contiguous(), bx.cuda().contiguous(), km.to(torch.uint8).cuda().contiguous()
    out = torch.empty((B, Nq, D), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_mha(d, path, U.P(q_d), U.P(kv_d[0]), U.P(kv_d[1]), U.P(out), B, Nq, Nk, heads,
                                 U.P(km_d) if masked else None, U.P(by_d) if biased else None,
                                 U.P(bx_d) if biased else None, Hk, Wk, 1 if biased else 0, None), "op_mha")
    U.assert_close(out.float().cpu(), ref, mode, f"mha path={path} Nq={Nq} Nk={Nk}")


# ---- the phase-interleaved 256x256 kernel at sizes where every workgroup walks several output tiles ---------------
@pytest.mark.parametrize("case", ["neck_l0_3x3", "conv3x3_k576_padded_out", "linear_ragged_gelu_res", "convt_big"])
def test_gemm256p_many_tiles(case):
    """gemm256p_kernel (bf16) with more output tiles than CUs: the staging stream crosses output tiles, the persistent
    loop re-uses both LDS buffers and the epilogue strips.  Two launches must agree bit for bit (no race) and match
    the fp32 reference (necks.py:42-92 shapes, vitdet.py:585-590 MLP shape)."""
    d, tdt = U.DT["bf16"]
    lib = U.lib()
    if case in ("neck_l0_3x3", "conv3x3_k576_padded_out"):
        B, H, W, Cin, Cout, out_pad = (1, 288, 288, 256, 256, 0) if case == "neck_l0_3x3" else (3, 160, 160, 64, 256, 1)
        x = _rand(B, Cin, H, W, seed=1)
        w = _rand(Cout, Cin, 3, 3, seed=2) / (Cin * 9) ** 0.5
        b = _rand(Cout, seed=3) * 0.1
        ref = F.conv2d(_q(x, "bf16"), _q(w, "bf16"), b, padding=1)
        xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous().to("cuda", tdt)
        outs = []
        for _ in range(2):
            out = torch.full((B, H + 2 * out_pad, W + 2 * out_pad, Cout), 7.0, dtype=tdt, device="cuda")
            U.check(lib.esam3_op_conv3x3_padded(d, U.P(xp), U.H(U.np32(w)), U.H(U.np32(b)), U.P(out), B, H, W, Cin,
                                                Cout, 0, out_pad, None), "op_conv3x3_padded")
            outs.append(out.float().cpu())
        got = outs[0][:, 1:-1, 1:-1] if out_pad else outs[0]
        if out_pad:
            assert float(outs[0][:, 0].abs().max()) == 0 and float(outs[0][:, :, -1].abs().max()) == 0
        got = got.permute(0, 3, 1, 2)
    elif case == "linear_ragged_gelu_res":
        M, N, K = 20000, 4736, 1024
        a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2) / K ** 0.5, _rand(N, seed=3) * 0.1
        r = _rand(M, N, seed=4)
        ref = F.gelu(F.linear(_q(a, "bf16"), _q(w, "bf16"), b)) + _q(r, "bf16")
        a_d, r_d = a.to("cuda", tdt), r.to("cuda", tdt)
        outs = []
        for _ in range(2):
            out = torch.empty((M, N), dtype=tdt, device="cuda")
            U.check(lib.esam3_op_linear(d, U.P(a_d), U.H(U.np32(w)), U.H(U.np32(b)), U.P(r_d), U.P(out), M, N, K,
                                        U.ACT["gelu"], None), "op_linear")
            outs.append(out.float().cpu())
        got = outs[0]
    else:
        B, H, W, Cin, Cout = 4, 72, 72, 1024, 512
        x = _rand(B, Cin, H, W, seed=1)
        w = _rand(Cin, Cout, 2, 2, seed=2) / Cin ** 0.5
        b = _rand(Cout, seed=3) * 0.1
        ref = F.gelu(F.conv_transpose2d(_q(x, "bf16"), _q(w, "bf16"), b, stride=2))
        x_d = U.to_dev_nhwc(x, tdt)
        outs = []
        for _ in range(2):
            out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=tdt, device="cuda")
            U.check(lib.esam3_op_conv_transpose2x2(d, U.P(x_d), U.H(U.np32(w)), U.H(U.np32(b)), None, U.P(out), B, H, W,
                                                   Cin, Cout, U.ACT["gelu"], 1, None), "op_convT")
            outs.append(out.float().cpu())
        got = U.from_dev_nhwc(out)
    assert torch.equal(outs[0], outs[1]), "two launches differ: race in the staging pipeline"
    U.assert_close(got, ref, "bf16", f"gemm256p {case}")


@pytest.mark.parametrize("M,Cin,Hid,Cout,res", [(1000, 64, 128, 64, True), (4097, 64, 128, 64, True), (31, 64, 128, 64, False),
                                                (300001, 64, 128, 64, True)])
def test_fused_mlp(M, Cin, Hid, Cout, res):
    """fused_mlp.hip (RepViT channel mixer / TinyViT Mlp: 1x1 -> GELU -> 1x1 + shortcut in one launch, the hidden tensor in
    registers, the first GEMM's accumulator layout reused as the second GEMM's operand) against the two-layer torch
    reference with the hidden tensor rounded to bf16 where the layer-by-layer path stores it."""
    x, w1, b1 = _rand(M, Cin, seed=1), _rand(Hid, Cin, seed=2) / Cin ** 0.5, _rand(Hid, seed=3) * 0.1
    w2, b2 = _rand(Cout, Hid, seed=4) / Hid ** 0.5, _rand(Cout, seed=5) * 0.1
    r = _rand(M, Cout, seed=6) if res else None
    h = _q(F.gelu(F.linear(_q(x, "bf16"), _q(w1, "bf16"), b1)), "bf16")
    ref = F.linear(h, _q(w2, "bf16"), b2)
    if res:
        ref = ref + _q(r, "bf16")
    x_d = x.to("cuda", torch.bfloat16)
    r_d = r.to("cuda", torch.bfloat16) if res else None
    outs = []
    for _ in range(2):
        out = torch.empty((M, Cout), dtype=torch.bfloat16, device="cuda")
        U.check(U.lib().esam3_op_fused_mlp(U.P(x_d), U.H(U.np32(w1)), U.H(U.np32(b1)), U.H(U.np32(w2)), U.H(U.np32(b2)), U.P(r_d), U.P(out),
                                           M, Cin, Hid, Cout, U.ACT["gelu"], None), "op_fused_mlp")
        outs.append(out.float().cpu())
    assert torch.equal(outs[0], outs[1])
    U.assert_close(outs[0], ref, "bf16", f"fused_mlp {M}x{Cin}->{Hid}->{Cout}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,IH,OH,C,taps,act,pad", [(2, 32, 72, 64, 4, "gelu", 1), (1, 32, 72, 256, 1, None, 1), (2, 16, 36, 32, 4, None, 0),
                                                    (1, 7, 15, 8, 1, "gelu", 0), (1, 32, 32, 16, 4, None, 1)])
def test_resize_shuffle(mode, B, IH, OH, C, taps, act, pad):
    """resize_shuffle_kernel: bilinear resize of a ConvT-k2s2 / 1x1 layer's output computed on the small map, + bias,
    activation, pixel shuffle, zero border, against F.interpolate + torch's pixel shuffle of the same tensor.  Also the
    commutation the neck relies on: resize-then-layer == layer-then-resize (checked in fp32 on the torch side)."""
    d, tdt = U.DT[mode]
    y = _rand(B, taps * C, IH, IH, seed=1)          # the first layer's output on the small map, channels tap-major
    b = _rand(C, seed=2) * 0.1
    yq = _q(y, mode)
    up = F.interpolate(yq, size=(OH, OH), mode="bilinear", align_corners=False)          # [B, taps*C, OH, OH]
    if taps == 4:
        t = up.view(B, 2, 2, C, OH, OH)                                                   # [b, dy, dx, c, y, x]
        ref = t.permute(0, 3, 4, 1, 5, 2).reshape(B, C, 2 * OH, 2 * OH)
    else:
        ref = up
    ref = ref + b.view(1, C, 1, 1)
    if act == "gelu":
        ref = F.gelu(ref)
    s = 2 if taps == 4 else 1
    x_d = U.to_dev_nhwc(y, tdt)
    out = torch.full((B, s * OH + 2 * pad, s * OH + 2 * pad, C), 7.0, dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_resize_shuffle(d, U.P(x_d), U.H(U.np32(b)), U.P(out), B, IH, IH, OH, OH, C, taps, U.ACT[act], pad, None),
            "op_resize_shuffle")
    got = U.from_dev_nhwc(out)
    if pad:
        assert (got[:, :, 0, :] == 7.0).all() and (got[:, :, :, -1] == 7.0).all()       # the border belongs to the caller
        got = got[:, :, 1:-1, 1:-1]
    U.assert_close(got, ref, mode, f"resize_shuffle {IH}->{OH} C={C} taps={taps}")
    # the algebra: a per-pixel linear map commutes with the interpolation (weights sum to one, so the bias does too)
    wmat = _rand(5, C, seed=3)
    a_ = torch.einsum("oc,bchw->bohw", wmat, F.interpolate(y[:, :C], size=(OH, OH), mode="bilinear", align_corners=False)) + 0.3
    b_ = F.interpolate(torch.einsum("oc,bchw->bohw", wmat, y[:, :C]) + 0.3, size=(OH, OH), mode="bilinear", align_corners=False)
    assert float((a_ - b_).abs().max()) < 1e-4


def _rel_l2(got, ref):
    return float((got.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,H,W,Cin,Cmid,Cout,stride,res", [
    (2, 40, 40, 16, 64, 32, 2, 0),      # stages.0.op_list.0
    (1, 30, 26, 32, 128, 32, 1, 1),     # stages.0.op_list.1, ragged 8x16 tiles
    (1, 41, 23, 32, 128, 64, 2, 0),     # stages.1.op_list.0, odd sizes
    (2, 33, 50, 64, 256, 64, 1, 1),     # stages.1.op_list.1/2
    (2, 21, 19, 64, 256, 128, 2, 0),    # stages.2.op_list.0 (126 -> 63 style)
    (1, 16, 16, 128, 512, 128, 1, 1),   # EfficientViTBlock local module, stage 3 (one full tile row pair)
    (2, 63, 63, 128, 512, 128, 1, 1),   # the same at the real stage-3 size (ragged right / bottom tiles)
    (1, 31, 29, 128, 512, 256, 2, 0),   # stages.3.op_list.0: stride 2, 8 waves
    (1, 9, 9, 256, 1024, 256, 1, 1),    # stage-4 local module: 16 chunks, Cout 256, 8 waves
    (2, 32, 32, 256, 1024, 256, 1, 1),  # the same at the real stage-4 size
    (3, 64, 64, 16, 64, 32, 2, 0),      # several full tiles per image
])
def test_mbconv3(B, H, W, Cin, Cmid, Cout, stride, res):
    """Round-4 fused MBConv (csrc/evit_fused.hip: expand on MFMA -> depthwise 3x3 on v_mfma_f32_4x4x4 -> project on MFMA) vs the
    three-layer PyTorch reference (efficientvit/nn/ops.py:315-367 with Hardswish, BN folded), with the kernel's rounding points:
    bf16 activations between the layers, bf16 depthwise weights (the reference's autocast rounds them too)."""
    mode = "bf16"
    d, tdt = U.DT[mode]
    x = _rand(B, Cin, H, W, seed=1)
    w1, b1 = _rand(Cmid, Cin, 1, 1, seed=2) * (2.0 / Cin) ** 0.5, _rand(Cmid, seed=3) * 0.1
    wd, bd = _rand(Cmid, 1, 3, 3, seed=4) * 0.4, _rand(Cmid, seed=5) * 0.1
    w2, b2 = _rand(Cout, Cmid, 1, 1, seed=6) / Cmid ** 0.5, _rand(Cout, seed=7) * 0.1
    xq = _q(x, mode)
    m = _q(F.hardswish(F.conv2d(xq, _q(w1, mode), b1)), mode)
    m = _q(F.hardswish(F.conv2d(m, _q(wd, mode), bd, stride=stride, padding=1, groups=Cmid)), mode)
    ref = F.conv2d(m, _q(w2, mode), b2)
    if res:
        ref = ref + xq
    x_d = U.to_dev_nhwc(x, tdt)
    OH, OW = ref.shape[-2:]
    out = torch.full((B, OH, OW, Cout), float("nan"), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_mbconv3(U.P(x_d), U.H(U.np32(w1)), U.H(U.np32(b1)), U.H(U.np32(wd)), U.H(U.np32(bd)),
                                     U.H(U.np32(w2)), U.H(U.np32(b2)), U.P(out), B, H, W, Cin, Cmid, Cout, stride, res, None),
            "op_mbconv3")
    got = U.from_dev_nhwc(out)
    assert torch.isfinite(got).all()
    assert _rel_l2(got, ref) < 6e-3, _rel_l2(got, ref)
    U.assert_close(got, ref, mode, f"mbconv3 {Cin}->{Cmid}->{Cout} s{stride}", scale=0.7)


@pytest.mark.parametrize("B,H,W", [(1, 30, 26), (2, 33, 50), (2, 64, 64), (1, 252, 252)])
def test_mbconv3_gelu_variant(B, H, W):
    """Round 6: TinyViT's MBConv (tiny_vit.py:73-108: conv1 + BN -> GELU -> depthwise 3x3 + BN -> GELU -> conv3 + BN -> + x -> GELU,
    64 -> 256 -> 64) on the persistent matrix-core kernel of the EfficientViT MBConvs (mbconv3s with GELU epilogues) vs the layer-by-layer
    PyTorch reference with the kernel's rounding points; and against the round-2 kernel it replaces (same formula, same GELU polynomial:
    the two must agree to bf16 rounding of the intermediates)."""
    mode = "bf16"
    d, tdt = U.DT[mode]
    Cin, Cmid, Cout = 64, 256, 64
    x = _rand(B, Cin, H, W, seed=1)
    w1, b1 = _rand(Cmid, Cin, 1, 1, seed=2) * (2.0 / Cin) ** 0.5, _rand(Cmid, seed=3) * 0.1
    wd, bd = _rand(Cmid, 1, 3, 3, seed=4) * 0.4, _rand(Cmid, seed=5) * 0.1
    w2, b2 = _rand(Cout, Cmid, 1, 1, seed=6) / Cmid ** 0.5, _rand(Cout, seed=7) * 0.1
    xq = _q(x, mode)
    m = _q(F.gelu(F.conv2d(xq, _q(w1, mode), b1)), mode)
    m = _q(F.gelu(F.conv2d(m, _q(wd, mode), bd, padding=1, groups=Cmid)), mode)
    ref = F.gelu(F.conv2d(m, _q(w2, mode), b2) + xq)
    x_d = U.to_dev_nhwc(x, tdt)
    args = (U.H(U.np32(w1)), U.H(U.np32(b1)), U.H(U.np32(wd)), U.H(U.np32(bd)), U.H(U.np32(w2)), U.H(U.np32(b2)))
    out = torch.full((B, H, W, Cout), float("nan"), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_mbconv3(U.P(x_d), *args, U.P(out), B, H, W, Cin, Cmid, Cout, 1, 3, None), "op_mbconv3 (GELU)")
    got = U.from_dev_nhwc(out)
    assert torch.isfinite(got).all()
    assert _rel_l2(got, ref) < 6e-3, _rel_l2(got, ref)
    U.assert_close(got, ref, mode, "mbconv3 GELU 64->256->64", scale=0.7)
    old = torch.full((B, H, W, Cout), float("nan"), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_mbconv_fused(d, U.P(x_d), *args, U.P(old), B, H, W, Cin, Cmid, Cout, 1, 3, None), "op_mbconv_fused (GELU)")
    assert _rel_l2(got, U.from_dev_nhwc(old)) < 4e-3, _rel_l2(got, U.from_dev_nhwc(old))
    with pytest.raises(Exception):   # the variant is one shape: anything else is refused, not run on another kernel
        U.check(U.lib().esam3_op_mbconv3(U.P(x_d), *args, U.P(out), B, H, W, Cin, Cmid, Cout, 1, 2, None), "no shortcut")


def _lite_mla_block_ref(x, wqkv, wdw, wgrp, wproj, bproj, dim=16):
    """ops.py:521-671 + the ResidualBlock shortcut, with the fused kernels' rounding points (bf16 tensors between the layers)."""
    q = lambda t: t.to(torch.bfloat16).float()
    B, C, H, W = x.shape
    xq = q(x)
    qkv = q(F.conv2d(xq, q(wqkv)[:, :, None, None]))
    agg = q(F.conv2d(qkv, q(wdw), padding=2, groups=3 * C))
    agg = q(F.conv2d(agg, q(wgrp)[:, :, None, None], groups=3 * C // dim))
    ms = torch.cat([qkv, agg], dim=1).reshape(B, -1, 3 * dim, H * W).double()
    qq, kk, vv = F.relu(ms[:, :, :dim]), F.relu(ms[:, :, dim:2 * dim]), ms[:, :, 2 * dim:]
    v1 = F.pad(vv, (0, 0, 0, 1), value=1.0)
    o = torch.matmul(torch.matmul(v1, kk.transpose(-1, -2)), qq)
    att = q((o[:, :, :-1] / (o[:, :, -1:] + 1e-15)).reshape(B, -1, H, W).float())
    return F.conv2d(att, q(wproj)[:, :, None, None], bproj) + xq


@pytest.mark.parametrize("B,H,W,C", [
    (1, 8, 16, 128),     # exactly one tile
    (2, 21, 19, 128),    # ragged tiles, several partial kv tiles per image
    (1, 63, 63, 128),    # the real stage-3 map
    (1, 9, 9, 256),      # stage-4 width, one partial tile
    (2, 32, 32, 256),    # the real stage-4 map
])
def test_lite_mla_block(B, H, W, C):
    """Fused LiteMLA context module (csrc/evit_fused.hip: mla1 -> kvprep -> mla2) against the reference formula in fp64 with
    the kernels' bf16 rounding points."""
    tdt = torch.bfloat16
    x = _rand(B, C, H, W, seed=1)
    wqkv = _rand(3 * C, C, seed=2) / C ** 0.5
    wdw = _rand(3 * C, 1, 5, 5, seed=3) * 0.2
    wgrp = _rand(3 * C, 16, seed=4) / 4.0
    wproj = _rand(C, 2 * C, seed=5) / (2 * C) ** 0.5
    bproj = _rand(C, seed=6) * 0.1
    ref = _lite_mla_block_ref(x, wqkv, wdw, wgrp, wproj, bproj)
    x_d = U.to_dev_nhwc(x, tdt)
    out = torch.full((B, H, W, C), float("nan"), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_lite_mla_block(U.P(x_d), U.H(U.np32(wqkv)), U.H(U.np32(wdw)), U.H(U.np32(wgrp)), U.H(U.np32(wproj)),
                                            U.H(U.np32(bproj)), U.P(out), B, H, W, C, None), "op_lite_mla_block")
    got = U.from_dev_nhwc(out)
    assert torch.isfinite(got).all()
    assert _rel_l2(got, ref) < 8e-3, _rel_l2(got, ref)
    U.assert_close(got, ref, "bf16", f"lite_mla_block C={C} {H}x{W}", scale=1.0)


def _i2t_block_ref(x, wq, bq, peq, wo, bo, gamma, beta, tk, tv):
    """sam/transformer.py:177-182 (cross_attn_image_to_token + norm4) in fp64 with the kernel's rounding points: bf16 inputs and
    weights, q and the attention output rounded to bf16, the probabilities rounded to bf16 for the P.V product."""
    r = lambda t: t.to(torch.bfloat16).double()
    Bp, P, _ = x.shape
    T = tk.shape[1]
    xq = r(x)
    q = r((xq @ r(wq).T + bq.double() + r(peq)[None]).float())
    qh = q.reshape(Bp, P, 8, 16).permute(0, 2, 1, 3)
    kh = r(tk).reshape(Bp, T, 8, 16).permute(0, 2, 1, 3)
    vh = r(tv).reshape(Bp, T, 8, 16).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) * 0.25
    e = r(torch.exp(s - s.max(dim=-1, keepdim=True).values).float())
    o = (e @ vh) / e.sum(dim=-1, keepdim=True)
    o = r(o.permute(0, 2, 1, 3).reshape(Bp, P, 128).float())
    y = o @ r(wo).T + bo.double() + xq
    return F.layer_norm(y, (256,), gamma.double(), beta.double(), 1e-5).float()


@pytest.mark.parametrize("Bp,P,T", [
    (1, 16, 7),        # one group of 16 rows
    (2, 5184, 10),     # the real 72 x 72 map, point + box prompt
    (3, 144, 16),      # the most tokens the kernel takes
    (5, 80, 1),        # a single key: softmax = 1
    (37, 48, 9),       # more prompts than a wave's run of groups: the token operands are reloaded inside a run
])
def test_i2t_block(Bp, P, T):
    """Fused image-to-token block of the two-way transformer (csrc/decoder_fused.hip) against the reference formula in fp64."""
    tdt = torch.bfloat16
    x = _rand(Bp, P, 256, seed=1)
    wq, bq = _rand(128, 256, seed=2) / 16.0, _rand(128, seed=3) * 0.1
    peq = _rand(P, 128, seed=4) * 0.5
    wo, bo = _rand(256, 128, seed=5) / 128 ** 0.5, _rand(256, seed=6) * 0.1
    gamma, beta = torch.rand(256, generator=torch.Generator().manual_seed(7)) + 0.5, _rand(256, seed=8) * 0.1
    tk, tv = _rand(Bp, T, 128, seed=9), _rand(Bp, T, 128, seed=10)
    ref = _i2t_block_ref(x, wq, bq, peq, wo, bo, gamma, beta, tk, tv)
    x_d = x.to("cuda", tdt).contiguous()
    out = torch.full((Bp, P, 256), float("nan"), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_i2t_block(U.P(x_d), U.H(U.np32(wq)), U.H(U.np32(bq)), U.H(U.np32(peq)), U.H(U.np32(wo)), U.H(U.np32(bo)),
                                       U.H(U.np32(gamma)), U.H(U.np32(beta)), U.H(U.np32(tk)), U.H(U.np32(tv)), U.P(out), Bp, P, T, None),
            "op_i2t_block")
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert _rel_l2(got, ref) < 6e-3, _rel_l2(got, ref)
    U.assert_close(got, ref, "bf16", f"i2t_block Bp={Bp} P={P} T={T}", scale=1.0)
    # in place, as the engine runs it
    U.check(U.lib().esam3_op_i2t_block(U.P(x_d), U.H(U.np32(wq)), U.H(U.np32(bq)), U.H(U.np32(peq)), U.H(U.np32(wo)), U.H(U.np32(bo)),
                                       U.H(U.np32(gamma)), U.H(U.np32(beta)), U.H(U.np32(tk)), U.H(U.np32(tv)), U.P(x_d), Bp, P, T, None),
            "op_i2t_block in place")
    assert torch.equal(x_d.float().cpu(), got)


@pytest.mark.parametrize("B,Nq,Nk,merged", [(32, 10, 5184, 1), (1, 16, 2001, 0), (3, 1, 1100, 1), (2, 7, 64, 0), (1, 9, 65, 1), (2, 10, 3, 0)])
def test_attention_t2i_mfma(B, Nq, Nk, merged):
    """token -> image attention on the matrix cores (csrc/decoder_fused.hip: t2i_mfma_kernel + the fixed-order merge), with k / v as
    separate tensors and as the two halves of the merged [k | v] projection rows"""
    tdt = torch.bfloat16
    heads, hd, D = 8, 16, 128
    q, k, v = _rand(B, Nq, D, seed=1), _rand(B, Nk, D, seed=2), _rand(B, Nk, D, seed=3)

    def split(t):
        return _q(t, "bf16").reshape(t.shape[0], t.shape[1], heads, hd).transpose(1, 2)

    ref = F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, Nq, D)
    o_d = torch.full((B, Nq, D), float("nan"), dtype=tdt, device="cuda")
    q_d = q.to("cuda", tdt)
    if merged:
        kv_d = torch.cat([k, v], dim=-1).to("cuda", tdt).contiguous()
        U.check(U.lib().esam3_op_attention(1, U.P(q_d), U.P(kv_d), None, U.P(o_d), B, Nq, Nk, heads, hd, 3, None), "op_attention 3")
    else:
        k_d, v_d = k.to("cuda", tdt), v.to("cuda", tdt)
        U.check(U.lib().esam3_op_attention(1, U.P(q_d), U.P(k_d), U.P(v_d), U.P(o_d), B, Nq, Nk, heads, hd, 2, None), "op_attention 2")
    U.assert_close(o_d.float().cpu(), ref, "bf16", "attention t2i mfma")


@pytest.mark.parametrize("rows,P,bias,table", [(32, 32, True, True), (5184 * 2, 5184, True, True), (96, 0, False, False), (64 * 37, 64, True, False),
                                              (5184 * 3 + 0, 5184, False, True)])
def test_rowlin256(rows, P, bias, table):
    """256 -> 256 row-wise linear with a position table (csrc/decoder_fused.hip: rowlin256_kernel, the merged k | v projection)"""
    tdt = torch.bfloat16
    x = _rand(rows, 256, seed=1)
    w = _rand(256, 256, seed=2) / 16.0
    b = _rand(256, seed=3) * 0.2 if bias else None
    tb = _rand(P, 256, seed=4) * 0.5 if table else None
    q = lambda t: t.to(tdt).double()  # noqa: E731
    ref = q(x) @ q(w).T
    if bias:
        ref = ref + b.double()
    if table:
        ref = ref + q(tb).repeat(rows // P, 1)
    x_d = x.to("cuda", tdt)
    out = torch.full((rows, 256), float("nan"), dtype=tdt, device="cuda")
    U.check(U.lib().esam3_op_rowlin256(U.P(x_d), U.H(U.np32(w)), U.H(U.np32(b)) if bias else None, U.H(U.np32(tb)) if table else None, P, U.P(out),
                                       rows, None), "op_rowlin256")
    U.assert_close(out.float().cpu(), ref.float(), "bf16", f"rowlin256 rows={rows}")
