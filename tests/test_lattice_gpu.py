"""EXACT-arithmetic tests of the bf16-only fused kernels ("What's weak" 2: the north-star tolerance is proven on
the f32 layer-by-layer graph, the benchmarked bf16 graph runs kernels that have no f32 instantiation).

The fused kernels are MFMA lane-layout chains (evit_fused.hip, decoder_fused.hip, fused_mlp.hip): their failure mode is a wrong lane,
a wrong tap, a wrong tile edge or a wrong K / N index -- errors a 6e-3 relative-L2 bound against a formula with the kernel's
own rounding points can hide.  Here every operand is chosen on an integer lattice on which EVERY intermediate of the kernel is
exactly representable in bf16 and every fp32 accumulation is exact in any order:

  * inputs and weights are small integers (weights sparse with entries +-1, so sums stay below 2^8 where bf16 holds them);
  * Hardswish is exact on multiples of 3 (x <= -3 -> 0, x >= 3 -> x; also in the kernel's fma / clamp form), GELU on multiples of 6
    to within 1e-8 (absorbed by the integer it is added to);
  * softmax attention is made one-hot (the selected key's score is >= 40 above every other, exp(-40) ~ 4e-18 vanishes in fp32), so
    the output is one row of V bit for bit;
  * the one inexact operation of LiteMLA (the fp32 division) is IEEE in the kernel and in torch, on exact integer operands.

The kernel's output must then equal the fp64 formula BIT FOR BIT (torch.equal): a single swapped lane, tap or channel
anywhere in a fused graph changes an integer somewhere.  Where a vanishing term exists (the e^-40 tails of a one-hot softmax, GELU's
-6e-9 at -6) it is absorbed by every non-zero integer it is added to but survives next to an exact ZERO: there `_same_integers`
asks for bit equality of every non-zero expected value and |got| <= 1e-6 where zero is expected (first GPU run of round 5: exactly the
1 / 17 of the one-hot attention outputs whose selected value is 0 came back as 1e-17).  The i2t block ends in a LayerNorm (a
reduction whose fp32 result depends on the order): there every element must be within one bf16 ulp (or 1e-5 near zero) and >= 99.5 %
of them bit-equal.

These tests complement, not replace, the random-data tests of test_ops_gpu.py (which exercise rounding) and the end-to-end
fixtures (which exercise the composition)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests import util as U  # noqa: E402


def _ints(*shape, lo, hi, seed, mult=1):
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(lo, hi + 1, shape, generator=g) * mult).float()


def _sparse_pm1(rows, cols, nnz, seed):
    """[rows, cols] with `nnz` entries +-1 per row at random distinct columns; every column is used by some row when rows * nnz >= cols"""
    g = torch.Generator().manual_seed(seed)
    w = torch.zeros(rows, cols)
    base = torch.randperm(cols, generator=g)
    for r in range(rows):
        idx = torch.stack([base[(r * nnz + j) % cols] for j in range(nnz)]) if nnz <= cols else torch.arange(cols)
        if len(set(idx.tolist())) < nnz:
            idx = torch.randperm(cols, generator=g)[:nnz]
        sign = torch.randint(0, 2, (nnz,), generator=g).float() * 2 - 1
        w[r, idx] = sign
    return w


def _bf16_exact(t):
    return torch.equal(t.to(torch.bfloat16).float(), t.float())


def _same_integers(got, ref, what=""):
    """bit equality where the expected integer is non-zero, |got| <= 1e-6 where it is zero (module docstring)"""
    ref = ref.float()
    nz = ref != 0
    bad = (nz & (got != ref)) | (~nz & (got.abs() > 1e-6))
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements differ on exact data; first at {tuple(bad.nonzero()[0].tolist())}: "
                           f"got {float(got[tuple(bad.nonzero()[0].tolist())])!r}, expected {float(ref[tuple(bad.nonzero()[0].tolist())])!r}")


@pytest.mark.parametrize("B,H,W,Cin,Cmid,Cout,stride,res", [
    (2, 40, 40, 16, 64, 32, 2, 0), (1, 30, 26, 32, 128, 32, 1, 1), (1, 41, 23, 32, 128, 64, 2, 0), (2, 33, 50, 64, 256, 64, 1, 1),
    (2, 21, 19, 64, 256, 128, 2, 0), (1, 16, 16, 128, 512, 128, 1, 1), (2, 63, 63, 128, 512, 128, 1, 1), (1, 31, 29, 128, 512, 256, 2, 0),
    (1, 9, 9, 256, 1024, 256, 1, 1), (2, 32, 32, 256, 1024, 256, 1, 1), (3, 64, 64, 16, 64, 32, 2, 0),
    (4, 126, 126, 64, 256, 64, 1, 1),   # many persistent-loop iterations per workgroup (tile hand-over, prefetched fragments)
    (9, 126, 126, 64, 256, 128, 2, 0),  # round 6: the stride-2 form of the 8-wave kernel, several tiles per workgroup
])
def test_mbconv3_exact_on_the_lattice(B, H, W, Cin, Cmid, Cout, stride, res):
    """every fused MBConv variant (mbconv3s / mbconv3b / mbconv3 generic) on multiples of 3: expand -> Hardswish -> depthwise ->
    Hardswish -> project (+ shortcut) with all intermediates exact"""
    x, w1, b1, wd, bd, w2, b2, ref = mbconv_lattice(B, H, W, Cin, Cmid, Cout, stride, res)
    x_d = U.to_dev_nhwc(x, torch.bfloat16)
    OH, OW = ref.shape[-2:]
    out = torch.full((B, OH, OW, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    U.check(U.lib().esam3_op_mbconv3(U.P(x_d), U.H(U.np32(w1[:, :, None, None])), U.H(U.np32(b1)), U.H(U.np32(wd)), U.H(U.np32(bd)),
                                     U.H(U.np32(w2[:, :, None, None])), U.H(U.np32(b2)), U.P(out), B, H, W, Cin, Cmid, Cout, stride, res, None),
            "op_mbconv3")
    got = U.from_dev_nhwc(out)
    bad = got != ref.float()
    assert not bad.any(), f"{int(bad.sum())}/{bad.numel()} elements differ on exact data; first at {tuple(bad.nonzero()[0].tolist())}"


def mbconv_lattice(B, H, W, Cin, Cmid, Cout, stride, res):
    x = _ints(B, Cin, H, W, lo=-2, hi=2, seed=1, mult=3)
    w1, b1 = _sparse_pm1(Cmid, Cin, 2, seed=2), _ints(Cmid, lo=-1, hi=1, seed=3, mult=3)
    wd = torch.zeros(Cmid, 9)
    g = torch.Generator().manual_seed(4)
    for c in range(Cmid):
        taps = torch.randperm(9, generator=g)[:3]
        wd[c, taps] = torch.randint(0, 2, (3,), generator=g).float() * 2 - 1
    wd = wd.reshape(Cmid, 1, 3, 3)
    bd = _ints(Cmid, lo=-1, hi=1, seed=5, mult=3)
    w2, b2 = _sparse_pm1(Cout, Cmid, 3, seed=6), _ints(Cout, lo=-2, hi=2, seed=7)
    m = F.hardswish(F.conv2d(x.double(), w1.double()[:, :, None, None], b1.double()))
    assert _bf16_exact(m) and float(m.max()) > 0
    m = F.hardswish(F.conv2d(m, wd.double(), bd.double(), stride=stride, padding=1, groups=Cmid))
    assert _bf16_exact(m) and float(m.max()) > 0
    ref = F.conv2d(m, w2.double()[:, :, None, None], b2.double())
    if res:
        ref = ref + x.double()
    assert _bf16_exact(ref) and float(ref.abs().max()) < 256
    return x, w1, b1, wd, bd, w2, b2, ref


def _mla_lattice_ref(x, wqkv, wdw, wgrp, wsel, dim=16):
    """ops.py:521-671 + shortcut on integers: everything exact up to ONE correctly rounded fp32 division per (token, channel), then the
    bf16 rounding of the attention output, a +-1 selection as proj, the shortcut add in fp32 and the output's bf16 rounding"""
    B, C, H, W = x.shape
    qkv = F.conv2d(x.double(), wqkv.double()[:, :, None, None])
    agg = F.conv2d(qkv, wdw.double(), padding=2, groups=3 * C)
    agg = F.conv2d(agg, wgrp.double()[:, :, None, None], groups=3 * C // dim)
    assert _bf16_exact(qkv) and _bf16_exact(agg)
    ms = torch.cat([qkv, agg], dim=1).reshape(B, -1, 3 * dim, H * W)
    qq, kk, vv = F.relu(ms[:, :, :dim]), F.relu(ms[:, :, dim:2 * dim]), ms[:, :, 2 * dim:]
    v1 = F.pad(vv, (0, 0, 0, 1), value=1.0)
    kv = torch.matmul(v1, kk.transpose(-1, -2))            # [B, G, dim + 1, dim] exact integers
    assert float(kv.abs().max()) < 65536                  # the kernel holds kv as bf16 hi + lo: 16 significant bits
    o = torch.matmul(kv, qq)
    assert float(o.abs().max()) < 2 ** 24
    num, den = o[:, :, :-1].float(), o[:, :, -1:].float()
    att = (num / (den + torch.tensor(1e-15, dtype=torch.float32))).to(torch.bfloat16).float()      # fp32 division, as the kernel's
    att = att.reshape(B, -1, H, W)
    y = F.conv2d(att.double(), wsel.double()[:, :, None, None]).float() + x.float()            # one term per output: exact; one fp32 add
    return y.to(torch.bfloat16).float()


@pytest.mark.parametrize("B,H,W,C", [(1, 8, 16, 128), (2, 21, 19, 128), (1, 63, 63, 128), (1, 9, 9, 256), (2, 32, 32, 256),
                                     (11, 63, 63, 128)])   # round 6: 352 tiles on 256 persistent workgroups (tile hand-over, prefetched fragments)
def test_lite_mla_block_exact_on_the_lattice(B, H, W, C):
    """mla1 -> kvprep -> mla2 on integers: qkv GEMM, 5x5 depthwise (one random tap per channel = a shifted copy: checks every tap
    position and the halo), grouped 1x1 (a signed permutation inside each group of 16), kv / ksum sums over all pixels of the image
    (tile partials, fixed-order merge, bf16 hi + lo split), the per-head products, the division and the projection"""
    x, wqkv, wdw, wgrp, wsel, ref = mla_lattice(B, H, W, C)
    x_d = U.to_dev_nhwc(x, torch.bfloat16)
    out = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    U.check(U.lib().esam3_op_lite_mla_block(U.P(x_d), U.H(U.np32(wqkv)), U.H(U.np32(wdw)), U.H(U.np32(wgrp)), U.H(U.np32(wsel)),
                                            U.H(np.zeros(C, np.float32)), U.P(out), B, H, W, C, None), "op_lite_mla_block")
    got = U.from_dev_nhwc(out)
    bad = got != ref
    assert not bad.any(), f"{int(bad.sum())}/{bad.numel()} elements differ on exact data; first at {tuple(bad.nonzero()[0].tolist())}"


def mla_lattice(B, H, W, C):
    x = _ints(B, C, H, W, lo=-1, hi=1, seed=1)
    wqkv = _sparse_pm1(3 * C, C, 2, seed=2)
    g = torch.Generator().manual_seed(3)
    wdw = torch.zeros(3 * C, 25)
    wdw[torch.arange(3 * C), torch.randint(0, 25, (3 * C,), generator=g)] = torch.randint(0, 2, (3 * C,), generator=g).float() * 2 - 1
    wdw = wdw.reshape(3 * C, 1, 5, 5)
    wgrp = torch.zeros(3 * C, 16)
    for grp in range(3 * C // 16):
        perm = torch.randperm(16, generator=g)
        wgrp[grp * 16 + torch.arange(16), perm] = torch.randint(0, 2, (16,), generator=g).float() * 2 - 1
    wsel = torch.zeros(C, 2 * C)
    wsel[torch.arange(C), torch.randperm(2 * C, generator=g)[:C]] = torch.randint(0, 2, (C,), generator=g).float() * 2 - 1
    ref = _mla_lattice_ref(x, wqkv, wdw, wgrp, wsel)
    assert float((ref - x.float()).abs().max()) > 0.05, "degenerate lattice: the attention term vanished"
    return x, wqkv, wdw, wgrp, wsel, ref


@pytest.mark.parametrize("B,Nq,Nk,merged", [(32, 10, 5184, 1), (1, 16, 2001, 0), (3, 1, 1100, 1), (2, 7, 64, 0), (1, 9, 65, 1), (2, 10, 3, 0)])
def test_attention_t2i_mfma_is_a_gather_on_one_hot_scores(B, Nq, Nk, merged):
    """token -> image attention on the matrix cores: key j is the +-9 binary code of j in 13 of the 16 head dimensions, query (i, head)
    is the code of its chosen key -> its score is >= 40.5 above every other key's, the softmax is one-hot in fp32, and the output
    must be row j*(i, head) of V bit for bit, through the key tiling, the running maximum and the fixed-order merge of the partials"""
    heads, hd, D = 8, 16, 128
    q, k, v, ref = t2i_lattice(B, Nq, Nk)
    o_d = torch.full((B, Nq, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    q_d = q.to("cuda", torch.bfloat16)
    if merged:
        kv_d = torch.cat([k, v], dim=-1).to("cuda", torch.bfloat16).contiguous()
        U.check(U.lib().esam3_op_attention(1, U.P(q_d), U.P(kv_d), None, U.P(o_d), B, Nq, Nk, heads, hd, 3, None), "op_attention 3")
    else:
        k_d, v_d = k.to("cuda", torch.bfloat16), v.to("cuda", torch.bfloat16)
        U.check(U.lib().esam3_op_attention(1, U.P(q_d), U.P(k_d), U.P(v_d), U.P(o_d), B, Nq, Nk, heads, hd, 2, None), "op_attention 2")
    _same_integers(o_d.float().cpu(), ref, "t2i one-hot attention")


def t2i_lattice(B, Nq, Nk):
    heads, hd, D = 8, 16, 128
    g = torch.Generator().manual_seed(11 + Nk)
    bits = ((torch.arange(Nk)[:, None] >> torch.arange(13)[None]) & 1).float() * 2 - 1             # [Nk, 13]
    code = torch.zeros(Nk, hd)
    code[:, :13] = bits * 9.0
    k = code[None, :, None, :].expand(B, Nk, heads, hd).reshape(B, Nk, D).contiguous()
    pick = torch.randint(0, Nk, (B, Nq, heads), generator=g)
    q = code[pick].reshape(B, Nq, D).contiguous()
    v = _ints(B, Nk, D, lo=-8, hi=8, seed=5)
    vh = v.reshape(B, Nk, heads, hd)
    ref = torch.stack([torch.stack([torch.cat([vh[b, pick[b, i, h], h] for h in range(heads)]) for i in range(Nq)]) for b in range(B)])
    return q, k, v, ref


@pytest.mark.parametrize("Bp,P,T", [(1, 16, 7), (2, 5184, 10), (3, 144, 16), (5, 80, 1), (37, 48, 9)])
@pytest.mark.parametrize("q_from", ["x", "table"])
def test_i2t_block_on_one_hot_scores(Bp, P, T, q_from):
    """image -> token block: the query of (pixel, head) is 16 e_d (from the pixel's own channels through a +-1 selection Wq when
    q_from = "x", from the position table otherwise, the other path adding noise of magnitude <= 2), key t is 16 e_t: the selected
    token's score is >= 48 above the others and the attention output is its value row exactly; out_proj is a sparse +-1 matrix,
    so the pre-norm row is an exact integer vector.  The closing LayerNorm is a float reduction: every element within one bf16 ulp of
    the fp64 result, >= 99.5 % bit-equal."""
    a = i2t_lattice(Bp, P, T, q_from)
    ref = a["ref"]
    x_d = a["x"].to("cuda", torch.bfloat16).contiguous()
    out = torch.full((Bp, P, 256), float("nan"), dtype=torch.bfloat16, device="cuda")
    U.check(U.lib().esam3_op_i2t_block(U.P(x_d), U.H(U.np32(a["wq"])), U.H(U.np32(a["bq"])), U.H(U.np32(a["peq"])), U.H(U.np32(a["wo"])),
                                       U.H(U.np32(a["bo"])), U.H(U.np32(a["gamma"])), U.H(U.np32(a["beta"])), U.H(U.np32(a["tk"])),
                                       U.H(U.np32(a["tv"])), U.P(out), Bp, P, T, None), "op_i2t_block")
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    refb = ref.to(torch.bfloat16).double()
    ulp = (2.0 ** (torch.floor(torch.log2(ref.abs().clamp_min(2.0 ** -20))) - 7)).clamp_min(1e-5)   # near zero: the fp32 reduction's own noise
    far = (got - ref).abs() > ulp
    assert not far.any(), f"{int(far.sum())} elements beyond one bf16 ulp; worst {float(((got - ref).abs() / ulp).max()):.2f} ulp"
    assert float((got == refb).double().mean()) >= 0.995, float((got == refb).double().mean())


def i2t_lattice(Bp, P, T, q_from):
    heads, hd = 8, 16
    g = torch.Generator().manual_seed(100 * T + P)
    d_sel = torch.randint(0, T, (Bp, P, heads), generator=g)                       # chosen token (= head dimension) per (prompt, pixel, head)
    wq = torch.zeros(128, 256)
    x = torch.zeros(Bp, P, 256)
    peq = torch.zeros(P, 128)
    rows = torch.arange(128)
    if q_from == "x":
        wq[rows, rows] = 1.0                                                       # q[h*16 + d] = x[h*16 + d] + noise from the upper 128 channels
        x.view(Bp, P, 2, heads, hd)[:, :, 0].scatter_(-1, d_sel[..., None], 16.0)
        x[:, :, 128:] = _ints(Bp, P, 128, lo=-1, hi=1, seed=3)
        wq[rows, 128 + torch.randperm(128, generator=g)] = torch.randint(0, 2, (128,), generator=g).float() * 2 - 1   # noise, |.| <= 1
    else:
        d_sel = d_sel[:1].expand(Bp, P, heads).contiguous()                        # the table is shared by the prompts
        peq.view(P, heads, hd).scatter_(-1, d_sel[0][..., None], 16.0)
        x = _ints(Bp, P, 256, lo=-1, hi=1, seed=3)
        wq = _sparse_pm1(128, 256, 2, seed=4)                                      # noise, |.| <= 2
    bq = torch.zeros(128)
    tk = torch.zeros(Bp, T, heads, hd)
    tk[:, torch.arange(T), :, torch.arange(T)] = 16.0
    tk = tk.reshape(Bp, T, 128)
    tv = _ints(Bp, T, 128, lo=-8, hi=8, seed=5)
    wo, bo = _sparse_pm1(256, 128, 2, seed=6), _ints(256, lo=-2, hi=2, seed=7)
    gamma = (_ints(256, lo=2, hi=6, seed=8) / 4.0)
    beta = (_ints(256, lo=-2, hi=2, seed=9) / 4.0)
    tvh = tv.reshape(Bp, T, heads, hd)
    o = torch.gather(tvh[:, None].expand(Bp, P, T, heads, hd), 2, d_sel[:, :, None, :, None].expand(Bp, P, 1, heads, hd))[:, :, 0]
    y = o.reshape(Bp, P, 128).double() @ wo.double().T + bo.double() + x.double()
    ref = F.layer_norm(y, (256,), gamma.double(), beta.double(), 1e-5)
    return dict(x=x, wq=wq, bq=bq, peq=peq, wo=wo, bo=bo, gamma=gamma, beta=beta, tk=tk, tv=tv, ref=ref, y=y)


@pytest.mark.parametrize("M,Cin,Hid,Cout,res", [(1000, 64, 128, 64, True), (4097, 64, 128, 64, True), (31, 64, 128, 64, False)])
def test_fused_mlp_exact_on_the_lattice(M, Cin, Hid, Cout, res):
    """1x1 -> GELU -> 1x1 (+ shortcut) with hidden pre-activations in {.., -12, -6, 0, 6, 12, ..}: GELU(6 k) is 6 k (k > 0) or 0 to
    within 1e-8, which the bf16 rounding of the hidden tensor and the integer sums of the second layer absorb"""
    x = _ints(M, Cin, lo=-1, hi=1, seed=1, mult=6)
    w1, b1 = _sparse_pm1(Hid, Cin, 2, seed=2), _ints(Hid, lo=-1, hi=1, seed=3, mult=6)
    w2, b2 = _sparse_pm1(Cout, Hid, 3, seed=4), _ints(Cout, lo=-2, hi=2, seed=5)
    r = _ints(M, Cout, lo=-4, hi=4, seed=6) if res else None
    h = F.relu(x.double() @ w1.double().T + b1.double())          # GELU on the lattice = ReLU (|error| < 1e-8, gone after bf16 rounding)
    ref = h @ w2.double().T + b2.double()
    if res:
        ref = ref + r.double()
    assert _bf16_exact(ref)
    x_d = x.to("cuda", torch.bfloat16)
    r_d = r.to("cuda", torch.bfloat16) if res else None
    out = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    U.check(U.lib().esam3_op_fused_mlp(U.P(x_d), U.H(U.np32(w1)), U.H(U.np32(b1)), U.H(U.np32(w2)), U.H(U.np32(b2)), U.P(r_d), U.P(out),
                                       M, Cin, Hid, Cout, U.ACT["gelu"], None), "op_fused_mlp")
    _same_integers(out.float().cpu(), ref, "fused_mlp")


@pytest.mark.parametrize("rows,P", [(32, 32), (5184 * 2, 5184), (64 * 37, 64)])
def test_rowlin256_exact_on_the_lattice(rows, P):
    """the merged k | v projection (weights resident in LDS, transposed-MFMA C layout, position table) on integers"""
    x = _ints(rows, 256, lo=-3, hi=3, seed=1)
    w = _sparse_pm1(256, 256, 4, seed=2)
    b = _ints(256, lo=-4, hi=4, seed=3)
    tb = _ints(P, 256, lo=-8, hi=8, seed=4)
    ref = x.double() @ w.double().T + b.double() + tb.double().repeat(rows // P, 1)
    x_d = x.to("cuda", torch.bfloat16)
    out = torch.full((rows, 256), float("nan"), dtype=torch.bfloat16, device="cuda")
    U.check(U.lib().esam3_op_rowlin256(U.P(x_d), U.H(U.np32(w)), U.H(U.np32(b)), U.H(U.np32(tb)), P, U.P(out), rows, None), "op_rowlin256")
    assert torch.equal(out.float().cpu(), ref.float())


# ---- neck level 0: the composed up-conv (the dominant launch of the headline step and its SAM2-side narrow twin) ---------------------------
def upconv_lattice(B, H, W, Cin, C1, Cmid, Cout, nnz3, seed=0):
    """necks.py:74-98 level 0 behind the GELU: dconv_2x2_1 (ConvTranspose2d k2 s2, Cin -> C1) -> conv_1x1 (C1 -> Cmid) -> conv_3x3
    (Cmid -> Cout, pad 1) on integers.  Returns the layer weights, the ConvT with the 1x1 folded in (what the operator entry takes; integer
    arithmetic, exact) and the fp64 result of running the three layers one after the other."""
    x = _ints(B, Cin, H, W, lo=-2, hi=2, seed=seed + 1)
    wt_raw = _sparse_pm1(C1 * 4, Cin, 2, seed=seed + 2).reshape(C1, 4, Cin).permute(2, 0, 1).reshape(Cin, C1, 2, 2).contiguous()
    bt_raw = _ints(C1, lo=-1, hi=1, seed=seed + 3)
    w1 = _sparse_pm1(Cmid, C1, 2, seed=seed + 4)
    b1 = _ints(Cmid, lo=-1, hi=1, seed=seed + 5)
    w3 = _sparse_pm1(Cout, Cmid * 9, nnz3, seed=seed + 6).reshape(Cout, Cmid, 3, 3).contiguous()
    b3 = _ints(Cout, lo=-2, hi=2, seed=seed + 7)
    mid = F.conv_transpose2d(x.double(), wt_raw.double(), bt_raw.double(), stride=2)
    mid = F.conv2d(mid, w1.double()[:, :, None, None], b1.double())
    ref = F.conv2d(mid, w3.double(), b3.double(), padding=1)
    assert float(ref.abs().max()) < 2 ** 15 and float(ref.abs().max()) > 8  # fp32 sums of integers: exact in any order
    # ConvT o 1x1 (engine.hip: compose_convT_1x1): wt[ci][m][t] = sum_c raw[ci][c][t] * w1[m][c], bt = w1 bt_raw + b1
    wt = torch.einsum("ictu,mc->imtu", wt_raw.double(), w1.double()).contiguous()
    bt = w1.double() @ bt_raw.double() + b1.double()
    assert _bf16_exact(wt)
    return x, wt.float(), bt.float(), w3, b3, ref


def _run_upconv(x, wt, bt, w3, b3, Cout, narrow):
    B, Cin, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous().to("cuda", torch.bfloat16)  # [B][H+2][W+2][Cin], zero border
    out = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    U.check(U.lib().esam3_op_upconv(U.P(xp), U.H(U.np32(wt)), U.H(U.np32(bt)), U.H(U.np32(w3)), U.H(U.np32(b3)), U.P(out),
                                    B, H, W, Cin, wt.shape[1], Cout, int(narrow), None), "op_upconv")
    return U.from_dev_nhwc(out)


def _assert_upconv_exact(got, ref, what):
    exp = ref.to(torch.bfloat16).float()  # the accumulators hold the exact integer: ONE rounding, on the store
    bad = got != exp
    if bad.any():
        i = tuple(bad.nonzero()[0].tolist())
        H2, W2 = ref.shape[-2:]
        ring = bad[..., 0, :].sum() + bad[..., H2 - 1, :].sum() + bad[..., :, 0].sum() + bad[..., :, W2 - 1].sum()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} elements differ on exact data ({int(ring)} of them on the ring of the "
                             f"output image); first at (b, c, y, x) = {i}: got {float(got[i])!r}, expected {float(exp[i])!r}")


@pytest.mark.parametrize("B,H,W,Cin,C1,Cmid,Cout", [
    (3, 16, 16, 64, 64, 64, 256),       # one pixel tile per image: every pixel tile is an edge tile in both directions
    (2, 32, 48, 128, 64, 128, 256),     # B > 1, H != W, interior + edge tiles
    (1, 48, 32, 192, 128, 64, 512),     # two N tiles per parity class (class = n0 / Cout), 3 channel chunks
    (1, 144, 144, 512, 256, 256, 256),  # the real shape of the dominant launch (one image of the headline batch)
    (5, 64, 64, 64, 64, 64, 256),       # 80 pixel tiles x 4 classes > 256 workgroups: the persistent loop hands tiles over
])
def test_upconv_gather_exact_on_the_lattice(B, H, W, Cin, C1, Cmid, Cout):
    """gemm256p's up-conv gather (ksize 2): the four parity classes x 2x2 taps of the composed weight, the class shift of the gather on
    the zero-bordered input, the channel-chunk-major K order, the ConvT pixel-shuffle store with tap = class and the ring-pixel bias
    correction -- all of it bit for bit against ConvT -> 1x1 -> 3x3 run layer by layer in fp64"""
    x, wt, bt, w3, b3, ref = upconv_lattice(B, H, W, Cin, C1, Cmid, Cout, nnz3=9)
    got = _run_upconv(x, wt, bt, w3, b3, Cout, narrow=False)
    _assert_upconv_exact(got, ref, "upconv_gather")


@pytest.mark.parametrize("B,H,W,Cin,C1,Cmid", [
    (3, 16, 16, 64, 64, 64), (2, 32, 48, 128, 64, 128), (1, 48, 32, 48, 32, 64), (1, 144, 144, 512, 256, 256), (9, 96, 96, 32, 32, 32),
])
def test_upconv_narrow_exact_on_the_lattice(B, H, W, Cin, C1, Cmid):
    """upconv_narrow_kernel (32 output channels per parity class = conv_3x3 o conv_s0 of the SAM2 side, mask_decoder.py): the 18x18
    halo per 16-channel chunk, the 16 (halo shift, class) pairs, the four accumulator sets and the 2x2 output block store.  w3 here
    stands for the composed 3x3 o conv_s0 (the same integer lattice; its folding is test_unfused_layer_list_matches_golden's business)"""
    x, wt, bt, w3, b3, ref = upconv_lattice(B, H, W, Cin, C1, Cmid, 32, nnz3=24, seed=100)
    got = _run_upconv(x, wt, bt, w3, b3, 32, narrow=True)
    _assert_upconv_exact(got, ref, "upconv_narrow")


def test_upconv_gather_bias_only_ring():
    """zero input: the output is the composed bias, and on the ring of the output image the shares of the 3x3 taps that fall outside
    are taken back (GemmParams::border_corr) -- corners included, per parity class"""
    B, H, W, Cin, C1, Cmid, Cout = 1, 32, 32, 64, 64, 64, 256
    x, wt, bt, w3, b3, _ = upconv_lattice(B, H, W, Cin, C1, Cmid, Cout, nnz3=9, seed=7)
    x = torch.zeros_like(x)
    bt = _ints(Cmid, lo=-3, hi=3, seed=11)
    mid = bt.double()[None, :, None, None].expand(B, Cmid, 2 * H, 2 * W)
    ref = F.conv2d(mid, w3.double(), b3.double(), padding=1)
    assert not torch.equal(ref[..., 0, :], ref[..., 1, :])  # the ring differs from the interior: the correction is exercised
    got = _run_upconv(x, wt, bt, w3, b3, Cout, narrow=False)
    _assert_upconv_exact(got, ref, "upconv_gather bias ring")
