"""GPU: the kernels and layers the RepViT students need in training mode against torch.autograd --
``esam3_channel_scale`` / ``esam3_batched_coldot`` / the sigmoid activation / the dense stride-2 3x3 of the patch embedding, then RepVGGDW,
SqueezeExcite and whole RepViT blocks (sam3/backbones/repvit.py:27-36,84-93,125-161; timm SqueezeExcite) forwards and backwards on the
HIP kernels with DEVICE-resident fp32 parameters, as ``Stage1Trainer`` holds them.  torch on the CPU in fp32 is the reference (the layers
are torch's own); bf16 runs see bf16-quantised inputs and are held to the reference block's own bf16-autocast distance."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_train_blocks import TDT, _close, _inside_autocast_yardstick, _rel_l2
from tests.test_train_blocks_host import _bn, _to_nchw, _to_nhwc
from tests.test_train_repvit_host import _rand_params, _repvggdw_ref, _se_ref

pytestmark = pytest.mark.gpu


def _dev(p):
    return {k: v.float().cuda().contiguous() for k, v in p.items()}


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,HW,C", [(2, 35, 48), (3, 1, 24), (1, 4097, 640), (8, 63, 16)])
def test_channel_scale(mode, B, HW, C):
    from efficientsam3_amd import train_blocks as tb
    g = torch.Generator().manual_seed(B * 1000 + HW + C)
    x, add = torch.randn(B, HW, C, generator=g).to(TDT[mode]), torch.randn(B, HW, C, generator=g).to(TDT[mode])
    m1, b1 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    m2, b2 = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g)
    xf, af = x.float(), add.float()
    cases = [("per channel, +1, bias, add", dict(mul=m1, bias=b1, add=add, plus_one=True), af + xf * (m1 + 1) + b1),
             ("per channel, +1, add", dict(mul=m1, add=add, plus_one=True), af + xf * (m1 + 1)),
             ("per image", dict(mul=m2), xf * m2[:, None, :]),
             ("per image, per-image bias scaled", dict(mul=m2, bias=b2, bias_scale=0.125), xf * m2[:, None, :] + b2[:, None, :] * 0.125)]
    for what, kw, ref in cases:
        dkw = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
        got = tb.channel_scale(x.cuda(), **dkw)
        _close(got, ref, mode, f"channel_scale {what}", f32=2e-6, bf16=8e-3)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,HW,C", [(2, 35, 48), (3, 1, 24), (2, 63504, 48), (1, 4097, 1280), (8, 1024, 384), (32, 3969, 64), (2, 7, 2048)])
def test_batched_coldot(mode, B, HW, C):
    from efficientsam3_amd import train_blocks as tb
    g = torch.Generator().manual_seed(B * 1000 + HW + C)
    a, b2 = torch.randn(B, HW, C, generator=g).to(TDT[mode]), torch.randn(B, HW, C, generator=g).to(TDT[mode])
    ad, bd = a.cuda(), b2.cuda()
    mean = tb.batched_coldot(ad, None, scale=1.0 / HW)
    dot = tb.batched_coldot(ad, bd)
    flat = tb.batched_coldot(ad, bd, per_image=False)
    assert mean.shape == (B, C) and dot.shape == (B, C) and flat.shape == (C,) and mean.dtype == torch.float32
    # the kernel accumulates in fp32 in a fixed order: compare with a float64 sum of the same (possibly bf16-rounded) inputs
    a64, b64 = a.double(), b2.double()
    scale = float((a64 * b64).abs().sum(1).max())
    assert float((mean.cpu().double() - a64.mean(1)).abs().max()) <= 1e-5 * max(1.0, float(a64.abs().mean(1).max())) + 2e-6
    assert float((dot.cpu().double() - (a64 * b64).sum(1)).abs().max()) <= 2e-6 * scale
    assert float((flat.cpu().double() - (a64 * b64).sum((0, 1))).abs().max()) <= 2e-6 * scale * B
    assert torch.equal(tb.batched_coldot(ad, bd), dot)          # fixed summation order: bit-identical between calls


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_sigmoid_forward_backward(mode):
    from efficientsam3_amd import train_blocks as tb
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(6, 40, generator=g) * 4.0).to(TDT[mode])
    x.view(-1)[:4] = torch.tensor([0.0, -30.0, 30.0, 1e-3]).to(TDT[mode])
    dy = torch.randn(6, 40, generator=g).to(TDT[mode])
    xr = x.float().clone().requires_grad_(True)
    yr = torch.sigmoid(xr)
    yr.backward(dy.float())
    _close(tb.act_forward(x.cuda(), "sigmoid"), yr.detach(), mode, "sigmoid y", f32=2e-6, bf16=5e-3)
    _close(tb.act_backward(x.cuda(), dy.cuda(), "sigmoid"), xr.grad, mode, "sigmoid dx", f32=2e-6, bf16=5e-3)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 12, 10, 24, 48), (1, 9, 7, 32, 64), (2, 16, 16, 40, 80)])
def test_patch_embed_conv3x3_stride2_vs_autograd(mode, B, H, W, Cin, Cout):
    """Conv2d_BN(Cin, Cout, 3, 2, 1) of the patch embedding (repvit.py:230): the strided implicit GEMM, its data gradient through the
    zero-spread dy, its weight gradient tap by tap, the BatchNorm in training mode"""
    from efficientsam3_amd import train_repvit as tr
    g = torch.Generator().manual_seed(Cin)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    x = torch.randn(B, H, W, Cin, generator=g).to(TDT[mode])
    dy = torch.randn(B, (H + 1) // 2, (W + 1) // 2, Cout, generator=g).to(TDT[mode])

    def reference(amp):
        wr, gr, br = (t.clone().requires_grad_(True) for t in (w, gamma, beta))
        xr = _to_nchw(x.float()).requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            yr = _bn(F.conv2d(xr, wr, None, stride=2, padding=1), gr, br)
        yr.float().backward(_to_nchw(dy.float()))
        return {"y": _to_nhwc(yr.detach().float()), "dx": _to_nhwc(xr.grad), "weight": wr.grad, "gamma": gr.grad, "beta": br.grad}

    r32 = reference(False)
    layer = tr.Conv3x3S2Train(w.cuda(), gamma.cuda(), beta.cuda())
    y = layer.forward(x.cuda().contiguous())
    dx, grads = layer.backward(dy.cuda().contiguous())
    pairs = [(y, r32["y"], "y"), (dx, r32["dx"], "dx"), (grads["weight"], r32["weight"], "weight"), (grads["gamma"], r32["gamma"], "gamma"),
             (grads["beta"], r32["beta"], "beta")]
    if mode == "f32":
        for got, ref, what in pairs:
            _close(got, ref, mode, what, 3e-4)
    else:
        _inside_autocast_yardstick(pairs, reference(True), f"conv3x3 s2 {Cin}->{Cout}")


def _block_pairs(grads, ref_grads, typical_floor=1e-2):
    """(got, ref, name) for every parameter gradient; gradients that are zero up to rounding (a constant shift in front of a BatchNorm) are
    returned separately with the scale they are measured on"""
    typical = float(torch.stack([v.abs().max() for v in ref_grads.values()]).median())
    pairs, zeros = [], []
    for k, ref in ref_grads.items():
        (zeros if float(ref.abs().max()) < typical_floor * typical else pairs).append((grads[k].reshape(ref.shape), ref, k))
    return pairs, zeros, typical


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,C", [(2, 12, 10, 48), (1, 9, 9, 64), (2, 7, 5, 80)])
def test_repvggdw_forward_backward_vs_autograd(mode, B, H, W, C):
    from efficientsam3_amd import train_repvit as tr
    p = _rand_params({"conv.weight": (C, 1, 3, 3), "conv.gamma": (C,), "conv.beta": (C,), "conv1.weight": (C, 1, 1, 1), "conv1.bias": (C,),
                      "bn.gamma": (C,), "bn.beta": (C,)}, C)
    g = torch.Generator().manual_seed(2)
    x, dy = torch.randn(B, H, W, C, generator=g).to(TDT[mode]), torch.randn(B, H, W, C, generator=g).to(TDT[mode])

    def reference(amp):
        rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        xr = _to_nchw(x.float()).requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            yr = _repvggdw_ref(xr, rp)
        yr.float().backward(_to_nchw(dy.float()))
        out = {k: v.grad for k, v in rp.items()}
        out.update(y=_to_nhwc(yr.detach().float()), dx=_to_nhwc(xr.grad))
        return out

    r32 = reference(False)
    blk = tr.RepVGGDWTrain(_dev(p))
    y = blk.forward(x.cuda().contiguous())
    dx, grads = blk.backward(dy.cuda().contiguous())
    pairs, zeros, typical = _block_pairs(grads, {k: r32[k] for k in p})
    pairs = [(y, r32["y"], "y"), (dx, r32["dx"], "dx")] + pairs
    assert {k for _, _, k in zeros} == {"conv.beta", "conv1.bias"}
    for got, ref, k in zeros:
        assert float(got.abs().max()) <= (1e-4 if mode == "f32" else 3e-2) * typical, (k, float(got.abs().max()), typical)
    if mode == "f32":
        for got, ref, what in pairs:
            _close(got, ref, mode, what, 3e-4)
    else:
        _inside_autocast_yardstick(pairs, reference(True), f"RepVGGDW C={C}")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,C,R", [(2, 12, 10, 48, 16), (8, 9, 9, 64, 16), (1, 32, 32, 384, 96), (3, 5, 4, 80, 24)])
def test_squeeze_excite_forward_backward_vs_autograd(mode, B, H, W, C, R):
    """timm SqueezeExcite: the pooled mean, the two tiny fp32 GEMMs on [B, C] rows (B = 1 .. 8), the sigmoid gate, and all of it backwards"""
    from efficientsam3_amd import train_repvit as tr
    p = _rand_params({"fc1.weight": (R, C, 1, 1), "fc1.bias": (R,), "fc2.weight": (C, R, 1, 1), "fc2.bias": (C,)}, C + R)
    g = torch.Generator().manual_seed(4)
    x, dy = torch.randn(B, H, W, C, generator=g).to(TDT[mode]), torch.randn(B, H, W, C, generator=g).to(TDT[mode])

    def reference(amp):
        rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        xr = _to_nchw(x.float()).requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            yr = _se_ref(xr, rp)
        yr.float().backward(_to_nchw(dy.float()))
        out = {k: v.grad for k, v in rp.items()}
        out.update(y=_to_nhwc(yr.detach().float()), dx=_to_nhwc(xr.grad))
        return out

    r32 = reference(False)
    blk = tr.SqueezeExciteTrain(_dev(p))
    y = blk.forward(x.cuda().contiguous())
    dx, grads = blk.backward(dy.cuda().contiguous())
    pairs = [(y, r32["y"], "y"), (dx, r32["dx"], "dx")] + [(grads[k].reshape(r32[k].shape), r32[k], k) for k in p]
    if mode == "f32":
        for got, ref, what in pairs:
            _close(got, ref, mode, what, 3e-4)
    else:
        _inside_autocast_yardstick(pairs, reference(True), f"SqueezeExcite C={C} B={B}")


def _block_sd(C, Cout, stride, use_se, seed):
    """the parameters of one RepViTBlock under features.1.* in the reference's names (schema.repvit_schema's layout)"""
    shapes = {}
    q = "features.1."

    def conv_bn(name, cin, cout, k, groups=1):
        shapes[name + ".c.weight"] = (cout, cin // groups, k, k)
        shapes[name + ".bn.weight"], shapes[name + ".bn.bias"] = (cout,), (cout,)

    if stride == 2:
        conv_bn(q + "token_mixer.0", C, C, 3, groups=C)
        conv_bn(q + "token_mixer.2", C, Cout, 1)
    else:
        conv_bn(q + "token_mixer.0.conv", C, C, 3, groups=C)
        shapes[q + "token_mixer.0.conv1.weight"], shapes[q + "token_mixer.0.conv1.bias"] = (C, 1, 1, 1), (C,)
        shapes[q + "token_mixer.0.bn.weight"], shapes[q + "token_mixer.0.bn.bias"] = (C,), (C,)
        if use_se:
            R = max(8, int(C * 0.25 + 4) // 8 * 8)
            shapes.update({q + "token_mixer.1.fc1.weight": (R, C, 1, 1), q + "token_mixer.1.fc1.bias": (R,),
                           q + "token_mixer.1.fc2.weight": (C, R, 1, 1), q + "token_mixer.1.fc2.bias": (C,)})
    conv_bn(q + "channel_mixer.m.0", Cout, 2 * Cout, 1)
    conv_bn(q + "channel_mixer.m.2", 2 * Cout, Cout, 1)
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in shapes.items():
        if k.endswith("bn.weight"):
            out[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith(".c.weight") and shp[2] == 1:
            out[k] = torch.randn(shp, generator=g) * shp[1] ** -0.5
        else:
            out[k] = torch.randn(shp, generator=g) * 0.3
    return out


def _block_ref(x, sd, stride, use_se):
    q = "features.1."

    def conv_bn(t, base, stride=1, pad=0, groups=1):
        return _bn(F.conv2d(t, sd[base + ".c.weight"], None, stride=stride, padding=pad, groups=groups), sd[base + ".bn.weight"], sd[base + ".bn.bias"])

    c = x.shape[1]
    if stride == 2:
        t = conv_bn(conv_bn(x, q + "token_mixer.0", 2, 1, groups=c), q + "token_mixer.2")
    else:
        t = _repvggdw_ref(x, {"conv.weight": sd[q + "token_mixer.0.conv.c.weight"], "conv.gamma": sd[q + "token_mixer.0.conv.bn.weight"],
                              "conv.beta": sd[q + "token_mixer.0.conv.bn.bias"], "conv1.weight": sd[q + "token_mixer.0.conv1.weight"],
                              "conv1.bias": sd[q + "token_mixer.0.conv1.bias"], "bn.gamma": sd[q + "token_mixer.0.bn.weight"],
                              "bn.beta": sd[q + "token_mixer.0.bn.bias"]})
        if use_se:
            t = _se_ref(t, {n: sd[f"{q}token_mixer.1.{n}"] for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")})
    return t + conv_bn(F.gelu(conv_bn(t, q + "channel_mixer.m.0")), q + "channel_mixer.m.2")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,C,Cout,stride,use_se", [(2, 12, 10, 48, 48, 1, True), (2, 13, 10, 48, 96, 2, False), (1, 9, 9, 64, 64, 1, False),
                                                        (2, 8, 8, 160, 320, 2, False), (2, 6, 7, 384, 384, 1, True)])
def test_repvit_block_forward_backward_vs_autograd(mode, B, H, W, C, Cout, stride, use_se):
    """One RepViTBlock (repvit.py:125-161) in TRAINING mode, forwards and backwards on the HIP kernels from a state dict in the reference's
    names: output, input gradient, every parameter gradient under its state-dict name, the running statistics of every BatchNorm."""
    from efficientsam3_amd import train_repvit as tr
    sd = _block_sd(C, Cout, stride, use_se, seed=C + stride)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, C, generator=g).to(TDT[mode])
    dy = torch.randn(B, (H + stride - 1) // stride, (W + stride - 1) // stride, Cout, generator=g).to(TDT[mode])

    def reference(amp):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = _to_nchw(x.float()).requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            yr = _block_ref(xr, leaves, stride, use_se)
        yr.float().backward(_to_nchw(dy.float()))
        out = {k: v.grad for k, v in leaves.items()}
        out.update(y=_to_nhwc(yr.detach().float()), dx=_to_nhwc(xr.grad))
        return out

    r32 = reference(False)
    dsd = _dev(sd)
    blk = tr.RepViTBlockTrain(lambda k: dsd[k], lambda k: k in dsd, "features.1", stride, use_se)
    y = blk.forward(x.cuda().contiguous())
    dx, grads = blk.backward(dy.cuda().contiguous())
    assert sorted(grads) == sorted(sd)
    pairs, zeros, typical = _block_pairs(grads, {k: r32[k] for k in sd})
    pairs = [(y, r32["y"], "y"), (dx, r32["dx"], "dx")] + pairs
    for got, ref, k in zeros:
        assert float(got.abs().max()) <= (2e-4 if mode == "f32" else 5e-2) * typical, (k, float(got.abs().max()), typical)
    if mode == "f32":
        worst = max((_rel_l2(got, ref), what) for got, ref, what in pairs)
        print(f"[RepViT block C={C}->{Cout} s{stride} se={use_se} f32] worst relative L2 error {worst[0]:.2e} ({worst[1]})")
        for got, ref, what in pairs:
            _close(got, ref, mode, what, 5e-4)
    else:
        _inside_autocast_yardstick(pairs, reference(True), f"RepViT block {C}->{Cout} s{stride} se={use_se}")
    # running statistics of every BatchNorm of the block: one training-mode forward from (0, 1)
    names = dict(blk.norm_layers())
    assert set(names) == {k[:-len(".weight")] for k in sd if k.endswith("bn.weight")}


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,IH,IW,Cin,Cout,stride", [(2, 9, 7, 16, 24, 1), (1, 32, 32, 64, 64, 1), (8, 32, 32, 256, 128, 1), (2, 12, 10, 24, 48, 2),
                                                      (1, 13, 11, 40, 80, 2), (3, 1, 5, 8, 8, 1), (2, 64, 64, 32, 64, 2)])
def test_conv3x3_weight_gradient_one_launch(mode, B, IH, IW, Cin, Cout, stride):
    """``esam3_conv3x3_wgrad``: the nine taps of a dense 3x3 conv's weight gradient in one gathered launch (padding 1, stride 1 | 2, odd sizes,
    images smaller than a 64-row tile, several row splits) against autograd"""
    from efficientsam3_amd import stage1_train as st
    g = torch.Generator().manual_seed(B + IH + Cin)
    x = torch.randn(B, IH, IW, Cin, generator=g).to(TDT[mode])
    dy = torch.randn(B, (IH + stride - 1) // stride, (IW + stride - 1) // stride, Cout, generator=g).to(TDT[mode])
    wr = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(_to_nchw(x.float()), wr, None, stride=stride, padding=1).backward(_to_nchw(dy.float()))
    dw = st.conv3x3_wgrad(dy.cuda().contiguous(), x.cuda().contiguous(), stride)
    assert dw.shape == wr.grad.shape and dw.dtype == torch.float32
    _close(dw, wr.grad, "f32", f"conv3x3 wgrad s{stride}", f32=2e-5 if mode == "f32" else 2e-5)
    assert torch.equal(st.conv3x3_wgrad(dy.cuda().contiguous(), x.cuda().contiguous(), stride), dw)      # fixed summation order
