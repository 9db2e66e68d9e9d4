"""CPU: the COMPOSITION logic of efficientsam3_amd/train_repvit.py (RepVGGDW, SqueezeExcite, the RepViT block in its stride-1 and stride-2
forms, the dense stride-2 3x3 of the patch embedding, the whole trunk with its state-dict names) with every kernel wrapper replaced by a
plain torch stand-in of the same contract, against torch.autograd of the same layers written with torch functions
(sam3/backbones/repvit.py:27-36,84-93,125-161,226-238; timm SqueezeExcite).  The kernels are checked on the GPU by
tests/test_train_blocks.py, a whole training step against the reference's own run by tests/test_stage1_step.py."""
import pytest
import torch
import torch.nn.functional as F

from efficientsam3_amd import schema
from efficientsam3_amd import train_blocks as tb
from efficientsam3_amd import train_repvit as tr
from tests.test_train_blocks_host import _bn, _check, _to_nchw, _to_nhwc, cpu_kernels  # noqa: F401


def install_repvit_kernels(monkeypatch):  # noqa: F811
    def conv_nhwc(x, w, stride):
        return _to_nhwc(F.conv2d(_to_nchw(x), w, None, stride=stride, padding=1))

    def conv3x3(x, w, out_channels, dgrad):
        assert dgrad     # the data gradient of a stride-1 3x3 conv with weight w [Cin_of_x ... ] = conv_transpose
        return _to_nhwc(F.conv_transpose2d(_to_nchw(x), w, None, stride=1, padding=1))

    def conv_s2_wgrad(dy, x):
        wr = torch.zeros((dy.shape[-1], x.shape[-1], 3, 3), requires_grad=True)
        F.conv2d(_to_nchw(x), wr, None, stride=2, padding=1).backward(_to_nchw(dy))
        return wr.grad

    monkeypatch.setattr(tr, "conv3x3_s2_wgrad", conv_s2_wgrad)
    monkeypatch.setattr(tr, "conv3x3_s2_forward", lambda x, w: conv_nhwc(x, w, 2))
    monkeypatch.setattr(tr, "_conv3x3", conv3x3)
    monkeypatch.setattr(tb, "stem_forward", lambda img, w, dtype: _to_nhwc(F.conv2d(img, w, None, stride=2, padding=1)).to(dtype))


@pytest.fixture
def repvit_kernels(cpu_kernels, monkeypatch):  # noqa: F811
    install_repvit_kernels(monkeypatch)


def _rand_params(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in shapes.items():
        if k.endswith(("gamma", "bn.weight")):
            out[k] = torch.rand(shp, generator=g) + 0.5
        else:
            out[k] = torch.randn(shp, generator=g) * 0.3
    return out


def _repvggdw_ref(x, p):
    c = x.shape[1]
    a = _bn(F.conv2d(x, p["conv.weight"], None, padding=1, groups=c), p["conv.gamma"], p["conv.beta"])
    return _bn(a + F.conv2d(x, p["conv1.weight"], p["conv1.bias"], groups=c) + x, p["bn.gamma"], p["bn.beta"])


def _se_ref(x, p):
    g = x.mean((2, 3), keepdim=True)
    g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(g, p["fc1.weight"], p["fc1.bias"])), p["fc2.weight"], p["fc2.bias"]))
    return x * g


def test_repvggdw_composition(repvit_kernels):
    B, H, W, C = 2, 7, 6, 16
    p = _rand_params({"conv.weight": (C, 1, 3, 3), "conv.gamma": (C,), "conv.beta": (C,), "conv1.weight": (C, 1, 1, 1), "conv1.bias": (C,),
                      "bn.gamma": (C,), "bn.beta": (C,)}, 1)
    g = torch.Generator().manual_seed(2)
    x, dy = torch.randn(B, H, W, C, generator=g), torch.randn(B, H, W, C, generator=g)
    rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = _to_nchw(x).requires_grad_(True)
    yr = _repvggdw_ref(xr, rp)
    yr.backward(_to_nchw(dy))
    blk = tr.RepVGGDWTrain(p)
    y = blk.forward(x)
    dx, grads = blk.backward(dy)
    # conv.beta and conv1.bias shift the input of a BatchNorm by a constant: their gradient is zero up to rounding (1e-6 here against
    # gradients of order 1), so they are compared on the scale of the other gradients
    zero = ("conv.beta", "conv1.bias")
    _check([(y, _to_nhwc(yr.detach()), "y"), (dx, _to_nhwc(xr.grad), "dx")] + [(grads[k].reshape(rp[k].shape), rp[k].grad, k) for k in p if k not in zero])
    for k in zero:
        assert float(grads[k].abs().max()) <= 1e-4 and float(rp[k].grad.abs().max()) <= 1e-4, k


def test_squeeze_excite_composition(repvit_kernels):
    B, H, W, C, R = 3, 5, 4, 16, 8
    p = _rand_params({"fc1.weight": (R, C, 1, 1), "fc1.bias": (R,), "fc2.weight": (C, R, 1, 1), "fc2.bias": (C,)}, 3)
    g = torch.Generator().manual_seed(4)
    x, dy = torch.randn(B, H, W, C, generator=g), torch.randn(B, H, W, C, generator=g)
    rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = _to_nchw(x).requires_grad_(True)
    yr = _se_ref(xr, rp)
    yr.backward(_to_nchw(dy))
    blk = tr.SqueezeExciteTrain(p)
    y = blk.forward(x)
    dx, grads = blk.backward(dy)
    _check([(y, _to_nhwc(yr.detach()), "y"), (dx, _to_nhwc(xr.grad), "dx")] + [(grads[k].reshape(rp[k].shape), rp[k].grad, k) for k in p])


def test_conv3x3_stride2_composition(repvit_kernels):
    """the data gradient through the zero-spread dy (the weight gradient is one kernel: checked on the GPU), odd and even image sizes"""
    for H, W in ((8, 6), (7, 9)):
        B, Cin, Cout = 2, 8, 16
        g = torch.Generator().manual_seed(5)
        w, gamma, beta = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2, torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
        x = torch.randn(B, H, W, Cin, generator=g)
        dy = torch.randn(B, (H + 1) // 2, (W + 1) // 2, Cout, generator=g)
        wr, gr, br = (t.clone().requires_grad_(True) for t in (w, gamma, beta))
        xr = _to_nchw(x).requires_grad_(True)
        yr = _bn(F.conv2d(xr, wr, None, stride=2, padding=1), gr, br)
        yr.backward(_to_nchw(dy))
        layer = tr.Conv3x3S2Train(w, gamma, beta)
        y = layer.forward(x)
        dx, grads = layer.backward(dy)
        _check([(y, _to_nhwc(yr.detach()), "y"), (dx, _to_nhwc(xr.grad), "dx"), (grads["weight"], wr.grad, "w"), (grads["gamma"], gr.grad, "gamma"),
                (grads["beta"], br.grad, "beta")])


def _trunk_ref(img, sd, cfgs):
    """RepViT.features (repvit.py:226-238) with torch functions on a dict of leaf tensors in the reference's names"""
    def conv_bn(x, base, stride=1, pad=0, groups=1):
        return _bn(F.conv2d(x, sd[base + ".c.weight"], None, stride=stride, padding=pad, groups=groups), sd[base + ".bn.weight"], sd[base + ".bn.bias"])

    x = conv_bn(F.gelu(conv_bn(img, "features.0.0", 2, 1)), "features.0.2", 2, 1)
    for i, (_k, _t, _c, use_se, _hs, stride) in enumerate(cfgs, start=1):
        q = f"features.{i}"
        c = x.shape[1]
        if stride == 2:
            x = conv_bn(x, q + ".token_mixer.0", 2, 1, groups=c)
            x = conv_bn(x, q + ".token_mixer.2")
        else:
            t = q + ".token_mixer.0"
            a = conv_bn(x, t + ".conv", 1, 1, groups=c)
            x = _bn(a + F.conv2d(x, sd[t + ".conv1.weight"], sd[t + ".conv1.bias"], groups=c) + x, sd[t + ".bn.weight"], sd[t + ".bn.bias"])
            if use_se:
                x = _se_ref(x, {n: sd[f"{q}.token_mixer.1.{n}"] for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")})
        x = x + conv_bn(F.gelu(conv_bn(x, q + ".channel_mixer.m.0")), q + ".channel_mixer.m.2")
    return x


def test_repvit_trunk_composition_and_names(repvit_kernels):
    """the whole m0.9 trunk at a small image: every parameter of the reference's state dict gets a gradient under its own name and shape, equal
    to autograd's"""
    full = schema.synthetic_state_dict("repvit", "m0.9", seed=3)
    pre = "backbone.vision_backbone.trunk.model.backbone.model."
    sd = {k[len(pre):]: v.float() for k, v in full.items() if k.startswith(pre)}
    params = {k: v for k, v in sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    g = torch.Generator().manual_seed(6)
    img = torch.randn(2, 3, 96, 80, generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yr = _trunk_ref(img, leaves, schema.REPVIT_CFG["m0.9"])
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    trunk = tr.RepViTTrunkTrain({k: v.clone() for k, v in sd.items()}, "m0.9", dtype=torch.float32)
    y = trunk.forward(img)
    order = []
    grads = trunk.backward(_to_nhwc(dy), sink=lambda n, gv: order.append(n))
    assert sorted(grads) == sorted(params) and sorted(order) == sorted(params)
    assert order[0].startswith("features.26.") and order[-1].startswith("features.0.0.")          # last layer first
    d, m = float((y - _to_nhwc(yr.detach())).abs().max()), float(yr.detach().abs().max())
    assert d <= 1e-3 * m, (d, m)
    # gradients that are zero up to rounding (a constant shift in front of a BatchNorm: conv1.bias, conv.bn.bias of every RepVGGDW) are
    # measured on the scale of the typical gradient instead of their own
    typical = float(torch.stack([leaves[k].grad.abs().max() for k in params]).median())
    worst = max((float((grads[k] - leaves[k].grad).abs().max()) / max(float(leaves[k].grad.abs().max()), 1e-2 * typical), k) for k in params)
    print("worst relative gradient error", worst, "typical gradient", typical)
    assert worst[0] <= 5e-3, worst
    for k in params:
        assert tuple(grads[k].shape) == tuple(params[k].shape), k
    # running statistics: every BatchNorm of the state dict is reachable under its name and was updated
    norms = dict(trunk.norm_layers())
    want = {k[:-len(".running_mean")] for k in sd if k.endswith("running_mean")}
    assert set(norms) == want
    moved = [k for k, layer in norms.items() if not torch.equal(layer.running_mean.cpu(), sd[k + ".running_mean"])]
    assert len(moved) == len(want)


REFERENCE = "/root/reference"


@pytest.mark.parametrize("name", ["m0.9", "m1.1", "m2.3"])
@pytest.mark.skipif(not __import__("os").path.isdir(REFERENCE + "/sam3"), reason="the reference tree is only present in the build container")
def test_repvit_trunk_vs_the_reference_module(repvit_kernels, name):
    """where the reference is present (this container, never the GPU box): the same comparison against the REAL module -- repvit_m0_9 of
    sam3/backbones/repvit.py as stage1/model.py:386-395 builds it (num_classes 0, no distillation head), run layer by layer over
    ``model.features`` as RepViTAdapter.forward does (stage1/model.py:293-296), in train mode, loaded with the synthetic state dict --
    instead of this file's own restatement of it (all three students: 26 / 24 / 58 blocks)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pth in (os.path.join(root, "oracle", "shims"), REFERENCE + "/sam3"):      # as tests/test_train_blocks_host.py imports the EfficientViT module
        if pth not in sys.path:
            sys.path.insert(0, pth)
    from sam3.backbones import repvit as ref_repvit
    model = getattr(ref_repvit, "repvit_" + name.replace(".", "_"))(pretrained=False, num_classes=0, distillation=False)
    full = schema.synthetic_state_dict("repvit", name, seed=4)
    pre = "backbone.vision_backbone.trunk.model.backbone.model."
    sd = {k[len(pre):]: v.float() for k, v in full.items() if k.startswith(pre)}
    model.load_state_dict(sd, strict=True)
    model.train()
    g = torch.Generator().manual_seed(8)
    img = torch.randn(2, 3, 128, 96, generator=g)
    yr = img
    for layer in model.features:
        yr = layer(yr)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    ref_grads = {n: p.grad for n, p in model.named_parameters()}
    trunk = tr.RepViTTrunkTrain({k: v.clone() for k, v in sd.items()}, name, dtype=torch.float32)
    y = trunk.forward(img)
    grads = trunk.backward(_to_nhwc(dy))
    assert sorted(grads) == sorted(ref_grads)
    d, m = float((y - _to_nhwc(yr.detach())).abs().max()), float(yr.detach().abs().max())
    assert d <= 1e-3 * m, (d, m)
    typical = float(torch.stack([v.abs().max() for v in ref_grads.values()]).median())
    worst = max((float((grads[k] - ref_grads[k]).abs().max()) / max(float(ref_grads[k].abs().max()), 1e-2 * typical), k) for k in grads)
    print("worst relative gradient error against the reference module", worst)
    assert worst[0] <= 5e-3, worst
    # BatchNorm buffers after one training-mode forward: the module's own running statistics
    ref_buf = {k: v for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    for name, layer in trunk.norm_layers():
        for stat in ("running_mean", "running_var"):
            got, want = getattr(layer, stat), ref_buf[f"{name}.{stat}"]
            assert float((got - want).abs().max()) <= 1e-4 * max(float(want.abs().max()), 1.0), (name, stat)
