"""CPU: the COMPOSITION logic of efficientsam3_amd/train_blocks.py (which tensor goes into which gradient kernel, the channel layout of the
multi-scale qkv tensor, the block-diagonal form of the grouped 1x1 conv, residual paths) with every kernel wrapper replaced by a plain torch
stand-in of the same contract, against torch.autograd of the block written with torch functions.  The kernels themselves are checked on
the GPU by tests/test_train_blocks.py; this test keeps the host side honest where there is no GPU."""
import pytest
import torch
import torch.nn.functional as F

from efficientsam3_amd import train_blocks as tb


def _to_nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _to_nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _act(x, act):
    return {None: lambda t: t, "relu": F.relu, "gelu": F.gelu, "hswish": F.hardswish, "sigmoid": torch.sigmoid}[act](x)


def install_cpu_kernels(monkeypatch):
    """stand-ins with the kernels' contracts (NHWC tensors, PyTorch-layout weights, fp32 weight gradients)"""
    def act_backward(x, dy, act):
        xr = x.clone().requires_grad_(True)
        _act(xr, act).backward(dy)
        return xr.grad

    def bn_fwd(x, gamma, beta, rm, rv, momentum, eps, sync=None):   # `sync`: the layer's SyncBatchNorm setting (None here: one rank)
        c = x.shape[-1]
        x2 = x.reshape(-1, c)
        mean, var = x2.mean(0), x2.var(0, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + eps)
        n = x2.shape[0]
        rm.mul_(1 - momentum).add_(momentum * mean)
        rv.mul_(1 - momentum).add_(momentum * var * n / (n - 1))
        return ((x - mean) * rstd * gamma + beta), mean, rstd

    def bn_bwd(x, dy, gamma, mean, rstd, sync=None):
        c = x.shape[-1]
        n = x.numel() // c
        xh = (x - mean) * rstd
        dbeta, dgamma = dy.reshape(n, c).sum(0), (dy * xh).reshape(n, c).sum(0)
        return gamma * rstd * (dy - dbeta / n - xh * dgamma / n), dgamma, dbeta

    def dw_fwd(x, w, stride=1, bias=None):
        return _to_nhwc(F.conv2d(_to_nchw(x), w, bias, stride=stride, padding=w.shape[-1] // 2, groups=x.shape[-1]))

    def dw_dgrad(dy, w, in_hw, stride=1):
        xr = torch.zeros((dy.shape[0], dy.shape[-1]) + tuple(in_hw), requires_grad=True)
        F.conv2d(xr, w, None, stride=stride, padding=w.shape[-1] // 2, groups=dy.shape[-1]).backward(_to_nchw(dy))
        return _to_nhwc(xr.grad)

    def dw_wgrad(x, dy, stride=1, ksize=3):
        wr = torch.zeros((x.shape[-1], 1, ksize, ksize), requires_grad=True)
        F.conv2d(_to_nchw(x), wr, None, stride=stride, padding=ksize // 2, groups=x.shape[-1]).backward(_to_nchw(dy))
        return wr.grad

    def mla_bwd(ms, dout, groups, dim, eps=1e-15):
        b, n, _ = ms.shape
        qkv = ms.permute(0, 2, 1).reshape(b, groups, 3 * dim, n).clone().requires_grad_(True)
        q, k, v = F.relu(qkv[:, :, :dim]), F.relu(qkv[:, :, dim:2 * dim]), qkv[:, :, 2 * dim:]
        out = torch.matmul(torch.matmul(F.pad(v, (0, 0, 0, 1), value=1), k.transpose(-1, -2)), q)
        y = (out[:, :, :-1] / (out[:, :, -1:] + eps)).reshape(b, groups * dim, n)
        y.backward(dout.permute(0, 2, 1).contiguous())
        return qkv.grad.reshape(b, groups * 3 * dim, n).permute(0, 2, 1).contiguous(), y.detach().permute(0, 2, 1).contiguous()

    def channel_scale(x, mul, bias=None, add=None, plus_one=False, bias_scale=1.0):
        b, c = x.shape[0], x.shape[-1]
        shape = lambda t: t.reshape((b,) + (1,) * (x.dim() - 2) + (c,)) if t.dim() == 2 else t  # noqa: E731
        out = x * (shape(mul) + (1.0 if plus_one else 0.0))
        if add is not None:
            out = out + add
        if bias is not None:
            out = out + shape(bias) * bias_scale
        return out

    def batched_coldot(a, b2=None, scale=1.0, per_image=True):
        prod = a if b2 is None else a * b2
        c = a.shape[-1]
        return prod.reshape(a.shape[0], -1, c).sum(1) * scale if per_image else prod.reshape(-1, c).sum(0) * scale

    monkeypatch.setattr(tb, "channel_scale", channel_scale)
    monkeypatch.setattr(tb, "batched_coldot", batched_coldot)
    monkeypatch.setattr(tb, "FUSE_BN_ACT", False)       # the compositions are checked with the unfused BatchNorm / activation stand-ins
    monkeypatch.setattr(tb, "DEVICE", "cpu")
    monkeypatch.setattr(tb, "act_forward", lambda x, act: _act(x, act))
    monkeypatch.setattr(tb, "act_backward", act_backward)
    monkeypatch.setattr(tb, "linear_forward", lambda x, w, bias=None: x @ w.t() + (0 if bias is None else bias))
    monkeypatch.setattr(tb, "colsum", lambda dy: dy.reshape(-1, dy.shape[-1]).sum(0))
    monkeypatch.setattr(tb, "linear_dgrad", lambda dy, w: dy @ w)
    monkeypatch.setattr(tb, "linear_wgrad", lambda dy, x: dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1]))
    monkeypatch.setattr(tb, "dwconv_forward", dw_fwd)
    monkeypatch.setattr(tb, "dwconv_dgrad", dw_dgrad)
    monkeypatch.setattr(tb, "dwconv_wgrad", dw_wgrad)
    monkeypatch.setattr(tb, "bn_train_forward", bn_fwd)
    monkeypatch.setattr(tb, "bn_train_backward", bn_bwd)
    monkeypatch.setattr(tb, "lite_mla_backward", mla_bwd)
    monkeypatch.setattr(tb, "lite_mla_forward", lambda ms, groups, dim, eps=1e-15: mla_bwd(ms, torch.zeros(ms.shape[0], ms.shape[1], groups * dim), groups, dim, eps)[1])


@pytest.fixture
def cpu_kernels(monkeypatch):
    install_cpu_kernels(monkeypatch)


def _bn(h, gamma, beta):
    c = gamma.numel()
    return F.batch_norm(h, torch.zeros(c), torch.ones(c), gamma, beta, training=True, momentum=0.1, eps=1e-5)


def _check(pairs, tol=2e-4):
    for got, ref, what in pairs:
        d, m = float((got - ref).abs().max()), float(ref.abs().max())
        assert d <= tol * max(m, 1e-6), (what, d, m)


@pytest.mark.parametrize("stride,residual", [(1, True), (2, False)])
def test_mbconv_composition(cpu_kernels, stride, residual):
    B, H, W, Cin, Cmid = 2, 9, 8, 16, 48
    Cout = Cin if residual else 24
    g = torch.Generator().manual_seed(1)
    mk = lambda *s, k=1.0: torch.randn(*s, generator=g) * k  # noqa: E731
    p = {"inverted.weight": mk(Cmid, Cin, k=0.3), "inverted.gamma": torch.rand(Cmid, generator=g) + 0.5, "inverted.beta": mk(Cmid, k=0.2),
         "depth.weight": mk(Cmid, 1, 3, 3, k=0.4), "depth.gamma": torch.rand(Cmid, generator=g) + 0.5, "depth.beta": mk(Cmid, k=0.2),
         "point.weight": mk(Cout, Cmid, k=0.2), "point.gamma": torch.rand(Cout, generator=g) + 0.5, "point.beta": mk(Cout, k=0.2)}
    x = mk(B, H, W, Cin)
    dy = mk(B, (H + stride - 1) // stride, (W + stride - 1) // stride, Cout)
    rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = _to_nchw(x).requires_grad_(True)
    h = F.hardswish(_bn(F.conv2d(xr, rp["inverted.weight"].view(Cmid, Cin, 1, 1)), rp["inverted.gamma"], rp["inverted.beta"]))
    h = F.hardswish(_bn(F.conv2d(h, rp["depth.weight"], None, stride=stride, padding=1, groups=Cmid), rp["depth.gamma"], rp["depth.beta"]))
    h = _bn(F.conv2d(h, rp["point.weight"].view(Cout, Cmid, 1, 1)), rp["point.gamma"], rp["point.beta"])
    yr = xr + h if residual else h
    yr.backward(_to_nchw(dy))
    blk = tb.MBConvTrain(p, residual=residual, stride=stride)
    y = blk.forward(x)
    dx, grads = blk.backward(dy)
    _check([(y, _to_nhwc(yr.detach()), "y"), (dx, _to_nhwc(xr.grad), "dx")] + [(grads[k].reshape(rp[k].shape), rp[k].grad, k) for k in p])


def test_dsconv_and_lite_mla_composition(cpu_kernels):
    B, H, W, Cc, dim = 2, 7, 6, 32, 16
    heads = Cc // dim
    g = torch.Generator().manual_seed(2)
    mk = lambda *s, k=1.0: torch.randn(*s, generator=g) * k  # noqa: E731
    x, dy = mk(B, H, W, Cc), mk(B, H, W, Cc)
    # DSConv
    p = {"depth.weight": mk(Cc, 1, 3, 3, k=0.4), "depth.gamma": torch.rand(Cc, generator=g) + 0.5, "depth.beta": mk(Cc, k=0.2),
         "point.weight": mk(Cc, Cc, k=0.2), "point.gamma": torch.rand(Cc, generator=g) + 0.5, "point.beta": mk(Cc, k=0.2)}
    rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = _to_nchw(x).requires_grad_(True)
    h = F.hardswish(_bn(F.conv2d(xr, rp["depth.weight"], None, padding=1, groups=Cc), rp["depth.gamma"], rp["depth.beta"]))
    h = _bn(F.conv2d(h, rp["point.weight"].view(Cc, Cc, 1, 1)), rp["point.gamma"], rp["point.beta"])
    (xr + h).backward(_to_nchw(dy))
    blk = tb.DSConvTrain(p)
    y = blk.forward(x)
    dx, grads = blk.backward(dy)
    _check([(y, _to_nhwc((xr + h).detach()), "y"), (dx, _to_nhwc(xr.grad), "dx")] + [(grads[k].reshape(rp[k].shape), rp[k].grad, k) for k in p])
    # LiteMLA
    p = {"qkv.weight": mk(3 * Cc, Cc, k=Cc ** -0.5), "aggreg.dw.weight": mk(3 * Cc, 1, 5, 5, k=0.2), "aggreg.pw.weight": mk(3 * Cc, dim, 1, 1, k=0.25),
         "proj.weight": mk(Cc, 2 * Cc, k=0.12), "proj.gamma": torch.rand(Cc, generator=g) + 0.5, "proj.beta": mk(Cc, k=0.2)}
    rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = _to_nchw(x).requires_grad_(True)
    qkv = F.conv2d(xr, rp["qkv.weight"].view(3 * Cc, Cc, 1, 1))
    agg = F.conv2d(F.conv2d(qkv, rp["aggreg.dw.weight"], None, padding=2, groups=3 * Cc), rp["aggreg.pw.weight"], None, groups=3 * heads)
    ms = torch.cat([qkv, agg], dim=1).reshape(B, -1, 3 * dim, H * W)
    q, k, v = F.relu(ms[:, :, :dim]), F.relu(ms[:, :, dim:2 * dim]), ms[:, :, 2 * dim:]
    out = torch.matmul(torch.matmul(F.pad(v, (0, 0, 0, 1), value=1), k.transpose(-1, -2)), q)
    att = (out[:, :, :-1] / (out[:, :, -1:] + 1e-15)).reshape(B, -1, H, W)
    h = _bn(F.conv2d(att, rp["proj.weight"].view(Cc, 2 * Cc, 1, 1)), rp["proj.gamma"], rp["proj.beta"])
    (xr + h).backward(_to_nchw(dy))
    blk = tb.LiteMLATrain(p, dim)
    y = blk.forward(x)
    dx, grads = blk.backward(dy)
    _check([(y, _to_nhwc((xr + h).detach()), "y"), (dx, _to_nhwc(xr.grad), "dx")] + [(grads[k].reshape(rp[k].shape), rp[k].grad, k) for k in p], tol=1e-3)


def test_efficientvit_block_composition(cpu_kernels):
    """EfficientViTBlock = ResidualBlock(LiteMLA) then ResidualBlock(MBConv with conv biases on the first two layers, one BatchNorm last)."""
    B, H, W, Cc, dim, Cmid = 2, 6, 7, 32, 16, 128
    heads = Cc // dim
    g = torch.Generator().manual_seed(3)
    mk = lambda *s, k=1.0: torch.randn(*s, generator=g) * k  # noqa: E731
    x, dy = mk(B, H, W, Cc), mk(B, H, W, Cc)
    pc = {"qkv.weight": mk(3 * Cc, Cc, k=Cc ** -0.5), "aggreg.dw.weight": mk(3 * Cc, 1, 5, 5, k=0.2), "aggreg.pw.weight": mk(3 * Cc, dim, 1, 1, k=0.25),
          "proj.weight": mk(Cc, 2 * Cc, k=0.12), "proj.gamma": torch.rand(Cc, generator=g) + 0.5, "proj.beta": mk(Cc, k=0.2)}
    pl = {"inverted.weight": mk(Cmid, Cc, k=0.2), "inverted.bias": mk(Cmid, k=0.3), "depth.weight": mk(Cmid, 1, 3, 3, k=0.4), "depth.bias": mk(Cmid, k=0.3),
          "point.weight": mk(Cc, Cmid, k=0.1), "point.gamma": torch.rand(Cc, generator=g) + 0.5, "point.beta": mk(Cc, k=0.2)}
    rc = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    rl = {k: v.clone().requires_grad_(True) for k, v in pl.items()}
    xr = _to_nchw(x).requires_grad_(True)
    qkv = F.conv2d(xr, rc["qkv.weight"].view(3 * Cc, Cc, 1, 1))
    agg = F.conv2d(F.conv2d(qkv, rc["aggreg.dw.weight"], None, padding=2, groups=3 * Cc), rc["aggreg.pw.weight"], None, groups=3 * heads)
    ms = torch.cat([qkv, agg], dim=1).reshape(B, -1, 3 * dim, H * W)
    q, k, v = F.relu(ms[:, :, :dim]), F.relu(ms[:, :, dim:2 * dim]), ms[:, :, 2 * dim:]
    out = torch.matmul(torch.matmul(F.pad(v, (0, 0, 0, 1), value=1), k.transpose(-1, -2)), q)
    att = (out[:, :, :-1] / (out[:, :, -1:] + 1e-15)).reshape(B, -1, H, W)
    c1 = xr + _bn(F.conv2d(att, rc["proj.weight"].view(Cc, 2 * Cc, 1, 1)), rc["proj.gamma"], rc["proj.beta"])
    h = F.hardswish(F.conv2d(c1, rl["inverted.weight"].view(Cmid, Cc, 1, 1), rl["inverted.bias"]))
    h = F.hardswish(F.conv2d(h, rl["depth.weight"], rl["depth.bias"], padding=1, groups=Cmid))
    yr = c1 + _bn(F.conv2d(h, rl["point.weight"].view(Cc, Cmid, 1, 1)), rl["point.gamma"], rl["point.beta"])
    yr.backward(_to_nchw(dy))
    blk = tb.EfficientViTBlockTrain(pc, pl, dim)
    y = blk.forward(x)
    dx, grads = blk.backward(dy)
    pairs = [(y, _to_nhwc(yr.detach()), "y"), (dx, _to_nhwc(xr.grad), "dx")]
    pairs += [(grads[f"context.{k}"].reshape(rc[k].shape), rc[k].grad, f"context.{k}") for k in pc]
    pairs += [(grads[f"local.{k}"].reshape(rl[k].shape), rl[k].grad, f"local.{k}") for k in pl]
    _check(pairs, tol=2e-3)


def test_trunk_composition_against_the_reference_module(cpu_kernels, monkeypatch):
    """EfficientViTTrunkTrain built from the state dict of the REAL reference module (efficientvit_backbone_b1 in train mode) -- layer order,
    strides, which layers carry a BatchNorm and which a bias, the state-dict names of every gradient -- with the kernel wrappers replaced by torch
    stand-ins: output and all parameter gradients against autograd through the reference module itself.  Needs /root/reference (build container)."""
    import os
    import sys
    if not os.path.isdir("/root/reference/sam3"):
        pytest.skip("the reference checkout is not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pth in (os.path.join(root, "oracle", "shims"), "/root/reference/sam3"):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    from sam3.backbones.efficientvit.backbone import efficientvit_backbone_b1
    monkeypatch.setattr(tb, "stem_forward", lambda img, w, dtype: _to_nhwc(F.conv2d(img, w, None, stride=2, padding=1)))
    torch.manual_seed(0)
    ref = efficientvit_backbone_b1()
    with torch.no_grad():   # non-trivial BatchNorm parameters and conv biases
        for n, p_ in ref.named_parameters():
            if n.endswith("norm.weight"):
                p_.uniform_(0.5, 1.5)
            elif n.endswith(".bias"):
                p_.normal_(0.0, 0.2)
    ref.train()
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    img = torch.randn(2, 3, 128, 128)
    out = ref(img)["stage_final"]
    dy = torch.randn_like(out)
    out.backward(dy)
    trunk = tb.EfficientViTTrunkTrain(sd, [16, 32, 64, 128, 256], [1, 2, 3, 3, 4], dim=16)
    y = trunk.forward(img)
    grads = trunk.backward(_to_nhwc(dy))
    _check([(y, _to_nhwc(out.detach()), "stage_final")], tol=2e-3)
    named = dict(ref.named_parameters())
    assert set(grads) == set(named), (sorted(set(named) - set(grads))[:5], sorted(set(grads) - set(named))[:5])
    # A BatchNorm shift (or conv bias) that feeds straight into another BatchNorm has an analytically ZERO gradient (the next layer subtracts
    # the batch mean): such gradients are rounding noise in both implementations, so the bound is relative to the parameter's own gradient
    # scale with a floor at 1e-4 of the largest gradient in the network.  2 % : fp32 through ~60 BatchNorms over batches as small as 32 samples.
    gmax = max(float(p_.grad.abs().max()) for p_ in named.values())
    for n, p_ in named.items():
        err = float((grads[n].reshape(p_.shape) - p_.grad).abs().max())
        assert err <= 2.5e-2 * float(p_.grad.abs().max()) + 1e-4 * gmax, (n, err, float(p_.grad.abs().max()), gmax)
    # the first ConvLayer's running statistics moved like the module's buffers
    assert torch.allclose(trunk.stem.running_mean, ref.state_dict()["input_stem.op_list.0.norm.running_mean"], atol=1e-5)
