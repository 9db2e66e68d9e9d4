"""Gradient kernels of the student-trunk building blocks against torch.autograd: activations, the 1x1-conv /
Linear weight gradient (a reduction over the pixels on the matrix cores), the depthwise 3x3 weight gradient, and a whole
EfficientViT MBConv block (conv -> BatchNorm in TRAINING mode -> Hardswish, x3, + residual) run forwards and backwards on the HIP
kernels.  The reference layers are torch's own (backbones/efficientvit/nn/ops.py:39-81,310-360 build them from nn.Conv2d,
nn.BatchNorm2d, nn.Hardswish), so torch on the CPU in fp32 IS the reference; bf16 runs see bf16-quantised inputs."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TDT = {"f32": torch.float32, "bf16": torch.bfloat16}


def _close(got, ref, mode, what, f32=1e-4, bf16=2e-2):
    tol = (f32 if mode == "f32" else bf16) * max(1e-6, float(ref.abs().max()))
    d = float((got.float().cpu() - ref).abs().max())
    assert d <= tol, (what, mode, d, tol)


def _rel_l2(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm()) / max(1e-12, float(b.float().norm()))


def _inside_autocast_yardstick(pairs, ref16, what_test, factor=1.5, floor=1e-2):
    """bf16 blocks are held to the reference block's OWN bf16 behaviour: `ref16` is the same torch block run under
    torch.autocast("cpu", bfloat16); a quantity may be `factor` x as far (relative L2) from the fp32 run as that run is, plus a
    floor for quantities autocast happens to leave in fp32.  (An activation that lands on the other side of a Hardswish kink after
    rounding changes its derivative by 0.5, a legitimate O(1) change of single elements: hence L2, not the maximum.)"""
    rows, bad = [], []
    for got, ref, what in pairs:
        e, y = _rel_l2(got, ref), _rel_l2(ref16[what], ref)
        rows.append(f"{what} {e:.3f}/{y:.3f}")
        if e > factor * y + floor:
            bad.append((what, e, y))
    print(f"[{what_test} bf16] relative L2 error / the reference's own autocast distance: " + ", ".join(rows))
    assert not bad, bad


def _close_l2(got, ref, what, rel):
    """bf16 through a whole block: an activation that lands on the other side of a Hardswish kink after rounding changes its
    derivative by 0.5 (a legitimate O(1) change of single elements), so the block test bounds the RELATIVE L2 error in bf16 and the
    maximum error only in fp32."""
    num = float((got.float().cpu() - ref).norm())
    den = max(1e-12, float(ref.norm()))
    assert num / den <= rel, (what, num / den, rel)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("act", ["hswish", "relu", "gelu", None])
def test_activation_forward_backward(mode, act):
    from efficientsam3_amd import train_blocks as tb
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(5, 7, 24, generator=g) * 3.0).to(TDT[mode])
    x.view(-1)[:6] = torch.tensor([-3.0, 3.0, 0.0, -3.5, 3.5, 1e-3]).to(TDT[mode])      # Hardswish's kinks and flat parts
    dy = torch.randn(5, 7, 24, generator=g).to(TDT[mode])
    xr = x.float().clone().requires_grad_(True)
    fn = {"hswish": F.hardswish, "relu": F.relu, "gelu": F.gelu, None: lambda t: t * 1.0}[act]
    yr = fn(xr)
    yr.backward(dy.float())
    y = tb.act_forward(x.cuda(), act)
    dx = tb.act_backward(x.cuda(), dy.cuda(), act)
    _close(y, yr.detach(), mode, f"act {act} y", f32=2e-6, bf16=5e-3)
    _close(dx, xr.grad, mode, f"act {act} dx", f32=2e-6, bf16=5e-3)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(1000, 64, 32), (4133, 128, 256), (77, 8, 8), (20000, 96, 40), (64, 72, 136), (300000, 32, 128)])
def test_linear_wgrad(mode, M, N, K):
    """dw = dy^T x and dbias = sum dy over M rows (M not a multiple of the 64-row tile, N / K not multiples of the 64-wide tile)."""
    from efficientsam3_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(TDT[mode])
    dy = (torch.randn(M, N, generator=g) * 0.5).to(TDT[mode])
    ref = dy.double().t() @ x.double()
    lib = _lib.load()
    xd, dyd = x.cuda(), dy.cuda()
    dw = torch.empty((N, K), dtype=torch.float32, device="cuda")
    db = torch.empty(N, dtype=torch.float32, device="cuda")
    ws = torch.empty(int(lib.esam3_linear_wgrad_workspace(M, N, K)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.esam3_linear_wgrad(0 if mode == "f32" else 1, dyd.data_ptr(), xd.data_ptr(), M, N, K, dw.data_ptr(), db.data_ptr(),
                                      ws.data_ptr(), None), "esam3_linear_wgrad")
    torch.cuda.synchronize()
    # both modes accumulate exact products of their inputs in fp32: the error is summation error only
    tol = 2e-6 * float((dy.double().abs().t() @ x.double().abs()).max())
    assert float((dw.cpu().double() - ref).abs().max()) <= max(tol, 1e-5), (float((dw.cpu().double() - ref).abs().max()), tol)
    assert torch.allclose(db.cpu().double(), dy.double().sum(0), rtol=1e-5, atol=1e-5 * float(dy.double().abs().sum(0).max()))
    dw2 = torch.empty_like(dw)   # deterministic
    _lib.check(lib.esam3_linear_wgrad(0 if mode == "f32" else 1, dyd.data_ptr(), xd.data_ptr(), M, N, K, dw2.data_ptr(), None, ws.data_ptr(), None),
               "esam3_linear_wgrad")
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,Cc,stride,ks", [(2, 9, 7, 24, 1, 3), (1, 16, 16, 64, 2, 3), (3, 5, 6, 8, 1, 3), (2, 13, 11, 256, 2, 3),
                                                (1, 4, 4, 2048, 1, 3), (2, 9, 8, 48, 1, 5), (1, 12, 13, 384, 1, 5), (1, 7, 7, 16, 2, 5)])
def test_dwconv_wgrad_and_dgrad(mode, B, H, W, Cc, stride, ks):
    from efficientsam3_amd import train_blocks as tb
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, H, W, Cc, generator=g).to(TDT[mode])
    w = torch.randn(Cc, 1, ks, ks, generator=g) * 0.3
    oh, ow = (H + stride - 1) // stride, (W + stride - 1) // stride
    dy = torch.randn(B, oh, ow, Cc, generator=g).to(TDT[mode])
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=stride, padding=ks // 2, groups=Cc)
    assert yr.shape[2:] == (oh, ow)
    yr.backward(dy.float().permute(0, 3, 1, 2).contiguous())
    dw = tb.dwconv_wgrad(x.cuda(), dy.cuda(), stride, ks)
    _close(dw, wr.grad, mode, "dw wgrad", f32=1e-5, bf16=1e-5)    # exact products, fp32 sums in both modes
    dx = tb.dwconv_dgrad(dy.cuda(), w, (H, W), stride)     # stride 2: a transposed convolution
    dx_ref = xr.grad
    if mode == "bf16" and stride == 1:
        # round 5: the stride-1 data gradient runs on the forward depthwise kernels, which hold the taps in bf16 on the matrix cores
        # (as the reference's autocast rounds a conv's weight): the expectation is the gradient for the ROUNDED kernel
        xq = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        F.conv2d(xq, w.to(torch.bfloat16).float(), None, stride=1, padding=ks // 2, groups=Cc).backward(dy.float().permute(0, 3, 1, 2).contiguous())
        dx_ref = xq.grad
    _close(dx, dx_ref.permute(0, 2, 3, 1), mode, "dw dgrad", f32=1e-5, bf16=1e-2)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,Cin,Cmid,Cout,residual,stride", [(2, 12, 10, 32, 128, 32, True, 1), (1, 9, 9, 16, 64, 24, False, 1),
                                                                   (2, 13, 10, 32, 128, 64, False, 2)])
def test_mbconv_block_forward_backward_vs_autograd(mode, B, H, W, Cin, Cmid, Cout, residual, stride):
    """One ResidualBlock(MBConv) of the EfficientViT trunk in TRAINING mode, forwards and backwards on the HIP kernels, against the
    same block built from torch modules: output, input gradient, every weight / BatchNorm gradient, the running statistics."""
    from efficientsam3_amd import train_blocks as tb
    g = torch.Generator().manual_seed(11)
    mk = lambda *s, k=1.0: torch.randn(*s, generator=g) * k  # noqa: E731
    p = {"inverted.weight": mk(Cmid, Cin, k=Cin ** -0.5), "inverted.gamma": torch.rand(Cmid, generator=g) + 0.5, "inverted.beta": mk(Cmid, k=0.2),
         "depth.weight": mk(Cmid, 1, 3, 3, k=0.4), "depth.gamma": torch.rand(Cmid, generator=g) + 0.5, "depth.beta": mk(Cmid, k=0.2),
         "point.weight": mk(Cout, Cmid, k=Cmid ** -0.5), "point.gamma": torch.rand(Cout, generator=g) + 0.5, "point.beta": mk(Cout, k=0.2)}
    x = mk(B, H, W, Cin).to(TDT[mode])
    dy = mk(B, (H + stride - 1) // stride, (W + stride - 1) // stride, Cout).to(TDT[mode])
    # reference: fp32, and (bf16 mode) the same block under torch.autocast as the yardstick
    def reference(amp):
        rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        stats = {n: (torch.zeros(c), torch.ones(c)) for n, c in (("inverted", Cmid), ("depth", Cmid), ("point", Cout))}
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            h = F.conv2d(xr, rp["inverted.weight"].view(Cmid, Cin, 1, 1))
            h = F.hardswish(F.batch_norm(h, *stats["inverted"], rp["inverted.gamma"], rp["inverted.beta"], training=True, momentum=0.1, eps=1e-5))
            h = F.conv2d(h, rp["depth.weight"], None, stride=stride, padding=1, groups=Cmid)
            h = F.hardswish(F.batch_norm(h, *stats["depth"], rp["depth.gamma"], rp["depth.beta"], training=True, momentum=0.1, eps=1e-5))
            h = F.conv2d(h, rp["point.weight"].view(Cout, Cmid, 1, 1))
            h = F.batch_norm(h, *stats["point"], rp["point.gamma"], rp["point.beta"], training=True, momentum=0.1, eps=1e-5)
            yr = xr + h if residual else h
        yr.float().backward(dy.float().permute(0, 3, 1, 2).contiguous())
        out = {"y": yr.detach().float().permute(0, 2, 3, 1), "dx": xr.grad.permute(0, 2, 3, 1)}
        for name in ("inverted", "depth", "point"):
            out.update({f"{name}.weight": rp[f"{name}.weight"].grad, f"{name}.gamma": rp[f"{name}.gamma"].grad, f"{name}.beta": rp[f"{name}.beta"].grad,
                        f"{name}.running_mean": stats[name][0], f"{name}.running_var": stats[name][1]})
        return out

    r32 = reference(False)
    # HIP kernels
    blk = tb.MBConvTrain(p, residual=residual, stride=stride)
    y = blk.forward(x.cuda().contiguous())
    dx, grads = blk.backward(dy.cuda().contiguous())
    pairs = [(y, r32["y"], "y"), (dx, r32["dx"], "dx")]
    for name in ("inverted", "depth", "point"):
        pairs += [(grads[f"{name}.weight"].reshape(r32[f"{name}.weight"].shape), r32[f"{name}.weight"], f"{name}.weight"),
                  (grads[f"{name}.gamma"], r32[f"{name}.gamma"], f"{name}.gamma"), (grads[f"{name}.beta"], r32[f"{name}.beta"], f"{name}.beta")]
    for layer, name in ((blk.inv, "inverted"), (blk.dw, "depth"), (blk.pw, "point")):
        pairs += [(layer.running_mean, r32[f"{name}.running_mean"], f"{name}.running_mean"), (layer.running_var, r32[f"{name}.running_var"], f"{name}.running_var")]
    if mode == "f32":
        for got, ref, what in pairs:
            _close(got, ref, mode, what, 2e-4)
    else:
        _inside_autocast_yardstick(pairs, reference(True), f"mbconv {Cin}->{Cmid}->{Cout} s{stride}")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_dsconv_block_forward_backward_vs_autograd(mode):
    """ResidualBlock(DSConv) of the EfficientViT input stem in training mode (depthwise 3x3 + BN + Hardswish, 1x1 + BN, + x)."""
    from efficientsam3_amd import train_blocks as tb
    B, H, W, Cc = 2, 11, 14, 16
    g = torch.Generator().manual_seed(5)
    mk = lambda *s, k=1.0: torch.randn(*s, generator=g) * k  # noqa: E731
    p = {"depth.weight": mk(Cc, 1, 3, 3, k=0.4), "depth.gamma": torch.rand(Cc, generator=g) + 0.5, "depth.beta": mk(Cc, k=0.2),
         "point.weight": mk(Cc, Cc, k=Cc ** -0.5), "point.gamma": torch.rand(Cc, generator=g) + 0.5, "point.beta": mk(Cc, k=0.2)}
    x, dy = mk(B, H, W, Cc).to(TDT[mode]), mk(B, H, W, Cc).to(TDT[mode])
    rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    h = F.conv2d(xr, rp["depth.weight"], None, stride=1, padding=1, groups=Cc)
    h = F.hardswish(F.batch_norm(h, torch.zeros(Cc), torch.ones(Cc), rp["depth.gamma"], rp["depth.beta"], training=True, momentum=0.1, eps=1e-5))
    h = F.conv2d(h, rp["point.weight"].view(Cc, Cc, 1, 1))
    h = F.batch_norm(h, torch.zeros(Cc), torch.ones(Cc), rp["point.gamma"], rp["point.beta"], training=True, momentum=0.1, eps=1e-5)
    (xr + h).backward(dy.float().permute(0, 3, 1, 2).contiguous())
    blk = tb.DSConvTrain(p)
    y = blk.forward(x.cuda().contiguous())
    dx, grads = blk.backward(dy.cuda().contiguous())
    pairs = [(y, (xr + h).detach().permute(0, 2, 3, 1), "y"), (dx, xr.grad.permute(0, 2, 3, 1), "dx")]
    pairs += [(grads[k].reshape(rp[k].shape) if k.endswith("weight") else grads[k], rp[k].grad, k) for k in p]
    for got, ref, what in pairs:
        if mode == "f32":
            _close(got, ref, mode, what, 2e-4)
        else:
            _close_l2(got, ref, what, 1e-1)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,N,G,dim", [(2, 100, 4, 16), (1, 1024, 16, 16), (2, 77, 2, 32), (1, 3969, 2, 16), (2, 1100, 3, 32), (1, 257, 2, 16)])
def test_lite_mla_backward_vs_autograd(mode, B, N, G, dim):
    """esam3_lite_mla_backward against autograd through the reference's formula (LiteMLA.relu_linear_att, ops.py:584-621: ReLU kernels,
    v padded with a row of ones, vk = v k^T, out = vk q, normalised by its last row + 1e-15), fp32 on the (bf16-quantised) inputs."""
    from efficientsam3_amd import train_blocks as tb
    g = torch.Generator().manual_seed(N + G)
    ms = (torch.randn(B, N, G * 3 * dim, generator=g) * 0.8 + 0.2).to(TDT[mode])
    dout = torch.randn(B, N, G * dim, generator=g).to(TDT[mode])
    qkv = ms.float().permute(0, 2, 1).reshape(B, G, 3 * dim, N).clone().requires_grad_(True)     # the reference's [B, -1, 3 dim, HW]
    q, k, v = F.relu(qkv[:, :, :dim]), F.relu(qkv[:, :, dim:2 * dim]), qkv[:, :, 2 * dim:]
    vp = F.pad(v, (0, 0, 0, 1), mode="constant", value=1)
    out = torch.matmul(torch.matmul(vp, k.transpose(-1, -2)), q)
    out = out[:, :, :-1] / (out[:, :, -1:] + 1e-15)
    yr = out.reshape(B, G * dim, N)
    yr.backward(dout.float().permute(0, 2, 1).contiguous())
    dref = qkv.grad.reshape(B, G * 3 * dim, N).permute(0, 2, 1)
    dms, y = tb.lite_mla_backward(ms.cuda().contiguous(), dout.cuda().contiguous(), G, dim)
    y_fwd = tb.lite_mla_forward(ms.cuda().contiguous(), G, dim)          # the forward-only entry (dout = NULL)
    _close(y_fwd, y.float().cpu(), mode, "forward-only y vs the backward kernels' y", 1e-5, 1e-5) if mode == "f32" else _close_l2(y_fwd, y.float().cpu(), "y fwd", 1e-2)
    if mode == "f32":
        _close(y, yr.detach().permute(0, 2, 1), mode, "y", 2e-4)
        _close(dms, dref, mode, "d_ms", 5e-4)
    else:
        _close_l2(y, yr.detach().permute(0, 2, 1), "y", 1e-2)
        _close_l2(dms, dref, "d_ms", 2e-2)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_lite_mla_block_forward_backward_vs_autograd(mode):
    """ResidualBlock(LiteMLA) -- the context module of an EfficientViTBlock (ops.py:521-640) -- in training mode, forwards and backwards on the
    HIP kernels, against the same block written with torch functions: qkv 1x1, depthwise 5x5 + grouped 1x1 aggregation, ReLU linear
    attention over both scales, proj 1x1 + BatchNorm, + x."""
    from efficientsam3_amd import train_blocks as tb
    B, H, W, Cc, dim = 2, 9, 8, 32, 16
    heads = Cc // dim
    g = torch.Generator().manual_seed(21)
    mk = lambda *s, k=1.0: torch.randn(*s, generator=g) * k  # noqa: E731
    p = {"qkv.weight": mk(3 * Cc, Cc, k=Cc ** -0.5), "aggreg.dw.weight": mk(3 * Cc, 1, 5, 5, k=0.2), "aggreg.pw.weight": mk(3 * Cc, dim, 1, 1, k=dim ** -0.5),
         "proj.weight": mk(Cc, 2 * Cc, k=(2 * Cc) ** -0.5), "proj.gamma": torch.rand(Cc, generator=g) + 0.5, "proj.beta": mk(Cc, k=0.2)}
    x, dy = mk(B, H, W, Cc).to(TDT[mode]), mk(B, H, W, Cc).to(TDT[mode])
    rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    qkv = F.conv2d(xr, rp["qkv.weight"].view(3 * Cc, Cc, 1, 1))
    agg = F.conv2d(F.conv2d(qkv, rp["aggreg.dw.weight"], None, padding=2, groups=3 * Cc), rp["aggreg.pw.weight"], None, groups=3 * heads)
    ms = torch.cat([qkv, agg], dim=1).reshape(B, -1, 3 * dim, H * W)
    q, k, v = F.relu(ms[:, :, :dim]), F.relu(ms[:, :, dim:2 * dim]), ms[:, :, 2 * dim:]
    out = torch.matmul(torch.matmul(F.pad(v, (0, 0, 0, 1), value=1), k.transpose(-1, -2)), q)
    att = (out[:, :, :-1] / (out[:, :, -1:] + 1e-15)).reshape(B, -1, H, W)
    h = F.batch_norm(F.conv2d(att, rp["proj.weight"].view(Cc, 2 * Cc, 1, 1)), torch.zeros(Cc), torch.ones(Cc), rp["proj.gamma"], rp["proj.beta"],
                     training=True, momentum=0.1, eps=1e-5)
    (xr + h).backward(dy.float().permute(0, 3, 1, 2).contiguous())
    blk = tb.LiteMLATrain(p, dim)
    y = blk.forward(x.cuda().contiguous())
    dx, grads = blk.backward(dy.cuda().contiguous())
    pairs = [(y, (xr + h).detach().permute(0, 2, 3, 1), "y"), (dx, xr.grad.permute(0, 2, 3, 1), "dx")]
    pairs += [(grads[k].reshape(rp[k].shape), rp[k].grad, k) for k in p]
    for got, ref, what in pairs:
        if mode == "f32":
            _close(got, ref, mode, what, 5e-4)
        else:
            _close_l2(got, ref, what, 1e-1)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("M,N", [(1000, 64), (77, 8), (300000, 128), (4133, 200)])
def test_colsum(mode, M, N):
    """esam3_colsum: the gradient of a conv bias (sum of dy over all rows), fixed-order split reduction."""
    from efficientsam3_amd import train_blocks as tb
    dy = torch.randn(M, N, generator=torch.Generator().manual_seed(M + N)).to(TDT[mode])
    got = tb.colsum(dy.cuda())
    ref = dy.double().sum(0)
    assert float((got.cpu().double() - ref).abs().max()) <= 2e-6 * float(dy.double().abs().sum(0).max())
    assert torch.equal(got, tb.colsum(dy.cuda()))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_efficientvit_block_forward_backward_vs_autograd(mode):
    """A whole EfficientViTBlock (ops.py:670-730) in training mode on the HIP kernels: ResidualBlock(LiteMLA) then ResidualBlock(MBConv with
    conv biases on its first two layers and one BatchNorm at the end) -- the repeated block of stages 3 and 4 -- against the same block written
    with torch functions: output, input gradient and all thirteen parameter gradients."""
    from efficientsam3_amd import train_blocks as tb
    B, H, W, Cc, dim, Cmid = 2, 8, 9, 32, 16, 128
    heads = Cc // dim
    g = torch.Generator().manual_seed(33)
    mk = lambda *s, k=1.0: torch.randn(*s, generator=g) * k  # noqa: E731
    x, dy = mk(B, H, W, Cc).to(TDT[mode]), mk(B, H, W, Cc).to(TDT[mode])
    pc = {"qkv.weight": mk(3 * Cc, Cc, k=Cc ** -0.5), "aggreg.dw.weight": mk(3 * Cc, 1, 5, 5, k=0.2), "aggreg.pw.weight": mk(3 * Cc, dim, 1, 1, k=0.25),
          "proj.weight": mk(Cc, 2 * Cc, k=0.12), "proj.gamma": torch.rand(Cc, generator=g) + 0.5, "proj.beta": mk(Cc, k=0.2)}
    pl = {"inverted.weight": mk(Cmid, Cc, k=0.2), "inverted.bias": mk(Cmid, k=0.3), "depth.weight": mk(Cmid, 1, 3, 3, k=0.4), "depth.bias": mk(Cmid, k=0.3),
          "point.weight": mk(Cc, Cmid, k=0.1), "point.gamma": torch.rand(Cc, generator=g) + 0.5, "point.beta": mk(Cc, k=0.2)}
    def reference(amp):
        rc = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
        rl = {k: v.clone().requires_grad_(True) for k, v in pl.items()}
        bn = lambda h, ga, be: F.batch_norm(h, torch.zeros(ga.numel()), torch.ones(ga.numel()), ga, be, training=True, momentum=0.1, eps=1e-5)  # noqa: E731
        xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            qkv = F.conv2d(xr, rc["qkv.weight"].view(3 * Cc, Cc, 1, 1))
            agg = F.conv2d(F.conv2d(qkv, rc["aggreg.dw.weight"], None, padding=2, groups=3 * Cc), rc["aggreg.pw.weight"], None, groups=3 * heads)
            # LiteMLA.forward is decorated @autocast(enabled=False) and computes relu linear attention in fp32 (ops.py:588-639)
            with torch.autocast("cpu", enabled=False):
                ms = torch.cat([qkv, agg], dim=1).float().reshape(B, -1, 3 * dim, H * W)
                q, k, v = F.relu(ms[:, :, :dim]), F.relu(ms[:, :, dim:2 * dim]), ms[:, :, 2 * dim:]
                out = torch.matmul(torch.matmul(F.pad(v, (0, 0, 0, 1), value=1), k.transpose(-1, -2)), q)
                att = (out[:, :, :-1] / (out[:, :, -1:] + 1e-15)).reshape(B, -1, H, W)
            c1 = xr + bn(F.conv2d(att, rc["proj.weight"].view(Cc, 2 * Cc, 1, 1)), rc["proj.gamma"], rc["proj.beta"])
            h = F.hardswish(F.conv2d(c1, rl["inverted.weight"].view(Cmid, Cc, 1, 1), rl["inverted.bias"]))
            h = F.hardswish(F.conv2d(h, rl["depth.weight"], rl["depth.bias"], padding=1, groups=Cmid))
            yr = c1 + bn(F.conv2d(h, rl["point.weight"].view(Cc, Cmid, 1, 1)), rl["point.gamma"], rl["point.beta"])
        yr.float().backward(dy.float().permute(0, 3, 1, 2).contiguous())
        out = {"y": yr.detach().float().permute(0, 2, 3, 1), "dx": xr.grad.permute(0, 2, 3, 1)}
        out.update({f"context.{k}": rc[k].grad for k in pc})
        out.update({f"local.{k}": rl[k].grad for k in pl})
        return out

    r32 = reference(False)
    blk = tb.EfficientViTBlockTrain(pc, pl, dim)
    y = blk.forward(x.cuda().contiguous())
    dx, grads = blk.backward(dy.cuda().contiguous())
    pairs = [(y, r32["y"], "y"), (dx, r32["dx"], "dx")]
    pairs += [(grads[f"context.{k}"].reshape(pc[k].shape), r32[f"context.{k}"], f"context.{k}") for k in pc]
    pairs += [(grads[f"local.{k}"].reshape(pl[k].shape), r32[f"local.{k}"], f"local.{k}") for k in pl]
    if mode == "f32":
        for got, ref, what in pairs:
            _close(got, ref, mode, what, 1e-3)
    else:
        _inside_autocast_yardstick(pairs, reference(True), "EfficientViTBlock")


@pytest.mark.parametrize("size,fwd,worst,median", [(128, 1e-4, 5e-3, 1e-3), (1008, 2e-3, 1e-1, 1.5e-2)])
def test_trunk_train_vs_autograd_f32(size, fwd, worst, median):
    """The whole EfficientViT-B1 trunk in TRAINING mode (EfficientViTTrunkTrain: every block class chained, BatchNorm batch statistics)
    forwards and backwards on the HIP kernels in fp32, against torch.autograd through the same architecture written with torch functions
    (tools/trunk_train_gpu_check.py; its layer list is pinned against the real reference module on the CPU in tests/test_train_blocks_host.py).
    128^2 and the REAL 1008^2 input (batch 2): relative L2 of the output and of every one of the 159 parameter gradients.  The bf16 trunk at
    the real shape is covered by tests/test_stage1_step.py against the reference's own autocast run."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import trunk_train_gpu_check as chk
    from efficientsam3_amd import train_blocks as tb
    sd = chk.make_state_dict()
    img = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(1))
    out, p = chk.reference_forward(sd, img)
    dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    out.backward(dy)
    trunk = tb.EfficientViTTrunkTrain(sd, chk.WIDTHS, chk.DEPTHS, chk.DIM, dtype=torch.float32)
    y = trunk.forward(img.cuda())
    grads = trunk.backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
    e_y = _rel_l2(y, out.detach().permute(0, 2, 3, 1))
    gmax = max(float(v.grad.abs().max()) for v in p.values())
    rows = sorted(((float((grads[n].reshape(v.shape).float().cpu() - v.grad).norm()) / max(float(v.grad.norm()), 1e-4 * gmax * v.numel() ** 0.5), n)
                   for n, v in p.items()), reverse=True)
    print(f"[trunk f32 @{size}] output rel L2 {e_y:.3e}; {len(rows)} gradients: worst {rows[0][0]:.3e} ({rows[0][1]}), median {rows[len(rows) // 2][0]:.3e}")
    assert e_y <= fwd and rows[0][0] <= worst and rows[len(rows) // 2][0] <= median


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("act", ["hswish", "gelu", "relu"])
@pytest.mark.parametrize("C", [48, 64])   # 64: a thread's channel group is fixed (constants in registers); 48: the general map
def test_bn_act_fused_passes_equal_the_separate_ones(mode, act, C):
    """``esam3_bn_act_train_forward / _backward`` (the ConvLayer's activation inside the BatchNorm kernels' passes) against BatchNorm then
    activation as separate kernels: same y, same act(y) bit for bit; gradients equal up to the one rounding the separate form spends on
    storing dy act'(pre) (none in fp32).  The recomputing backward (no saved BatchNorm output: ``esam3_bn_act_train_backward_rc``) equals the
    one that reads it bit for bit, and the forward that does not write y gives the same act(y)."""
    from efficientsam3_amd import stage1, train_blocks as tb
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(3, 17, 19, C, generator=g) * 2.0 + 1.0).to(TDT[mode]).cuda()
    dy = torch.randn(3, 17, 19, C, generator=g).to(TDT[mode]).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5).cuda()
    rm1, rv1, rm2, rv2 = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros(C).cuda(), torch.ones(C).cuda()
    y1, m1, r1 = stage1.bn_train_forward(x, gamma, beta, rm1, rv1, 0.1, 1e-5)
    a1 = tb.act_forward(y1, act)
    y2, a2, m2, r2 = stage1.bn_act_train_forward(x, gamma, beta, rm2, rv2, 0.1, 1e-5, act)
    assert torch.equal(y1, y2) and torch.equal(a1, a2) and torch.equal(m1, m2) and torch.equal(r1, r2) and torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
    dx1, dg1, db1 = stage1.bn_train_backward(x, tb.act_backward(y1, dy, act), gamma, m1, r1)
    dx2, dg2, db2 = stage1.bn_act_train_backward(x, dy, y2, act, gamma, m2, r2)
    tol = 2e-6 if mode == "f32" else 1.5e-2
    for got, ref, what in ((dx2, dx1, "dx"), (dg2, dg1, "dgamma"), (db2, db1, "dbeta")):
        d, m = float((got.float() - ref.float()).abs().max()), float(ref.float().abs().max())
        assert d <= tol * m, (what, mode, act, d, m)
    rm3, rv3 = torch.zeros(C).cuda(), torch.ones(C).cuda()
    y3, a3, m3, r3 = stage1.bn_act_train_forward(x, gamma, beta, rm3, rv3, 0.1, 1e-5, act, keep_pre=False)
    assert y3 is None and torch.equal(a3, a2) and torch.equal(m3, m2) and torch.equal(r3, r2) and torch.equal(rm3, rm2) and torch.equal(rv3, rv2)
    dx3, dg3, db3 = stage1.bn_act_train_backward(x, dy, None, act, gamma, m2, r2, beta=beta)
    assert torch.equal(dx3, dx2) and torch.equal(dg3, dg2) and torch.equal(db3, db2)
