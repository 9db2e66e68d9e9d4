"""GPU: the kernels and layers the TinyViT students need in training mode against torch.autograd --
``esam3_ln_train_*`` (LayerNorm), ``esam3_win_attn_train_*`` (window attention with the gathered bias table), ``esam3_attn_bias_gather_sum``,
then whole layers (window attention with its Linear layers, TinyViTBlock with padding and DropPath factors, MBConv, PatchMerging:
sam3/backbones/tiny_vit.py:87-154,196-386) forwards and backwards on the HIP kernels with DEVICE-resident fp32 parameters.  torch on the CPU
in fp32 is the reference; bf16 runs see bf16-quantised inputs and are held to the reference layer's own bf16-autocast distance."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_train_blocks import TDT, _close, _inside_autocast_yardstick, _rel_l2
from tests.test_train_blocks_host import _bn, _to_nchw, _to_nhwc
from tests.test_train_tinyvit_host import _attention_ref, _attn_core, _block_ref, _block_sd

pytestmark = pytest.mark.gpu


def _dev(p):
    return {k: v.float().cuda().contiguous() for k, v in p.items()}


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("M,C", [(1000, 128), (77, 160), (5, 320), (40000, 448), (3, 576), (6000, 1024), (130, 8)])
def test_layernorm_forward_backward(mode, M, C):
    from efficientsam3_amd import train_tinyvit as tt
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 2.0 + 0.5).to(TDT[mode])
    x[0] = 0                                                      # a padding token: LayerNorm(0) = its bias, rstd = eps ** -0.5
    dy = torch.randn(M, C, generator=g).to(TDT[mode])
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    xr, gr, br = x.float().clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gr, br, 1e-5)
    yr.backward(dy.float())
    y, mean, rstd = tt.layernorm_forward(x.cuda(), gamma.cuda(), beta.cuda())
    dx, dgamma, dbeta = tt.layernorm_backward(x.cuda(), dy.cuda(), gamma.cuda(), mean, rstd)
    assert torch.allclose(y[0].float().cpu(), beta, atol=1e-2 if mode == "bf16" else 1e-6)
    _close(y, yr.detach(), mode, "ln y", f32=3e-6, bf16=8e-3)
    _close(mean, x.float().mean(1), "f32", "ln mean", f32=3e-6)
    # dx of the all-zero row is rstd = 316 times larger than the others': compare it apart
    _close(dx[1:], xr.grad[1:], mode, "ln dx", f32=2e-5, bf16=8e-3)
    _close(dx[:1], xr.grad[:1], mode, "ln dx (zero row)", f32=2e-5, bf16=8e-3)
    _close(dgamma, gr.grad, "f32", "ln dgamma", f32=2e-5 if mode == "f32" else 2e-5)
    _close(dbeta, br.grad, "f32", "ln dbeta", f32=2e-5)
    dx2, dg2, db2 = tt.layernorm_backward(x.cuda(), dy.cuda(), gamma.cuda(), mean, rstd)
    assert torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)       # fixed summation order


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("nw,ws,heads", [(5, 7, 2), (3, 14, 5), (40, 7, 10), (2, 14, 18), (1, 7, 1)])
def test_window_attention_kernels_vs_autograd(mode, nw, ws, heads):
    """softmax(q k^T scale + bias) v per window and head: output, log-sum-exp, d(qkv), the bias gradient summed over the windows, and the
    adjoint of the attention_bias_idxs gather"""
    from efficientsam3_amd import train_tinyvit as tt
    n = ws * ws
    g = torch.Generator().manual_seed(nw * 100 + ws + heads)
    qkv = (torch.randn(nw, n, heads * 96, generator=g) * 1.5).to(TDT[mode])
    dout = torch.randn(nw, n, heads * 32, generator=g).to(TDT[mode])
    biases = torch.randn(heads, n, generator=g)
    idxs = torch.from_numpy(tt.attention_bias_idxs(ws))
    qr, br = qkv.float().clone().requires_grad_(True), biases.clone().requires_grad_(True)
    out_r, lse_r = _attn_core(qr, br[:, idxs], heads, 32 ** -0.5)
    out_r.backward(dout.float())
    ab = tt.attn_bias_gather(biases.cuda(), ws)
    assert torch.equal(ab.cpu(), biases[:, idxs])
    out, lse = tt.win_attn_forward(qkv.cuda(), ab, heads, 32 ** -0.5)
    _close(out, out_r.detach(), mode, "attention out", f32=2e-5, bf16=8e-3)
    _close(lse, lse_r.detach(), "f32", "log-sum-exp", f32=2e-5)
    dqkv, dbias_full = tt.win_attn_backward(qkv.cuda(), ab, out, lse, dout.cuda(), heads, 32 ** -0.5)
    dbias = tt.attn_bias_grad(dbias_full, ws)
    if mode == "f32":
        _close(dqkv, qr.grad, mode, "dqkv", f32=5e-5)
        _close(dbias, br.grad, mode, "d attention_biases", f32=5e-5)
    else:   # D_i = dO_i . O_i uses the bf16-rounded output: an error of 2^-9 |dO| |O| on every logit gradient of the row
        assert _rel_l2(dqkv, qr.grad) <= 1.5e-2 and _rel_l2(dbias, br.grad) <= 1.5e-2, (_rel_l2(dqkv, qr.grad), _rel_l2(dbias, br.grad))
    # the form that takes the attention_biases parameter itself (indexed from LDS by the bf16 kernels): the same results as with the gathered table
    out_t, lse_t = tt.win_attn_forward(qkv.cuda(), ab, heads, 32 ** -0.5, tab=biases.cuda().contiguous(), ws=ws)
    dqkv_t, dbias_full_t = tt.win_attn_backward(qkv.cuda(), ab, out, lse, dout.cuda(), heads, 32 ** -0.5, tab=biases.cuda().contiguous(), ws=ws)
    assert torch.equal(out_t, out) and torch.equal(lse_t, lse) and torch.equal(dqkv_t, dqkv) and torch.equal(dbias_full_t, dbias_full)
    if nw >= 3:   # the chunked walk over the windows (bounded logits-gradient tensor): same d(qkv) bit for bit, the bias gradient within the
        old = tt.DS_CHUNK_BYTES                                           # rounding of another summation order
        tt.DS_CHUNK_BYTES = 2 * heads * n * n * 4                        # two windows per chunk, a ragged last chunk for odd counts
        try:
            dqkv2, dbias_full2 = tt.win_attn_backward(qkv.cuda(), ab, out, lse, dout.cuda(), heads, 32 ** -0.5)
        finally:
            tt.DS_CHUNK_BYTES = old
        assert torch.equal(dqkv2, dqkv)
        _close(dbias_full2, dbias_full.cpu(), "f32", "chunked bias gradient", f32=2e-5)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("H,W,ws,C,heads", [(9, 11, 7, 64, 2), (7, 7, 7, 128, 4), (16, 15, 14, 160, 5), (8, 8, 7, 320, 10)])
def test_tinyvit_block_forward_backward_vs_autograd(mode, H, W, ws, C, heads):
    """One TinyViTBlock (tiny_vit.py:296-386) in TRAINING mode: zero padding to whole windows, attention, crop, residual with DropPath factors,
    depthwise 3x3 + BatchNorm, residual MLP -- output, input gradient, every parameter gradient under its state-dict name"""
    from efficientsam3_amd import train_tinyvit as tt
    B, base = 3, "layers.1.blocks.0"
    sd = _block_sd(C, heads, ws, H * W + C)
    g = torch.Generator().manual_seed(3)
    x, dy = torch.randn(B, H, W, C, generator=g).to(TDT[mode]), torch.randn(B, H, W, C, generator=g).to(TDT[mode])
    factors = {0: torch.tensor([1.25, 0.0, 1.25]), 1: torch.tensor([0.0, 1.25, 1.25])}

    def reference(amp):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.float().clone().requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            yr = _block_ref(xr, leaves, base, heads, ws, factors[0], factors[1])
        yr.float().backward(dy.float())
        out = {k: v.grad for k, v in leaves.items()}
        out.update(y=yr.detach().float(), dx=xr.grad)
        return out

    r32 = reference(False)
    dsd = _dev(sd)
    blk = tt.TinyViTBlockTrain(lambda k: dsd[k], lambda k: k in dsd, base, heads, ws, 0.2, lambda name, call, b, keep: factors[call])
    y = blk.forward(x.cuda().contiguous())
    dx, grads = blk.backward(dy.cuda().contiguous())
    assert sorted(grads) == sorted(sd)
    pairs = [(y, r32["y"], "y"), (dx, r32["dx"], "dx")] + [(grads[k].reshape(r32[k].shape), r32[k], k) for k in sd]
    if mode == "f32":
        worst = max((_rel_l2(got, ref), what) for got, ref, what in pairs)
        print(f"[TinyViT block {H}x{W} ws{ws} C={C} f32] worst relative L2 error {worst[0]:.2e} ({worst[1]})")
        for got, ref, what in pairs:
            _close(got, ref, mode, what, 5e-4)
    else:
        _inside_autocast_yardstick(pairs, reference(True), f"TinyViT block {H}x{W} ws{ws} C={C}")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("kind", ["mbconv", "merging"])
def test_tinyvit_conv_layers_forward_backward_vs_autograd(mode, kind):
    """MBConv (GELU after the shortcut, a DropPath factor on the branch; tiny_vit.py:87-125) and PatchMerging (stride-2 depthwise; :128-154)"""
    from efficientsam3_amd import train_tinyvit as tt
    B, H, W, C = 3, 10, 9, 64
    Cm, Co = (4 * C, C) if kind == "mbconv" else (128, 128)
    base = "layers.0.blocks.1" if kind == "mbconv" else "layers.0.downsample"
    g = torch.Generator().manual_seed(5)
    r = lambda *s, k=0.3: torch.randn(*s, generator=g) * k  # noqa: E731
    pos = lambda n: torch.rand(n, generator=g) + 0.5  # noqa: E731
    sd = {f"{base}.conv1.c.weight": r(Cm, C, 1, 1, k=C ** -0.5), f"{base}.conv1.bn.weight": pos(Cm), f"{base}.conv1.bn.bias": r(Cm),
          f"{base}.conv2.c.weight": r(Cm, 1, 3, 3), f"{base}.conv2.bn.weight": pos(Cm), f"{base}.conv2.bn.bias": r(Cm),
          f"{base}.conv3.c.weight": r(Co, Cm, 1, 1, k=Cm ** -0.5), f"{base}.conv3.bn.weight": pos(Co), f"{base}.conv3.bn.bias": r(Co)}
    stride = 1 if kind == "mbconv" else 2
    x = torch.randn(B, H, W, C, generator=g).to(TDT[mode])
    dy = torch.randn(B, (H + stride - 1) // stride, (W + stride - 1) // stride, Co, generator=g).to(TDT[mode])
    factor = torch.tensor([1.25, 0.0, 1.25])

    def reference(amp):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = _to_nchw(x.float()).requires_grad_(True)
        cb = lambda t, n, **kw: _bn(F.conv2d(t, leaves[f"{base}.{n}.c.weight"], None, **kw), leaves[f"{base}.{n}.bn.weight"], leaves[f"{base}.{n}.bn.bias"])  # noqa: E731
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            h = cb(F.gelu(cb(F.gelu(cb(xr, "conv1")), "conv2", stride=stride, padding=1, groups=Cm)), "conv3")
            yr = F.gelu(xr + h * factor.view(B, 1, 1, 1)) if kind == "mbconv" else h
        yr.float().backward(_to_nchw(dy.float()))
        out = {k: v.grad for k, v in leaves.items()}
        out.update(y=_to_nhwc(yr.detach().float()), dx=_to_nhwc(xr.grad))
        return out

    r32 = reference(False)
    dsd = _dev(sd)
    get, has = (lambda k: dsd[k]), (lambda k: k in dsd)
    layer = tt.MBConvTrain(get, has, base, 0.2, lambda name, call, b, keep: factor) if kind == "mbconv" else tt.PatchMergingTrain(get, has, base)
    y = layer.forward(x.cuda().contiguous())
    dx, grads = layer.backward(dy.cuda().contiguous())
    assert sorted(grads) == sorted(sd)
    pairs = [(y, r32["y"], "y"), (dx, r32["dx"], "dx")] + [(grads[k].reshape(r32[k].shape), r32[k], k) for k in sd]
    if mode == "f32":
        for got, ref, what in pairs:
            _close(got, ref, mode, what, 5e-4)
    else:
        _inside_autocast_yardstick(pairs, reference(True), f"TinyViT {kind}")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,C,ws", [(2, 9, 11, 64, 7), (1, 14, 14, 32, 7), (3, 16, 15, 160, 14), (2, 63, 63, 8, 14)])
def test_window_partition_kernel_equals_the_torch_formulation(mode, B, H, W, C, ws):
    """esam3_window_partition (tiny_vit.py:350-374 as one copy each way) against F.pad + the transposed reshape, and its inverse against the
    transposed reshape + the slice: bit for bit, zeros in the padding"""
    from efficientsam3_amd import _lib
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    x = torch.randn(B, H, W, C, generator=g).to(TDT[mode]).cuda()
    pad_b, pad_r = (ws - H % ws) % ws, (ws - W % ws) % ws
    ph, pw = H + pad_b, W + pad_r
    ref = F.pad(x, (0, 0, 0, pad_r, 0, pad_b)).view(B, ph // ws, ws, pw // ws, ws, C).transpose(2, 3).reshape(B * (ph // ws) * (pw // ws), ws * ws, C).contiguous()
    out = torch.full_like(ref, 7.0)
    dt = 0 if mode == "f32" else 1
    _lib.check(_lib.load().esam3_window_partition(dt, x.data_ptr(), out.data_ptr(), B, H, W, C, ws, 0, torch.cuda.current_stream().cuda_stream), "partition")
    assert torch.equal(out, ref)
    back = torch.full_like(x, 7.0)
    _lib.check(_lib.load().esam3_window_partition(dt, out.data_ptr(), back.data_ptr(), B, H, W, C, ws, 1, torch.cuda.current_stream().cuda_stream), "reverse")
    assert torch.equal(back, x)
