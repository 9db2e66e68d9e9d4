"""CPU: the COMPOSITION logic of efficientsam3_amd/train_tinyvit.py (window attention with its gathered bias table, zero-padding to whole
windows and the partition / reverse / crop around it, DropPath factors, the MBConv and PatchMerging forms, the whole trunk with its
state-dict names) with every kernel wrapper replaced by a plain torch stand-in of the same contract, against torch.autograd of the same
layers written with torch functions (sam3/backbones/tiny_vit.py:67-154,196-386) and, where the reference tree is present, against the real
module.  The kernels are checked on the GPU by tests/test_train_tinyvit.py, whole training steps against the reference's own runs by
tests/test_stage1_step.py (GPU) and tests/test_stage1_trainer_host.py (CPU)."""
import os

import pytest
import torch
import torch.nn.functional as F

from efficientsam3_amd import schema
from efficientsam3_amd import train_blocks as tb
from efficientsam3_amd import train_tinyvit as tt
from tests.test_train_blocks_host import _bn, _check, _to_nchw, _to_nhwc, cpu_kernels  # noqa: F401
from tests.test_train_repvit_host import repvit_kernels  # noqa: F401


def _attn_core(qkv, bias, heads, scale, tab=None, ws=0):   # tab / ws: the compact-table form of the device kernels (the stand-in reads `bias`)
    nw, n, _ = qkv.shape
    q, k, v = qkv.view(nw, n, heads, 96).split([32, 32, 32], dim=3)
    q, k, v = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
    s = (q @ k.transpose(-2, -1)) * scale + bias
    return (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(nw, n, heads * 32), torch.logsumexp(s, dim=-1)


def install_tinyvit_kernels(monkeypatch):  # noqa: F811
    def ln_fwd(x, gamma, beta, eps=1e-5):
        mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + eps)
        return (x - mean) * rstd * gamma + beta, mean.reshape(-1), rstd.reshape(-1)

    def ln_bwd(x, dy, gamma, mean, rstd):
        c = x.shape[-1]
        xh = (x - mean.reshape(x.shape[:-1] + (1,))) * rstd.reshape(x.shape[:-1] + (1,))
        g = dy * gamma
        dx = rstd.reshape(x.shape[:-1] + (1,)) * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
        return dx, (dy * xh).reshape(-1, c).sum(0), dy.reshape(-1, c).sum(0)

    def attn_bwd(qkv, bias, out, lse, dout, heads, scale, tab=None, ws=0):
        qr, br = qkv.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        o, _ = _attn_core(qr, br, heads, scale)
        o.backward(dout)
        return qr.grad, br.grad

    def bias_grad(dbias_full, ws):
        idxs = torch.from_numpy(tt.attention_bias_idxs(ws)).reshape(-1)
        return torch.zeros(dbias_full.shape[0], ws * ws).index_add_(1, idxs, dbias_full.reshape(dbias_full.shape[0], -1))

    monkeypatch.setattr(tt, "layernorm_forward", ln_fwd)
    monkeypatch.setattr(tt, "layernorm_backward", ln_bwd)
    monkeypatch.setattr(tt, "win_attn_forward", _attn_core)
    monkeypatch.setattr(tt, "win_attn_backward", attn_bwd)
    monkeypatch.setattr(tt, "attn_bias_grad", bias_grad)


@pytest.fixture
def tinyvit_kernels(repvit_kernels, monkeypatch):  # noqa: F811
    install_tinyvit_kernels(monkeypatch)


def _ln(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def _attention_ref(xw, p, heads, ws):
    """Attention.forward (tiny_vit.py:265-293) on window rows"""
    idxs = torch.from_numpy(tt.attention_bias_idxs(ws))
    qkv = F.linear(_ln(xw, p["norm.weight"], p["norm.bias"]), p["qkv.weight"], p["qkv.bias"])
    out, _ = _attn_core(qkv, p["attention_biases"][:, idxs], heads, 32 ** -0.5)
    return F.linear(out, p["proj.weight"], p["proj.bias"])


def _block_ref(x, sd, base, heads, ws, f1, f2):
    """TinyViTBlock.forward (tiny_vit.py:344-386) on [B, H, W, C]; f1 / f2 = the DropPath factors [B] of its two residual branches"""
    b, h, w, c = x.shape
    pad_b, pad_r = (ws - h % ws) % ws, (ws - w % ws) % ws
    xp = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    ph, pw = h + pad_b, w + pad_r
    xw = xp.view(b, ph // ws, ws, pw // ws, ws, c).transpose(2, 3).reshape(-1, ws * ws, c)
    a = _attention_ref(xw, {k: sd[f"{base}.attn.{k}"] for k in ("norm.weight", "norm.bias", "qkv.weight", "qkv.bias", "proj.weight", "proj.bias",
                                                               "attention_biases")}, heads, ws)
    a = a.view(b, ph // ws, pw // ws, ws, ws, c).transpose(2, 3).reshape(b, ph, pw, c)[:, :h, :w]
    x1 = x + a * f1.view(b, 1, 1, 1)
    x2 = _to_nhwc(_bn(F.conv2d(_to_nchw(x1), sd[base + ".local_conv.c.weight"], None, padding=1, groups=c), sd[base + ".local_conv.bn.weight"],
                      sd[base + ".local_conv.bn.bias"]))
    m = F.linear(F.gelu(F.linear(_ln(x2, sd[base + ".mlp.norm.weight"], sd[base + ".mlp.norm.bias"]), sd[base + ".mlp.fc1.weight"], sd[base + ".mlp.fc1.bias"])),
                 sd[base + ".mlp.fc2.weight"], sd[base + ".mlp.fc2.bias"])
    return x2 + m * f2.view(b, 1, 1, 1)


def _block_sd(C, heads, ws, seed, base="layers.1.blocks.0"):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, k=0.3: torch.randn(*s, generator=g) * k  # noqa: E731
    pos = lambda n: torch.rand(n, generator=g) + 0.5  # noqa: E731
    return {f"{base}.attn.norm.weight": pos(C), f"{base}.attn.norm.bias": r(C), f"{base}.attn.qkv.weight": r(3 * C, C, k=C ** -0.5),
            f"{base}.attn.qkv.bias": r(3 * C), f"{base}.attn.proj.weight": r(C, C, k=C ** -0.5), f"{base}.attn.proj.bias": r(C),
            f"{base}.attn.attention_biases": r(heads, ws * ws, k=1.0), f"{base}.local_conv.c.weight": r(C, 1, 3, 3),
            f"{base}.local_conv.bn.weight": pos(C), f"{base}.local_conv.bn.bias": r(C), f"{base}.mlp.norm.weight": pos(C), f"{base}.mlp.norm.bias": r(C),
            f"{base}.mlp.fc1.weight": r(4 * C, C, k=C ** -0.5), f"{base}.mlp.fc1.bias": r(4 * C), f"{base}.mlp.fc2.weight": r(C, 4 * C, k=(4 * C) ** -0.5),
            f"{base}.mlp.fc2.bias": r(C)}


def test_attention_bias_idxs_table():
    """the offsets table: symmetric in (i, j) (the kernels read a row of the gathered bias as a column), ws * ws distinct offsets, and the
    first row numbers them in order of appearance (tiny_vit.py:240-251)"""
    for ws in (7, 14):
        idx = tt.attention_bias_idxs(ws)
        assert idx.shape == (ws * ws, ws * ws) and (idx == idx.T).all() and idx.max() == ws * ws - 1
        assert (idx[0] == range(ws * ws)).all() and (idx.diagonal() == 0).all()
        _, start, items = tt._bias_tables(ws, "cpu")
        assert int(start[-1]) == ws ** 4 and sorted(items.tolist()) == list(range(ws ** 4))
        flat = idx.reshape(-1)
        for o in (0, 1, ws * ws - 1):
            assert all(flat[i] == o for i in items[int(start[o]):int(start[o + 1])].tolist())


def test_window_attention_composition(tinyvit_kernels):
    C, heads, ws, nw = 64, 2, 7, 5
    sd = _block_sd(C, heads, ws, 1)
    p = {k[len("layers.1.blocks.0.attn."):]: v for k, v in sd.items() if ".attn." in k}
    g = torch.Generator().manual_seed(2)
    xw, dy = torch.randn(nw, ws * ws, C, generator=g), torch.randn(nw, ws * ws, C, generator=g)
    rp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = xw.clone().requires_grad_(True)
    yr = _attention_ref(xr, rp, heads, ws)
    yr.backward(dy)
    layer = tt.WindowAttentionTrain(p, heads, ws)
    y = layer.forward(xw)
    dx, grads = layer.backward(dy)
    _check([(y, yr.detach(), "y"), (dx, xr.grad, "dx")] + [(grads[k].reshape(rp[k].shape), rp[k].grad, k) for k in p], tol=1e-3)


@pytest.mark.parametrize("H,W,ws", [(9, 11, 7), (7, 7, 7), (14, 7, 7), (16, 15, 14)])
def test_tinyvit_block_composition_with_padding_and_drop_path(tinyvit_kernels, H, W, ws):
    B, C, heads = 3, 64, 2
    base = "layers.1.blocks.0"
    sd = _block_sd(C, heads, ws, H * W)
    g = torch.Generator().manual_seed(3)
    x, dy = torch.randn(B, H, W, C, generator=g), torch.randn(B, H, W, C, generator=g)
    factors = {0: torch.tensor([1.25, 0.0, 1.25]), 1: torch.tensor([0.0, 1.25, 1.25])}          # keep = 0.8: one sample drops each branch
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = _block_ref(xr, leaves, base, heads, ws, factors[0], factors[1])
    yr.backward(dy)
    blk = tt.TinyViTBlockTrain(lambda k: sd[k], lambda k: k in sd, base, heads, ws, 0.2, lambda name, call, b, keep: factors[call])
    y = blk.forward(x)
    dx, grads = blk.backward(dy)
    assert sorted(grads) == sorted(sd)
    _check([(y, yr.detach(), "y"), (dx, xr.grad, "dx")] + [(grads[k].reshape(leaves[k].shape), leaves[k].grad, k) for k in sd], tol=2e-3)


def _trunk_ref(img, sd, name, factor_of):
    """TinyViTAdapter.forward (stage1/model.py:310-318) with torch functions on a dict of leaf tensors; ``factor_of(module name, call)`` -> [B]"""
    dims, depths, heads, windows = schema.TINYVIT_CFG[name]

    def conv_bn(x, base, stride=1, pad=0, groups=1):
        return _bn(F.conv2d(x, sd[base + ".c.weight"], None, stride=stride, padding=pad, groups=groups), sd[base + ".bn.weight"], sd[base + ".bn.bias"])

    b = img.shape[0]
    x = conv_bn(F.gelu(conv_bn(img, "patch_embed.seq.0", 2, 1)), "patch_embed.seq.2", 2, 1)
    for li, depth in enumerate(depths):
        for bi in range(depth):
            base = f"layers.{li}.blocks.{bi}"
            if li == 0:
                c = x.shape[1] * 4
                h = conv_bn(F.gelu(conv_bn(F.gelu(conv_bn(x, base + ".conv1")), base + ".conv2", 1, 1, groups=c)), base + ".conv3")
                x = F.gelu(x + h * factor_of(base + ".drop_path", 0).view(b, 1, 1, 1))
            else:
                x = _to_nchw(_block_ref(_to_nhwc(x), sd, base, heads[li], windows[li], factor_of(base + ".drop_path", 0), factor_of(base + ".drop_path", 1)))
        if li < len(depths) - 1:
            q = f"layers.{li}.downsample"
            c = dims[li + 1]
            x = conv_bn(F.gelu(conv_bn(F.gelu(conv_bn(x, q + ".conv1")), q + ".conv2", 2, 1, groups=c)), q + ".conv3")
    return x


def test_tinyvit_trunk_composition_and_names(tinyvit_kernels):
    """the whole 11m trunk (stochastic depth: seeded factors shared by both sides) at a small image: every parameter gets a gradient under its
    own name and shape, equal to autograd's; the last block's map comes back NHWC"""
    full = schema.synthetic_state_dict("tinyvit", "11m", seed=3)
    pre = "backbone.vision_backbone.trunk.model.backbone.model."
    sd = {k[len(pre):]: v.float() for k, v in full.items() if k.startswith(pre)}
    params = {k: v for k, v in sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    g = torch.Generator().manual_seed(6)
    img = torch.randn(2, 3, 160, 128, generator=g)
    drawn = {}

    def factor_of(name, call, batch=2, keep=None):
        if (name, call) not in drawn:
            rate = {"layers.0.blocks.0.drop_path": 0.0}.get(name, 0.3)
            drawn[(name, call)] = torch.ones(batch) if rate == 0.0 else torch.empty(batch).bernoulli_(0.7, generator=g) / 0.7
        return drawn[(name, call)]

    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yr = _trunk_ref(img, leaves, "11m", factor_of)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    trunk = tt.TinyViTTrunkTrain({k: v.clone() for k, v in sd.items()}, "11m", dtype=torch.float32, drop_path_sampler=lambda n, c, b, keep: factor_of(n, c))
    y = trunk.forward(img)
    order = []
    grads = trunk.backward(_to_nhwc(dy), sink=lambda n, gv: order.append(n))
    assert sorted(grads) == sorted(params) and sorted(order) == sorted(params)
    assert order[0].startswith("layers.3.blocks.1.") and order[-1].startswith("patch_embed.seq.0.")
    d, m = float((y - _to_nhwc(yr.detach())).abs().max()), float(yr.detach().abs().max())
    assert d <= 1e-3 * m, (d, m)
    typical = float(torch.stack([leaves[k].grad.abs().max() for k in params]).median())
    worst = max((float((grads[k] - leaves[k].grad).abs().max()) / max(float(leaves[k].grad.abs().max()), 1e-2 * typical), k) for k in params)
    print("worst relative gradient error", worst, "typical gradient", typical)
    assert worst[0] <= 5e-3, worst
    for k in params:
        assert tuple(grads[k].shape) == tuple(params[k].shape), k
    norms = dict(trunk.norm_layers())
    assert set(norms) == {k[:-len(".running_mean")] for k in sd if k.endswith("running_mean")}


REFERENCE = "/root/reference"


@pytest.mark.parametrize("name", ["5m", "11m", "21m"])
@pytest.mark.skipif(not os.path.isdir(REFERENCE + "/sam3"), reason="the reference tree is only present in the build container")
def test_tinyvit_trunk_vs_the_reference_module(tinyvit_kernels, name):
    """where the reference is present (this container, never the GPU box): against the REAL module -- tiny_vit_5m_224(img_size=...) of
    sam3/backbones/tiny_vit.py run as TinyViTAdapter.forward does (stage1/model.py:310-318: patch_embed, the four layers, tokens back to a
    map), in train mode with drop_path_rate 0 (stochastic depth is covered by the training-step fixtures), loaded with the synthetic state dict; all three students"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pth in (os.path.join(root, "oracle", "shims"), REFERENCE + "/sam3"):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    from sam3.backbones import tiny_vit as ref_tiny_vit
    H, W = 160, 160
    model = getattr(ref_tiny_vit, f"tiny_vit_{name}_224")(pretrained=False, img_size=H, drop_path_rate=0.0)   # stochastic depth: the fixture tests
    model.head, model.norm_head = torch.nn.Identity(), torch.nn.Identity()
    full = schema.synthetic_state_dict("tinyvit", name, seed=4)
    pre = "backbone.vision_backbone.trunk.model.backbone.model."
    sd = {k[len(pre):]: v.float() for k, v in full.items() if k.startswith(pre)}
    model.load_state_dict(sd, strict=True)
    model.train()
    g = torch.Generator().manual_seed(8)
    img = torch.randn(2, 3, H, W, generator=g)
    x = model.patch_embed(img)
    for layer in model.layers:
        x = layer(x)
    fh = H // 4
    for _ in range(3):
        fh = (fh - 1) // 2 + 1
    yr = x.view(2, fh, fh, -1)                                   # NHWC of the adapter's NCHW output
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    ref_grads = {n: p.grad for n, p in model.named_parameters()}
    trunk = tt.TinyViTTrunkTrain({k: v.clone() for k, v in sd.items()}, name, dtype=torch.float32, drop_path_sampler=lambda n, c, b, keep: torch.ones(b))
    y = trunk.forward(img)
    grads = trunk.backward(dy.contiguous())
    assert sorted(grads) == sorted(ref_grads)
    d, m = float((y - yr.detach()).abs().max()), float(yr.detach().abs().max())
    assert d <= 1e-3 * m, (d, m)
    typical = float(torch.stack([v.abs().max() for v in ref_grads.values()]).median())
    worst = max((float((grads[k] - ref_grads[k]).abs().max()) / max(float(ref_grads[k].abs().max()), 1e-2 * typical), k) for k in grads)
    print("worst relative gradient error against the reference module", worst)
    assert worst[0] <= 5e-3, worst
    ref_buf = {k: v for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    for name, layer in trunk.norm_layers():
        for stat in ("running_mean", "running_var"):
            got, want = getattr(layer, stat), ref_buf[f"{name}.{stat}"]
            assert float((got - want).abs().max()) <= 1e-4 * max(float(want.abs().max()), 1.0), (name, stat)
