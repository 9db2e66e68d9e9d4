"""ORACLE (test infrastructure only) -- CPU fp32 restatement of the EfficientSAM3
image hot path: ``Sam3Processor.set_image`` + ``Sam3Image.predict_inst``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package never does.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference
(``/root/reference/sam3`` through ``oracle/shims``) in the build container,
loads the same seeded state dict (``efficientsam3_amd.schema``) into it and
checks every stage boundary of this restatement against the reference's own
tensors (max-abs-err is recorded in ``tests/golden/manifest.json``); the
reference's outputs are committed as fixtures under ``tests/golden/``.
Third-party arithmetic that is *not* pinned (absent from /root/reference):
torchvision's uint8 antialiased ``v2.Resize`` (inputs here are already
1008x1008 so it short-circuits) and skimage ``label`` (restated with
scipy.ndimage 8-connectivity).

Written as plain functions over a ``state_dict`` with the reference's key names;
each function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

TRUNK = "backbone.vision_backbone.trunk.model."
EV_BB = TRUNK + "backbone.model."
NECK = "backbone.vision_backbone."
SAM = "inst_interactive_predictor.model."

IMG = 1008
EMB = 72
BN_EPS = 1e-5  # nn.BatchNorm2d default
EV_CFG = {
    "b0": ([8, 16, 32, 64, 128], [1, 2, 2, 2, 2], 16),
    "b1": ([16, 32, 64, 128, 256], [1, 2, 3, 3, 4], 16),
    "b2": ([24, 48, 96, 192, 384], [1, 3, 4, 4, 6], 32),
}


# --------------------------------------------------------------------------
# EfficientViT building blocks  (backbones/efficientvit/nn/ops.py)
# --------------------------------------------------------------------------
def _act(x: torch.Tensor, act: Optional[str]) -> torch.Tensor:
    if act is None:
        return x
    if act == "hswish":  # act.py:15 nn.Hardswish
        return F.hardswish(x)
    if act == "gelu":  # exact erf GELU (nn.GELU() default) -- model_builder.py:773, necks.py:47
        return F.gelu(x)
    if act == "relu":
        return F.relu(x)
    raise ValueError(act)


def conv_layer(sd: SD, p: str, x: torch.Tensor, stride: int = 1, groups: int = 1,
               act: Optional[str] = None) -> torch.Tensor:
    """ConvLayer.forward: conv -> (BN) -> (act)   ops.py:39-80."""
    w = sd[p + ".conv.weight"]
    b = sd.get(p + ".conv.bias")
    x = F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2, groups=groups)
    if p + ".norm.weight" in sd:
        x = F.batch_norm(x, sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"],
                         sd[p + ".norm.weight"], sd[p + ".norm.bias"], False, 0.0, BN_EPS)
    return _act(x, act)


def dsconv(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """DSConv (ops.py:273-312), act=(hswish, None) from backbone.py:131-139."""
    c = x.shape[1]
    x = conv_layer(sd, p + "depth_conv", x, groups=c, act="hswish")
    return conv_layer(sd, p + "point_conv", x)


def mbconv(sd: SD, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """MBConv (ops.py:315-367), act=(hswish, hswish, None) from backbone.py:141-149."""
    x = conv_layer(sd, p + "inverted_conv", x, act="hswish")
    x = conv_layer(sd, p + "depth_conv", x, stride=stride, groups=x.shape[1], act="hswish")
    return conv_layer(sd, p + "point_conv", x)


def lite_mla(sd: SD, p: str, x: torch.Tensor, dim: int) -> torch.Tensor:
    """LiteMLA.forward + relu_linear_att (ops.py:584-671), scales=(5,), eps=1e-15."""
    qkv = conv_layer(sd, p + "qkv", x)
    c3 = qkv.shape[1]
    w_dw = sd[p + "aggreg.0.0.weight"]
    w_pw = sd[p + "aggreg.0.1.weight"]
    agg = F.conv2d(qkv, w_dw, None, padding=w_dw.shape[-1] // 2, groups=c3)
    agg = F.conv2d(agg, w_pw, None, groups=c3 // w_pw.shape[1])
    ms = torch.cat([qkv, agg], dim=1)
    B, _, H, W = ms.shape
    assert H * W > dim  # linear-attention branch (ops.py:665)
    t = ms.reshape(B, -1, 3 * dim, H * W)
    q, k, v = t[:, :, :dim], t[:, :, dim:2 * dim], t[:, :, 2 * dim:]
    q, k = F.relu(q), F.relu(k)
    v1 = F.pad(v, (0, 0, 0, 1), mode="constant", value=1.0)
    vk = torch.matmul(v1, k.transpose(-1, -2))
    out = torch.matmul(vk, q)
    out = out[:, :, :-1] / (out[:, :, -1:] + 1e-15)
    out = out.reshape(B, -1, H, W)
    return conv_layer(sd, p + "proj", out)


def evit_block(sd: SD, p: str, x: torch.Tensor, dim: int) -> torch.Tensor:
    """EfficientViTBlock = Residual(LiteMLA) -> Residual(MBConv)  ops.py:674-733."""
    x = x + lite_mla(sd, p + "context_module.main.", x, dim)
    return x + mbconv(sd, p + "local_module.main.", x, 1)


def efficientvit_backbone(sd: SD, x: torch.Tensor, model_name: str = "b1",
                          taps: Optional[dict] = None) -> torch.Tensor:
    """EfficientViTBackbone.forward -> 'stage_final' (backbone.py:150-156)."""
    widths, depths, dim = EV_CFG[model_name]
    p = EV_BB
    x = conv_layer(sd, p + "input_stem.op_list.0", x, stride=2, act="hswish")
    for i in range(depths[0]):
        x = x + dsconv(sd, p + f"input_stem.op_list.{i + 1}.main.", x)
    if taps is not None:
        taps["stage0"] = x
    for si, d in enumerate(depths[1:3]):
        for i in range(d):
            y = mbconv(sd, p + f"stages.{si}.op_list.{i}.main.", x, 2 if i == 0 else 1)
            x = y if i == 0 else x + y
        if taps is not None:
            taps[f"stage{si + 1}"] = x
    for si, d in enumerate(depths[3:], start=2):
        x = mbconv(sd, p + f"stages.{si}.op_list.0.main.", x, 2)
        for i in range(d):
            x = evit_block(sd, p + f"stages.{si}.op_list.{i + 1}.", x, dim)
        if taps is not None:
            taps[f"stage{si + 1}"] = x
    return x


# --------------------------------------------------------------------------
# RepViT (backbones/repvit.py)
# --------------------------------------------------------------------------
def _conv_bn(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 0,
             groups: int = 1) -> torch.Tensor:
    """Conv2d_BN (repvit.py:29-37): bias-free conv followed by BatchNorm2d (eval)."""
    x = F.conv2d(x, sd[p + ".c.weight"], None, stride, padding, 1, groups)
    return F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"],
                        sd[p + ".bn.weight"], sd[p + ".bn.bias"], False, 0.0, BN_EPS)


def squeeze_excite(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """timm.layers.SqueezeExcite (timm >= 1.0.17, not vendored in the reference): global mean ->
    1x1 fc1 -> ReLU -> 1x1 fc2 -> sigmoid gate (used at repvit.py:136,150)."""
    g = x.mean((2, 3), keepdim=True)
    g = F.relu(F.conv2d(g, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    g = F.conv2d(g, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    return x * torch.sigmoid(g)


def repvit_block(sd: SD, p: str, x: torch.Tensor, cfg) -> torch.Tensor:
    """RepViTBlock.forward (repvit.py:125-161), un-fused (training-time) parameterisation."""
    k, _t, _c, use_se, _use_hs, stride = cfg
    cin = x.shape[1]
    if stride == 2:
        x = _conv_bn(sd, p + "token_mixer.0", x, 2, (k - 1) // 2, groups=cin)
        if use_se:
            x = squeeze_excite(sd, p + "token_mixer.1.", x)
        x = _conv_bn(sd, p + "token_mixer.2", x)
    else:
        q = p + "token_mixer.0."  # RepVGGDW (repvit.py:84-93)
        y = (_conv_bn(sd, q + "conv", x, 1, 1, groups=cin)
             + F.conv2d(x, sd[q + "conv1.weight"], sd[q + "conv1.bias"], groups=cin)) + x
        x = F.batch_norm(y, sd[q + "bn.running_mean"], sd[q + "bn.running_var"], sd[q + "bn.weight"],
                         sd[q + "bn.bias"], False, 0.0, BN_EPS)
        if use_se:
            x = squeeze_excite(sd, p + "token_mixer.1.", x)
    m = F.gelu(_conv_bn(sd, p + "channel_mixer.m.0", x))
    return x + _conv_bn(sd, p + "channel_mixer.m.2", m)


def repvit_backbone(sd: SD, x: torch.Tensor, model_name: str = "m1.1",
                    taps: Optional[dict] = None) -> torch.Tensor:
    """RepViTTrunkWrapper.forward: all of model.features (model_builder.py:862-865, repvit.py:232-252)."""
    from efficientsam3_amd.schema import REPVIT_CFG
    cfgs = REPVIT_CFG[model_name.replace("_", ".")]
    p = EV_BB + "features."
    x = F.gelu(_conv_bn(sd, p + "0.0", x, 2, 1))
    x = _conv_bn(sd, p + "0.2", x, 2, 1)
    stage = 0
    for i, cfg in enumerate(cfgs, start=1):
        if cfg[5] == 2:
            if taps is not None:
                taps[f"stage{stage}"] = x
            stage += 1
        x = repvit_block(sd, p + f"{i}.", x, cfg)
    if taps is not None:
        taps[f"stage{stage}"] = x
    return x


# --------------------------------------------------------------------------
# TinyViT (backbones/tiny_vit.py), built with img_size=1008, num_classes=0
# --------------------------------------------------------------------------
def _tv_mbconv(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """MBConv.forward (tiny_vit.py:87-125): the GELU comes AFTER the shortcut add."""
    y = F.gelu(_conv_bn(sd, p + "conv1", x))
    y = F.gelu(_conv_bn(sd, p + "conv2", y, 1, 1, groups=y.shape[1]))
    y = _conv_bn(sd, p + "conv3", y)
    return F.gelu(y + x)


def _tv_patch_merging(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """PatchMerging.forward on NCHW (tiny_vit.py:128-154): 1x1, GELU, dw3x3 s2, GELU, 1x1."""
    x = F.gelu(_conv_bn(sd, p + "conv1", x))
    x = F.gelu(_conv_bn(sd, p + "conv2", x, 2, 1, groups=x.shape[1]))
    return _conv_bn(sd, p + "conv3", x)


def _tv_attention_bias_idxs(ws: int) -> torch.Tensor:
    """Attention.__init__ (tiny_vit.py:240-255): index of the (|dy|, |dx|) offset in order of first
    appearance, which for the row-major point list is dy * ws + dx."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    pts = torch.stack([ys.reshape(-1), xs.reshape(-1)], 1)
    d = (pts[:, None, :] - pts[None, :, :]).abs()
    return d[..., 0] * ws + d[..., 1]


def _tv_attention(sd: SD, p: str, x: torch.Tensor, heads: int, ws: int) -> torch.Tensor:
    """Attention.forward (tiny_vit.py:265-293) on windows x [Bw, N, C]; attn_ratio = 1."""
    bw, n, c = x.shape
    kd = c // heads
    x = F.layer_norm(x, (c,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).view(bw, n, heads, 3 * kd)
    q, k, v = qkv.split([kd, kd, kd], dim=3)
    q, k, v = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
    bias = sd[p + "attention_biases"][:, _tv_attention_bias_idxs(ws)]
    attn = (q @ k.transpose(-2, -1)) * (kd ** -0.5) + bias
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(bw, n, c)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def _tv_block(sd: SD, p: str, x: torch.Tensor, hw: int, heads: int, ws: int) -> torch.Tensor:
    """TinyViTBlock.forward (tiny_vit.py:339-380) on tokens [B, hw*hw, C]."""
    b, l, c = x.shape
    res = x
    y = x.view(b, hw, hw, c)
    pad = (ws - hw % ws) % ws
    if pad:
        y = F.pad(y, (0, 0, 0, pad, 0, pad))  # zero rows/cols BEFORE the attention's LayerNorm
    ph = hw + pad
    nw = ph // ws
    y = y.view(b, nw, ws, nw, ws, c).transpose(2, 3).reshape(b * nw * nw, ws * ws, c)
    y = _tv_attention(sd, p + "attn.", y, heads, ws)
    y = y.view(b, nw, nw, ws, ws, c).transpose(2, 3).reshape(b, ph, ph, c)
    if pad:
        y = y[:, :hw, :hw].contiguous()
    x = res + y.view(b, l, c)
    x = x.transpose(1, 2).reshape(b, c, hw, hw)
    x = _conv_bn(sd, p + "local_conv", x, 1, 1, groups=c)
    x = x.view(b, c, l).transpose(1, 2)
    m = F.layer_norm(x, (c,), sd[p + "mlp.norm.weight"], sd[p + "mlp.norm.bias"], 1e-5)
    m = F.gelu(F.linear(m, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(m, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def tinyvit_backbone(sd: SD, x: torch.Tensor, model_name: str = "11m",
                     taps: Optional[dict] = None) -> torch.Tensor:
    """TinyViTTrunkWrapper.forward (model_builder.py:883-896): patch_embed, layers, tokens -> NCHW."""
    from efficientsam3_amd.schema import TINYVIT_CFG
    dims, depths, heads, windows = TINYVIT_CFG[model_name]
    p = EV_BB
    x = F.gelu(_conv_bn(sd, p + "patch_embed.seq.0", x, 2, 1))
    x = _conv_bn(sd, p + "patch_embed.seq.2", x, 2, 1)
    if taps is not None:
        taps["stage0"] = x
    for bi in range(depths[0]):
        x = _tv_mbconv(sd, p + f"layers.0.blocks.{bi}.", x)
    x = _tv_patch_merging(sd, p + "layers.0.downsample.", x)
    if taps is not None:
        taps["stage1"] = x
    for li in range(1, len(dims)):
        b, c, hw, _ = x.shape
        t = x.flatten(2).transpose(1, 2)
        for bi in range(depths[li]):
            t = _tv_block(sd, p + f"layers.{li}.blocks.{bi}.", t, hw, heads[li], windows[li])
        x = t.view(b, hw, hw, c).permute(0, 3, 1, 2)
        if li < len(dims) - 1:
            x = _tv_patch_merging(sd, p + f"layers.{li}.downsample.", x)
        if taps is not None:
            taps[f"stage{li + 1}"] = x.contiguous()
    return x.contiguous()


# --------------------------------------------------------------------------
# ViT-H teacher trunk (model/vitdet.py as configured by model_builder.py:70-97)
# --------------------------------------------------------------------------
VIT_TRUNK = NECK + "trunk."


def _vit_rope_table(end: int, scale: float, head_dim: int = 64, theta: float = 10000.0):
    """compute_axial_cis (vitdet.py:41-57): angles [end*end, head_dim/2]; the first half of the
    complex pairs rotates with x, the second half with y."""
    f = 1.0 / (theta ** (torch.arange(0, head_dim, 4)[: head_dim // 4].float() / head_dim))
    t = torch.arange(end * end, dtype=torch.float32)
    tx = (t % end).float() * scale
    ty = torch.div(t, end, rounding_mode="floor").float() * scale
    return torch.cat([torch.outer(tx, f), torch.outer(ty, f)], dim=-1)


def _vit_apply_rope(x: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
    """apply_rotary_enc (vitdet.py:68-90): (x[2i] + i x[2i+1]) * exp(i ang[i]), fp32."""
    xr = x.float().reshape(*x.shape[:-1], -1, 2)
    c, s_ = torch.cos(ang), torch.sin(ang)
    out = torch.stack([xr[..., 0] * c - xr[..., 1] * s_, xr[..., 0] * s_ + xr[..., 1] * c], dim=-1)
    return out.flatten(-2).type_as(x)


def _vit_attention(sd: SD, p: str, x: torch.Tensor, heads: int, ang: torch.Tensor) -> torch.Tensor:
    """Attention.forward (vitdet.py:466-515) on [B', H, W, C] windows (or the full map)."""
    b, h, w, c = x.shape
    l = h * w
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(b, l, 3, heads, -1)
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
    q, k = _vit_apply_rope(q, ang), _vit_apply_rope(k, ang)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.view(b, heads, h, w, -1).permute(0, 2, 3, 1, 4).reshape(b, h, w, -1)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def vit_backbone(sd: SD, img: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """ViT.forward (vitdet.py:796-839): -> [B, 1024, 72, 72]."""
    from efficientsam3_amd.schema import VIT_CFG as C
    p = VIT_TRUNK
    d, heads, ws, g0 = C["embed_dim"], C["heads"], C["window"], C["pretrain_grid"]
    x = F.conv2d(img, sd[p + "patch_embed.proj.weight"], None, stride=C["patch"]).permute(0, 2, 3, 1)
    b, h, w, _ = x.shape
    pos = sd[p + "pos_embed"][:, 1:].reshape(1, g0, g0, d).permute(0, 3, 1, 2)  # drop cls, tile (get_abs_pos)
    pos = pos.tile([1, 1, h // g0 + 1, w // g0 + 1])[:, :, :h, :w].permute(0, 2, 3, 1)
    x = x + pos
    x = F.layer_norm(x, (d,), sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"], 1e-5)
    if taps is not None:
        taps["stage0"] = x.permute(0, 3, 1, 2)
    ang_win = _vit_rope_table(ws, 1.0)            # window blocks: rope over the 24x24 window
    ang_glob = _vit_rope_table(h, ws / h)         # global blocks: interpolated to the 24-px pre-training extent
    for i in range(C["depth"]):
        q = p + f"blocks.{i}."
        y = F.layer_norm(x, (d,), sd[q + "norm1.weight"], sd[q + "norm1.bias"], 1e-5)
        if i in C["global_blocks"]:
            y = _vit_attention(sd, q + "attn.", y, heads, ang_glob)
        else:
            nw = h // ws
            y = y.view(b, nw, ws, nw, ws, d).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, d)
            y = _vit_attention(sd, q + "attn.", y, heads, ang_win)
            y = y.view(b, nw, nw, ws, ws, d).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, d)
        x = x + y
        y = F.layer_norm(x, (d,), sd[q + "norm2.weight"], sd[q + "norm2.bias"], 1e-5)
        y = F.gelu(F.linear(y, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]))
        x = x + F.linear(y, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])
        if taps is not None and i in C["global_blocks"]:
            taps[f"stage{1 + C['global_blocks'].index(i)}"] = x.permute(0, 3, 1, 2)
    return x.permute(0, 3, 1, 2)


def backbone_family(model_name: str) -> str:
    """The reference's model names are disjoint across families (model_builder.py:807-890)."""
    if model_name in ("sam3", "vit_h"):
        return "sam3"
    if model_name in EV_CFG:
        return "efficientvit"
    if model_name.startswith("m"):
        return "repvit"
    if model_name.endswith("m"):
        return "tinyvit"
    raise ValueError(model_name)


def student_backbone(sd: SD, x: torch.Tensor, model_name: str, taps: Optional[dict] = None) -> torch.Tensor:
    fam = backbone_family(model_name)
    if fam == "efficientvit":
        return efficientvit_backbone(sd, x, model_name, taps)
    if fam == "repvit":
        return repvit_backbone(sd, x, model_name, taps)
    return tinyvit_backbone(sd, x, model_name, taps)


def student_head(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """ImageStudentEncoder.head + bilinear to 72x72 (model_builder.py:764-787)."""
    p = TRUNK + "head."
    x = F.conv2d(x, sd[p + "0.weight"])
    x = F.batch_norm(x, sd[p + "1.running_mean"], sd[p + "1.running_var"],
                     sd[p + "1.weight"], sd[p + "1.bias"], False, 0.0, BN_EPS)
    x = F.gelu(x)
    x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
    if x.shape[-1] != EMB or x.shape[-2] != EMB:
        x = F.interpolate(x, size=(EMB, EMB), mode="bilinear", align_corners=False)
    return x


def position_embedding_sine(h: int, w: int, num_pos_feats: int = 256,
                            temperature: float = 10000.0) -> torch.Tensor:
    """PositionEmbeddingSine.forward, normalize=True, scale=2pi (position_encoding.py:92-127)."""
    half = num_pos_feats // 2
    y = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).repeat(1, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).repeat(h, 1)
    eps = 1e-6
    y = y / (y[-1:, :] + eps) * (2 * math.pi)
    x = x / (x[:, -1:] + eps) * (2 * math.pi)
    dim_t = torch.arange(half, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / half)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)  # [C, H, W]


def fpn_neck(sd: SD, which: str, x: torch.Tensor) -> List[torch.Tensor]:
    """One SimpleFPN neck, levels x4, x2, x1 (x0.5 is dropped by scalp=1).

    necks.py:100-125 + vl_combiner.py:94-104.
    """
    p = NECK + which + "."
    l0 = F.conv_transpose2d(x, sd[p + "0.dconv_2x2_0.weight"], sd[p + "0.dconv_2x2_0.bias"], stride=2)
    l0 = F.gelu(l0)
    l0 = F.conv_transpose2d(l0, sd[p + "0.dconv_2x2_1.weight"], sd[p + "0.dconv_2x2_1.bias"], stride=2)
    l0 = F.conv2d(l0, sd[p + "0.conv_1x1.weight"], sd[p + "0.conv_1x1.bias"])
    l0 = F.conv2d(l0, sd[p + "0.conv_3x3.weight"], sd[p + "0.conv_3x3.bias"], padding=1)
    l1 = F.conv_transpose2d(x, sd[p + "1.dconv_2x2.weight"], sd[p + "1.dconv_2x2.bias"], stride=2)
    l1 = F.conv2d(l1, sd[p + "1.conv_1x1.weight"], sd[p + "1.conv_1x1.bias"])
    l1 = F.conv2d(l1, sd[p + "1.conv_3x3.weight"], sd[p + "1.conv_3x3.bias"], padding=1)
    l2 = F.conv2d(x, sd[p + "2.conv_1x1.weight"], sd[p + "2.conv_1x1.bias"])
    l2 = F.conv2d(l2, sd[p + "2.conv_3x3.weight"], sd[p + "2.conv_3x3.bias"], padding=1)
    return [l0, l1, l2]


def forward_image(sd: SD, img: torch.Tensor, model_name: str = "b1",
                  taps: Optional[dict] = None) -> dict:
    """SAM3VLBackbone.forward_image + the conv_s0/conv_s1 projection that
    Sam3Processor.set_image applies in place (vl_combiner.py:81-124,
    sam3_image_processor.py:62-75).  ``img``: [B,3,1008,1008] fp32 normalised."""
    if backbone_family(model_name) == "sam3":  # ViT-H teacher: the trunk output feeds the neck directly
        emb = vit_backbone(sd, img, taps)
    else:
        feat = student_backbone(sd, img, model_name, taps)
        if taps is not None:
            taps["stage_final"] = feat
        emb = student_head(sd, feat)
    if taps is not None:
        taps["trunk"] = emb
    sam3 = fpn_neck(sd, "convs", emb)
    out = {
        "vision_features": sam3[-1],
        "vision_pos_enc": [position_embedding_sine(t.shape[-2], t.shape[-1])[None].repeat(t.shape[0], 1, 1, 1)
                           for t in sam3],
        "backbone_fpn": sam3,
        "sam2_backbone_out": None,
    }
    if NECK + "sam2_convs.0.conv_3x3.weight" in sd:
        sam2 = fpn_neck(sd, "sam2_convs", emb)
        pos = [position_embedding_sine(t.shape[-2], t.shape[-1])[None].repeat(t.shape[0], 1, 1, 1)
               for t in sam2]
        md = SAM + "sam_mask_decoder."
        sam2_src = sam2[-1]
        sam2 = list(sam2)
        sam2[0] = F.conv2d(sam2[0], sd[md + "conv_s0.weight"], sd[md + "conv_s0.bias"])
        sam2[1] = F.conv2d(sam2[1], sd[md + "conv_s1.weight"], sd[md + "conv_s1.bias"])
        out["sam2_backbone_out"] = {"vision_features": sam2_src, "vision_pos_enc": pos,
                                    "backbone_fpn": sam2}
    return out


# --------------------------------------------------------------------------
# prompt encoder (sam/prompt_encoder.py)
# --------------------------------------------------------------------------
def _pe_encoding(sd: SD, coords01: torch.Tensor) -> torch.Tensor:
    """PositionEmbeddingRandom._pe_encoding (prompt_encoder.py:214-221)."""
    g = sd[SAM + "sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = 2 * coords01 - 1
    c = c @ g
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd: SD) -> torch.Tensor:
    """get_dense_pe -> [1,256,72,72] (prompt_encoder.py:63-72,223-234)."""
    grid = torch.ones((EMB, EMB), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / EMB
    x = (grid.cumsum(dim=1) - 0.5) / EMB
    pe = _pe_encoding(sd, torch.stack([x, y], dim=-1))
    return pe.permute(2, 0, 1).unsqueeze(0)


def embed_points(sd: SD, points: torch.Tensor, labels: torch.Tensor, pad: bool) -> torch.Tensor:
    """PromptEncoder._embed_points (prompt_encoder.py:74-118)."""
    pe = SAM + "sam_prompt_encoder."
    points = points + 0.5
    if pad:
        points = torch.cat([points, torch.zeros((points.shape[0], 1, 2))], dim=1)
        labels = torch.cat([labels, -torch.ones((labels.shape[0], 1))], dim=1)
    emb = _pe_encoding(sd, points / IMG)
    lab = labels.unsqueeze(-1)
    emb = torch.where(lab == -1, torch.zeros_like(emb) + sd[pe + "not_a_point_embed.weight"], emb)
    for i in range(4):
        emb = torch.where(lab == i, emb + sd[pe + f"point_embeddings.{i}.weight"], emb)
    return emb


def _layer_norm_2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6):
    """sam/common.py:27-39."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def embed_masks(sd: SD, masks: torch.Tensor) -> torch.Tensor:
    """PromptEncoder._embed_masks: mask_downscaling (prompt_encoder.py:51-59,131-134)."""
    p = SAM + "sam_prompt_encoder.mask_downscaling."
    x = F.conv2d(masks, sd[p + "0.weight"], sd[p + "0.bias"], stride=2)
    x = F.gelu(_layer_norm_2d(x, sd[p + "1.weight"], sd[p + "1.bias"]))
    x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], stride=2)
    x = F.gelu(_layer_norm_2d(x, sd[p + "4.weight"], sd[p + "4.bias"]))
    return F.conv2d(x, sd[p + "6.weight"], sd[p + "6.bias"])


# --------------------------------------------------------------------------
# two-way transformer + mask decoder (sam/transformer.py, sam/mask_decoder.py)
# --------------------------------------------------------------------------
def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def attention(sd: SD, p: str, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
              heads: int = 8) -> torch.Tensor:
    """Attention.forward (transformer.py:226-264): proj, split heads, SDPA, out_proj."""
    q, k, v = _lin(sd, p + "q_proj", q), _lin(sd, p + "k_proj", k), _lin(sd, p + "v_proj", v)

    def split(t):
        b, n, c = t.shape
        return t.reshape(b, n, heads, c // heads).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1]), dim=-1)
    o = att @ v
    b, h, n, c = o.shape
    return _lin(sd, p + "out_proj", o.transpose(1, 2).reshape(b, n, h * c))


def two_way_transformer(sd: SD, src: torch.Tensor, pos: torch.Tensor, tokens: torch.Tensor,
                        taps: Optional[dict] = None):
    """TwoWayTransformer.forward, depth 2 (transformer.py:62-106,155-182)."""
    p = SAM + "sam_mask_decoder.transformer."
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos.flatten(2).permute(0, 2, 1)
    queries, query_pe = tokens, tokens
    for li in range(2):
        q_ = p + f"layers.{li}."
        if li == 0:  # skip_first_layer_pe
            queries = attention(sd, q_ + "self_attn.", queries, queries, queries)
        else:
            q = queries + query_pe
            queries = queries + attention(sd, q_ + "self_attn.", q, q, queries)
        queries = _ln(sd, q_ + "norm1", queries)
        q, k = queries + query_pe, keys + key_pe
        queries = _ln(sd, q_ + "norm2",
                      queries + attention(sd, q_ + "cross_attn_token_to_image.", q, k, keys))
        mlp = _lin(sd, q_ + "mlp.lin2", F.relu(_lin(sd, q_ + "mlp.lin1", queries)))
        queries = _ln(sd, q_ + "norm3", queries + mlp)
        q, k = queries + query_pe, keys + key_pe
        keys = _ln(sd, q_ + "norm4",
                   keys + attention(sd, q_ + "cross_attn_image_to_token.", k, q, queries))
        if taps is not None:
            taps[f"twoway{li}_queries"] = queries
            taps[f"twoway{li}_keys"] = keys
    q, k = queries + query_pe, keys + key_pe
    queries = _ln(sd, p + "norm_final_attn",
                  queries + attention(sd, p + "final_attn_token_to_image.", q, k, keys))
    return queries, keys


def _mlp3(sd: SD, p: str, x: torch.Tensor, sigmoid: bool = False) -> torch.Tensor:
    """MLP with 3 layers, ReLU between (mask_decoder.py:296-319)."""
    x = F.relu(_lin(sd, p + "layers.0", x))
    x = F.relu(_lin(sd, p + "layers.1", x))
    x = _lin(sd, p + "layers.2", x)
    return torch.sigmoid(x) if sigmoid else x


def mask_decoder(sd: SD, image_embed: torch.Tensor, sparse: torch.Tensor, dense: torch.Tensor,
                 feat_s0: torch.Tensor, feat_s1: torch.Tensor, multimask_output: bool,
                 repeat_image: bool, taps: Optional[dict] = None):
    """MaskDecoder.forward/predict_masks (mask_decoder.py:107-242) as configured by
    sam3_tracker_base.py:194-212: pred_obj_scores(+mlp), high-res feats, IoU sigmoid,
    dynamic multimask via stability (delta .05, thr .98; model_builder.py:470-474)."""
    md = SAM + "sam_mask_decoder."
    out_tok = torch.cat([sd[md + "obj_score_token.weight"], sd[md + "iou_token.weight"],
                         sd[md + "mask_tokens.weight"]], dim=0)
    bp = sparse.shape[0]
    tokens = torch.cat([out_tok.unsqueeze(0).expand(bp, -1, -1), sparse], dim=1)
    src = torch.repeat_interleave(image_embed, bp, dim=0) if repeat_image else image_embed
    assert src.shape[0] == bp
    src = src + dense
    pos = torch.repeat_interleave(dense_pe(sd), bp, dim=0)
    b, c, h, w = src.shape
    hs, keys = two_way_transformer(sd, src, pos, tokens, taps)
    iou_tok = hs[:, 1, :]
    mask_toks = hs[:, 2:6, :]
    src = keys.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(src, sd[md + "output_upscaling.0.weight"], sd[md + "output_upscaling.0.bias"], stride=2)
    up = F.gelu(_layer_norm_2d(up + feat_s1, sd[md + "output_upscaling.1.weight"], sd[md + "output_upscaling.1.bias"]))
    up = F.conv_transpose2d(up, sd[md + "output_upscaling.3.weight"], sd[md + "output_upscaling.3.bias"], stride=2)
    up = F.gelu(up + feat_s0)
    hyper = torch.stack([_mlp3(sd, md + f"output_hypernetworks_mlps.{i}.", mask_toks[:, i, :])
                         for i in range(4)], dim=1)
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
    iou = _mlp3(sd, md + "iou_prediction_head.", iou_tok, sigmoid=True)
    obj = _mlp3(sd, md + "pred_obj_score_head.", hs[:, 0, :])
    if taps is not None:
        taps["hs"], taps["upscaled"], taps["all_masks"], taps["all_iou"] = hs, up, masks, iou
    if multimask_output:
        masks, iou = masks[:, 1:], iou[:, 1:]
    else:  # _dynamic_multimask_via_stability (mask_decoder.py:244-292)
        multi, multi_iou = masks[:, 1:], iou[:, 1:]
        best = torch.argmax(multi_iou, dim=-1)
        bi = torch.arange(multi_iou.size(0))
        best_masks, best_iou = multi[bi, best].unsqueeze(1), multi_iou[bi, best].unsqueeze(1)
        single, single_iou = masks[:, 0:1], iou[:, 0:1]
        flat = single.flatten(-2)
        area_i = torch.sum(flat > 0.05, dim=-1).float()
        area_u = torch.sum(flat > -0.05, dim=-1).float()
        stab = torch.where(area_u > 0, area_i / area_u, 1.0)
        stable = stab >= 0.98
        masks = torch.where(stable[..., None, None].expand_as(single), single, best_masks)
        iou = torch.where(stable.expand_as(single_iou), single_iou, best_iou)
    return masks, iou, obj


# --------------------------------------------------------------------------
# post-processing (model/utils/sam1_utils.py:77-119, perflib/connected_components.py)
# --------------------------------------------------------------------------
def fill_small_holes(masks: torch.Tensor, max_hole_area: float = 256.0,
                     mask_threshold: float = 0.0) -> torch.Tensor:
    """Background (score <= thr) 8-connected components of area <= max_hole_area -> thr+10."""
    from scipy import ndimage

    flat = masks.flatten(0, 1)
    out = flat.clone()
    st = np.ones((3, 3), dtype=np.int32)
    for i in range(flat.shape[0]):
        bg = (flat[i] <= mask_threshold).numpy()
        labels, n = ndimage.label(bg, structure=st)
        if n == 0:
            continue
        areas = np.bincount(labels.ravel(), minlength=n + 1)
        hole = (labels > 0) & (areas[labels] <= max_hole_area)
        out[i][torch.from_numpy(hole)] = mask_threshold + 10.0
    return out.view_as(masks)


def postprocess_masks(low_res: torch.Tensor, orig_hw: Tuple[int, int],
                      max_hole_area: float = 256.0) -> torch.Tensor:
    m = low_res.float()
    if max_hole_area > 0:
        m = fill_small_holes(m, max_hole_area)
    return F.interpolate(m, orig_hw, mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------
# end-to-end entry points mirroring the reference API
# --------------------------------------------------------------------------
def normalise_image_u8(img_chw_u8: torch.Tensor) -> torch.Tensor:
    """Sam3Processor.transform for an input that is already 1008x1008
    (sam3_image_processor.py:24-31): /255 then (x-0.5)/0.5."""
    assert img_chw_u8.dtype == torch.uint8 and tuple(img_chw_u8.shape[-2:]) == (IMG, IMG)
    x = img_chw_u8.to(torch.float32) / 255.0
    return (x - 0.5) / 0.5


def resize_u8_antialias(img_chw_u8: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """torchvision v2.Resize on a uint8 tensor the way the reference uses it
    (sam3_image_processor.py:27,57-58; torchvision is absent, so this restates the upstream tensor
    kernel: same size -> identity; otherwise fp32 bilinear with antialias=True,
    align_corners=False, round half to even, clamp, back to uint8).  PARITY UNPINNED: the
    reference holds no fixture for this boundary."""
    if tuple(img_chw_u8.shape[-2:]) == tuple(size):
        return img_chw_u8
    y = F.interpolate(img_chw_u8[None].to(torch.float32), size=tuple(size), mode="bilinear",
                      align_corners=False, antialias=True)[0]
    return y.round().clamp(0, 255).to(torch.uint8)


def processor_transform(img_chw_u8: torch.Tensor) -> torch.Tensor:
    """Sam3Processor.transform (sam3_image_processor.py:24-31): uint8 -> Resize(1008,1008) ->
    float/255 -> Normalize(0.5, 0.5).  -> [3,1008,1008] fp32."""
    assert img_chw_u8.dtype == torch.uint8 and img_chw_u8.dim() == 3
    return normalise_image_u8(resize_u8_antialias(img_chw_u8, (IMG, IMG)))


def set_image(sd: SD, img: torch.Tensor, orig_hw: Tuple[int, int], model_name: str = "b1",
              taps: Optional[dict] = None) -> dict:
    """img: [B,3,1008,1008] fp32 normalised (B=1 for set_image)."""
    return {"original_height": orig_hw[0], "original_width": orig_hw[1],
            "backbone_out": forward_image(sd, img, model_name, taps)}


def predict_inst(sd: SD, state: dict, point_coords=None, point_labels=None, box=None,
                 mask_input=None, multimask_output: bool = True, return_logits: bool = False,
                 normalize_coords: bool = True, img_idx: int = 0, taps: Optional[dict] = None):
    """Sam3Image.predict_inst -> SAM3InteractiveImagePredictor.predict/_predict
    (sam3_image.py:599-636, sam1_task_predictor.py:230-430)."""
    h, w = state["original_height"], state["original_width"]
    s2 = state["backbone_out"]["sam2_backbone_out"]
    fpn = s2["backbone_fpn"]
    image_embed = fpn[2][img_idx:img_idx + 1] + sd[SAM + "no_mem_embed"].view(1, -1, 1, 1)
    feat_s0, feat_s1 = fpn[0][img_idx:img_idx + 1], fpn[1][img_idx:img_idx + 1]

    coords = labels = None
    if point_coords is not None:
        assert point_labels is not None
        coords = torch.as_tensor(point_coords, dtype=torch.float).clone()
        if normalize_coords:
            coords[..., 0] = coords[..., 0] / w
            coords[..., 1] = coords[..., 1] / h
        coords = coords * IMG
        labels = torch.as_tensor(point_labels, dtype=torch.int)
        if coords.dim() == 2:
            coords, labels = coords[None], labels[None]
    if box is not None:
        bx = torch.as_tensor(box, dtype=torch.float).reshape(-1, 2, 2).clone()
        if normalize_coords:
            bx[..., 0] = bx[..., 0] / w
            bx[..., 1] = bx[..., 1] / h
        bx = bx * IMG
        bl = torch.tensor([[2, 3]], dtype=torch.int).repeat(bx.size(0), 1)
        if coords is not None:
            coords, labels = torch.cat([bx, coords], dim=1), torch.cat([bl, labels], dim=1)
        else:
            coords, labels = bx, bl
    if coords is not None:
        sparse = embed_points(sd, coords, labels.float(), pad=True)
        bp = coords.shape[0]
    else:
        sparse = torch.empty((1, 0, 256))
        bp = 1
    if mask_input is not None:
        mi = torch.as_tensor(mask_input, dtype=torch.float)
        if mi.dim() == 3:
            mi = mi[None]
        dense = embed_masks(sd, mi)
    else:
        dense = sd[SAM + "sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(bp, -1, EMB, EMB)
    batched = coords is not None and coords.shape[0] > 1
    low_res, iou, _obj = mask_decoder(sd, image_embed, sparse, dense, feat_s0, feat_s1,
                                      multimask_output, batched, taps)
    masks = postprocess_masks(low_res, (h, w))
    low_res = torch.clamp(low_res, -32.0, 32.0)
    if not return_logits:
        masks = masks > 0.0
    return (masks.squeeze(0).float().numpy(), iou.squeeze(0).float().numpy(),
            low_res.squeeze(0).float().numpy())


# --------------------------------------------------------------------------
# MobileCLIP-S0 student text encoder (text_encoder_student.py, backbones/mobile_clip.py)
# --------------------------------------------------------------------------
TEXT = "backbone.language_backbone."


def _bn2d(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, BN_EPS)


def _repmixer_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """RepMixerBlock.forward (mobile_clip.py:647-702) on [B, S, D], training-time branches:
    token mixer x + ls*(mixer(x) - norm(x)) with mixer = BN(x) + BN(dw1x11(x)), norm = BN(x)
    (MobileOneBlock, mobile_clip.py:49-160; the (1,11) kernel disables the scale branch), then
    x + ls * ConvFFN(x) (dw1x11 + BN, 1x1 -> GELU -> 1x1; mobile_clip.py:499-548)."""
    x = x.permute(0, 2, 1).unsqueeze(2)  # [B, D, 1, S]
    d = x.shape[1]
    t = p + "token_mixer."
    mixer = _bn2d(sd, t + "mixer.rbr_skip", x) + _bn2d(
        sd, t + "mixer.rbr_conv.0.bn", F.conv2d(x, sd[t + "mixer.rbr_conv.0.conv.weight"], None, 1, (0, 5), 1, d))
    norm = _bn2d(sd, t + "norm.rbr_skip", x)
    x = x + sd[t + "layer_scale"] * (mixer - norm)
    c = p + "convffn."
    y = _bn2d(sd, c + "conv.bn", F.conv2d(x, sd[c + "conv.conv.weight"], None, 1, (0, 5), 1, d))
    y = F.gelu(F.conv2d(y, sd[c + "fc1.weight"], sd[c + "fc1.bias"]))
    y = F.conv2d(y, sd[c + "fc2.weight"], sd[c + "fc2.bias"])
    x = x + sd[p + "layer_scale"] * y
    return x.squeeze(2).permute(0, 2, 1)


def _text_transformer_layer(sd: SD, p: str, x: torch.Tensor, heads: int, causal: bool = False) -> torch.Tensor:
    """TransformerEncoder.forward (mobile_clip.py:427-491): pre-norm MHA over ALL positions (the
    student passes no key-padding mask, text_encoder_student.py:48-50; S0 is non-causal) and
    pre-norm FFN."""
    b, s_len, d = x.shape
    y = F.layer_norm(x, (d,), sd[p + "pre_norm_mha.0.weight"], sd[p + "pre_norm_mha.0.bias"], 1e-5)
    a = p + "pre_norm_mha.1."
    qkv = F.linear(y, sd[a + "qkv_proj.weight"], sd[a + "qkv_proj.bias"]).reshape(b, s_len, 3, heads, -1)
    qkv = qkv.transpose(1, 3).contiguous()  # [B, heads, 3, S, hd]
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    attn = torch.matmul(q * (q.shape[-1] ** -0.5), k.transpose(-1, -2))
    if causal:  # build_attention_mask (mobile_clip.py:826-832): -inf above the diagonal
        attn = attn + torch.full((s_len, s_len), float("-inf")).triu_(1)
    attn = attn.float().softmax(-1)
    o = torch.matmul(attn, v).transpose(1, 2).reshape(b, s_len, -1)
    x = F.linear(o, sd[a + "out_proj.weight"], sd[a + "out_proj.bias"]) + x
    y = F.layer_norm(x, (d,), sd[p + "pre_norm_ffn.0.weight"], sd[p + "pre_norm_ffn.0.bias"], 1e-5)
    y = F.gelu(F.linear(y, sd[p + "pre_norm_ffn.1.weight"], sd[p + "pre_norm_ffn.1.bias"]))
    return x + F.linear(y, sd[p + "pre_norm_ffn.4.weight"], sd[p + "pre_norm_ffn.4.bias"])


def text_encoder_student(sd: SD, tokens: torch.Tensor, n_layers: int = 4, heads: int = 8, variant: str = "mct",
                         causal: bool = False):
    """TextStudentEncoder.forward after tokenisation (text_encoder_student.py:40-58):
    tokens int64 [B, S] -> (mask [B,S] True = padding, memory [S,B,256], embeds [S,B,dim]).
    embed_scale is computed by the reference but never applied (mobile_clip.py:743,815-823).
    variant "mct" (MobileCLIP-S0): RepMixerBlock, n_layers x TransformerEncoder, RepMixerBlock; "base" (the
    other students, model_builder.py:525-546): n_layers x TransformerEncoder, causal for MobileCLIP-B."""
    e = TEXT + "encoder."
    s_len = tokens.shape[1]
    emb = F.embedding(tokens, sd[e + "embedding_layer.weight"])
    emb = emb + sd[e + "positional_embedding.pos_embed.pos_embed"][0, 0, :s_len][None]
    if variant == "mct":
        x = _repmixer_block(sd, e + "transformer.0.", emb)
        for i in range(1, n_layers + 1):
            x = _text_transformer_layer(sd, e + f"transformer.{i}.", x, heads, causal)
        x = _repmixer_block(sd, e + f"transformer.{n_layers + 1}.", x)
    else:
        x = emb
        for i in range(n_layers):
            x = _text_transformer_layer(sd, e + f"transformer.{i}.", x, heads, causal)
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd[e + "final_layer_norm.weight"], sd[e + "final_layer_norm.bias"], 1e-5)
    mem = F.linear(x, sd[TEXT + "projector.weight"], sd[TEXT + "projector.bias"])
    return tokens == 0, mem.transpose(0, 1), emb.transpose(0, 1)
