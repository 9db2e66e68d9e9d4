"""State-dict schema of the EfficientSAM3 image hot path + a seeded initialiser.

The reference's checkpoints (``state_dict`` of ``Sam3Image``) are the weight
interchange format between the reference, the oracle and the HIP engine
(SURVEY.md §5 "Checkpoint / resume", Appendix C).  No checkpoint exists
offline, so this module re-creates the *names and shapes* of every tensor the
hot path reads and fills them from a seeded generator with realistic
statistics (randomised BatchNorm running stats, fan-in scaled weights) so that
activations and mask logits have O(1) magnitudes instead of the ~0.02 that
default ``nn.Module`` init produces.

Key layout follows (reference file:line):
  * EfficientViT backbone ........ sam3/sam3/backbones/efficientvit/efficientvit/backbone.py:33-156
  * ConvLayer / DSConv / MBConv /
    LiteMLA / EfficientViTBlock .. sam3/sam3/backbones/efficientvit/nn/ops.py:39-80,273-367,521-733
  * student head ................. sam3/sam3/model_builder.py:764-787
  * dual ViTDet neck ............. sam3/sam3/model/necks.py:13-98
  * prompt encoder ............... sam3/sam3/sam/prompt_encoder.py:12-61,203-212
  * two-way transformer .......... sam3/sam3/sam/transformer.py:16-60,108-153,185-215
  * mask decoder ................. sam3/sam3/sam/mask_decoder.py:12-105,294-319
  * tracker wrapper params ....... sam3/sam3/model/sam3_tracker_base.py:110,179-218
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

Shape = Tuple[int, ...]

TRUNK = "backbone.vision_backbone.trunk.model."  # ListWrapper -> ImageStudentEncoder
EV_BB = TRUNK + "backbone.model."  # EfficientViTTrunkWrapper -> EfficientViTBackbone
NECK = "backbone.vision_backbone."
SAM = "inst_interactive_predictor.model."

EFFICIENTVIT_CFG = {
    # name: (width_list, depth_list, dim)   backbone.py:159-196
    "b0": ([8, 16, 32, 64, 128], [1, 2, 2, 2, 2], 16),
    "b1": ([16, 32, 64, 128, 256], [1, 2, 3, 3, 4], 16),
    "b2": ([24, 48, 96, 192, 384], [1, 3, 4, 4, 6], 32),
}

# RepViT: (k, t, c, use_se, use_hs, stride) per block (repvit.py:291-384,430-506)
REPVIT_CFG = {
    # explicit tables (k, t, c, SE, HS, s), repvit.py:320-350 (m0_9), :386-416 (m1_1), :470-506 (m2_3)
    "m0.9": [(3, 2, 48, 1, 0, 1), (3, 2, 48, 0, 0, 1), (3, 2, 48, 0, 0, 1), (3, 2, 96, 0, 0, 2),
             (3, 2, 96, 1, 0, 1), (3, 2, 96, 0, 0, 1), (3, 2, 96, 0, 0, 1), (3, 2, 192, 0, 1, 2)]
            + [(3, 2, 192, 1 - (i % 2), 1, 1) for i in range(14)] + [(3, 2, 192, 0, 1, 1), (3, 2, 384, 0, 1, 2),
                                                                      (3, 2, 384, 1, 1, 1), (3, 2, 384, 0, 1, 1)],
    "m1.1": [(3, 2, 64, 1, 0, 1), (3, 2, 64, 0, 0, 1), (3, 2, 64, 0, 0, 1), (3, 2, 128, 0, 0, 2),
             (3, 2, 128, 1, 0, 1), (3, 2, 128, 0, 0, 1), (3, 2, 128, 0, 0, 1), (3, 2, 256, 0, 1, 2)]
            + [(3, 2, 256, 1 - (i % 2), 1, 1) for i in range(12)] + [(3, 2, 256, 0, 1, 1), (3, 2, 512, 0, 1, 2),
                                                                      (3, 2, 512, 1, 1, 1), (3, 2, 512, 0, 1, 1)],
    "m2.3": [(3, 2, 80, 1 - (i % 2) if i < 6 else 0, 0, 1) for i in range(7)] + [(3, 2, 160, 0, 0, 2)]
            + [(3, 2, 160, 1 - (i % 2) if i < 6 else 0, 0, 1) for i in range(7)] + [(3, 2, 320, 0, 1, 2)]
            + [(3, 2, 320, 1 - (i % 2), 1, 1) for i in range(34)] + [(3, 2, 320, 0, 1, 1), (3, 2, 640, 0, 1, 2),
                                                                      (3, 2, 640, 1, 1, 1), (3, 2, 640, 0, 1, 1)],
}

TINYVIT_CFG = {
    # name: (embed_dims, depths, num_heads, window_sizes)   tiny_vit.py:641-679; mlp_ratio 4, MBConv expand 4
    "5m": ([64, 128, 160, 320], [2, 2, 6, 2], [2, 4, 5, 10], [7, 7, 14, 7]),
    "11m": ([64, 128, 256, 448], [2, 2, 6, 2], [2, 4, 8, 14], [7, 7, 14, 7]),
    "21m": ([96, 192, 384, 576], [2, 2, 6, 2], [3, 6, 12, 18], [7, 7, 14, 7]),
}

EMBED_DIM = 1024  # ImageStudentEncoder embed_dim (model_builder.py:913-919)
D_MODEL = 256


# ----------------------------------------------------------------------------
# schema builders: name -> (shape, kind)
#   kind in {"conv", "dw", "linear", "bias", "bn_w", "bn_b", "bn_m", "bn_v",
#            "bn_n", "ln_w", "ln_b", "embed", "gauss"}
# ----------------------------------------------------------------------------
class _Schema(OrderedDict):
    def conv(self, name: str, cout: int, cin: int, k: int, groups: int = 1, bias: bool = False):
        kind = "dw" if groups == cin and groups == cout and groups > 1 else "conv"
        self[name + ".weight"] = ((cout, cin // groups, k, k), kind)
        if bias:
            self[name + ".bias"] = ((cout,), "bias")

    def convT(self, name: str, cin: int, cout: int, k: int):
        self[name + ".weight"] = ((cin, cout, k, k), "convT")
        self[name + ".bias"] = ((cout,), "bias")

    def bn(self, name: str, c: int):
        self[name + ".weight"] = ((c,), "bn_w")
        self[name + ".bias"] = ((c,), "bn_b")
        self[name + ".running_mean"] = ((c,), "bn_m")
        self[name + ".running_var"] = ((c,), "bn_v")
        self[name + ".num_batches_tracked"] = ((), "bn_n")

    def ln(self, name: str, c: int):
        self[name + ".weight"] = ((c,), "ln_w")
        self[name + ".bias"] = ((c,), "ln_b")

    def linear(self, name: str, cout: int, cin: int):
        self[name + ".weight"] = ((cout, cin), "linear")
        self[name + ".bias"] = ((cout,), "bias")

    # EfficientViT ConvLayer = conv (+bias) (+BN)
    def conv_layer(self, name: str, cin: int, cout: int, k: int, groups: int = 1,
                   bias: bool = False, norm: bool = True):
        self.conv(name + ".conv", cout, cin, k, groups=groups, bias=bias)
        if norm:
            self.bn(name + ".norm", cout)


def efficientvit_schema(model_name: str = "b1") -> _Schema:
    """EfficientViT-B{0,1,2} backbone tensors (backbone.py:33-156)."""
    widths, depths, dim = EFFICIENTVIT_CFG[model_name]
    s = _Schema()
    p = EV_BB
    # input stem: ConvLayer 3x3 s2 + depth_list[0] x Residual(DSConv)
    s.conv_layer(p + "input_stem.op_list.0", 3, widths[0], 3)
    for i in range(depths[0]):
        q = p + f"input_stem.op_list.{i + 1}.main."
        s.conv_layer(q + "depth_conv", widths[0], widths[0], 3, groups=widths[0])
        s.conv_layer(q + "point_conv", widths[0], widths[0], 1)
    cin = widths[0]
    # stages 1-2: MBConv (expand 4), BN after every conv
    for si, (w, d) in enumerate(zip(widths[1:3], depths[1:3])):
        for i in range(d):
            q = p + f"stages.{si}.op_list.{i}.main."
            mid = round(cin * 4)
            s.conv_layer(q + "inverted_conv", cin, mid, 1)
            s.conv_layer(q + "depth_conv", mid, mid, 3, groups=mid)
            s.conv_layer(q + "point_conv", mid, w, 1)
            cin = w
    # stages 3-4: MBConv s2 (fewer_norm) + d x EfficientViTBlock
    for si, (w, d) in enumerate(zip(widths[3:], depths[3:]), start=2):
        q = p + f"stages.{si}.op_list.0.main."
        mid = round(cin * 4)
        s.conv_layer(q + "inverted_conv", cin, mid, 1, bias=True, norm=False)
        s.conv_layer(q + "depth_conv", mid, mid, 3, groups=mid, bias=True, norm=False)
        s.conv_layer(q + "point_conv", mid, w, 1)
        cin = w
        heads = cin // dim
        total = heads * dim
        for i in range(d):
            q = p + f"stages.{si}.op_list.{i + 1}."
            c = q + "context_module.main."
            s.conv_layer(c + "qkv", cin, 3 * total, 1, norm=False)
            s.conv(c + "aggreg.0.0", 3 * total, 3 * total, 5, groups=3 * total)
            s.conv(c + "aggreg.0.1", 3 * total, 3 * total, 1, groups=3 * heads)
            s.conv_layer(c + "proj", 2 * total, cin, 1)
            m = q + "local_module.main."
            mid = round(cin * 4)
            s.conv_layer(m + "inverted_conv", cin, mid, 1, bias=True, norm=False)
            s.conv_layer(m + "depth_conv", mid, mid, 3, groups=mid, bias=True, norm=False)
            s.conv_layer(m + "point_conv", mid, cin, 1)
    return s


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    """timm.layers.make_divisible (SqueezeExcite uses round_limit=0.0)."""
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def repvit_out_channels(model_name: str) -> int:
    return REPVIT_CFG[model_name][-1][2]


def repvit_schema(model_name: str = "m1.1") -> _Schema:
    """RepViT feature trunk (repvit.py:29-47,84-161,232-252; classifier stripped by the builder,
    model_builder.py:845-860).  Keys: <trunk>.backbone.model.features.{i}..."""
    cfgs = REPVIT_CFG[model_name]
    s = _Schema()
    p = EV_BB + "features."

    def conv_bn(name, cin, cout, k, groups=1):
        s.conv(name + ".c", cout, cin, k, groups=groups)
        s.bn(name + ".bn", cout)

    c0 = cfgs[0][2]
    conv_bn(p + "0.0", 3, c0 // 2, 3)
    conv_bn(p + "0.2", c0 // 2, c0, 3)
    cin = c0
    for i, (k, t, c, use_se, use_hs, stride) in enumerate(cfgs, start=1):
        q = p + f"{i}."
        if stride == 2:
            conv_bn(q + "token_mixer.0", cin, cin, k, groups=cin)
            if use_se:
                raise NotImplementedError("SE in a stride-2 RepViT block")
            conv_bn(q + "token_mixer.2", cin, c, 1)
        else:
            assert cin == c
            conv_bn(q + "token_mixer.0.conv", c, c, 3, groups=c)
            s[q + "token_mixer.0.conv1.weight"] = ((c, 1, 1, 1), "dw1")
            s[q + "token_mixer.0.conv1.bias"] = ((c,), "bias")
            s[q + "token_mixer.0.bn.weight"] = ((c,), "bn_w_repdw")
            s[q + "token_mixer.0.bn.bias"] = ((c,), "bn_b")
            s[q + "token_mixer.0.bn.running_mean"] = ((c,), "bn_m")
            s[q + "token_mixer.0.bn.running_var"] = ((c,), "bn_v")
            s[q + "token_mixer.0.bn.num_batches_tracked"] = ((), "bn_n")
            if use_se:
                rd = make_divisible(c * 0.25, 8, round_limit=0.0)
                s.conv(q + "token_mixer.1.fc1", rd, c, 1, bias=True)
                s[q + "token_mixer.1.fc2.weight"] = ((c, rd, 1, 1), "conv")
                s[q + "token_mixer.1.fc2.bias"] = ((c,), "se_bias")
        conv_bn(q + "channel_mixer.m.0", c, 2 * c, 1)
        s.conv(q + "channel_mixer.m.2.c", c, 2 * c, 1)
        for suffix, kind in (("weight", "bn_w_res"), ("bias", "bn_b"), ("running_mean", "bn_m"),
                             ("running_var", "bn_v")):
            s[q + "channel_mixer.m.2.bn." + suffix] = ((c,), kind)
        s[q + "channel_mixer.m.2.bn.num_batches_tracked"] = ((), "bn_n")
        cin = c
    return s


def tinyvit_schema(model_name: str = "11m") -> _Schema:
    """TinyViT trunk built with img_size=1008, num_classes=0 (tiny_vit.py:67-154,196-386,453-536;
    model_builder.py:869-905).  Keys: <trunk>.backbone.model.{patch_embed,layers.N}...  The
    non-persistent attention_bias_idxs buffers are not part of the state dict."""
    dims, depths, heads, windows = TINYVIT_CFG[model_name]
    s = _Schema()
    p = EV_BB

    def conv_bn(name, cin, cout, k, groups=1, res_end=False):
        s.conv(name + ".c", cout, cin, k, groups=groups)
        if res_end:
            for suffix, kind in (("weight", "bn_w_res"), ("bias", "bn_b"), ("running_mean", "bn_m"),
                                 ("running_var", "bn_v")):
                s[name + ".bn." + suffix] = ((cout,), kind)
            s[name + ".bn.num_batches_tracked"] = ((), "bn_n")
        else:
            s.bn(name + ".bn", cout)

    conv_bn(p + "patch_embed.seq.0", 3, dims[0] // 2, 3)
    conv_bn(p + "patch_embed.seq.2", dims[0] // 2, dims[0], 3)
    for li, (dim, depth) in enumerate(zip(dims, depths)):
        q = p + f"layers.{li}."
        for bi in range(depth):
            b = q + f"blocks.{bi}."
            if li == 0:  # MBConv
                hid = int(dim * 4.0)
                conv_bn(b + "conv1", dim, hid, 1)
                conv_bn(b + "conv2", hid, hid, 3, groups=hid)
                conv_bn(b + "conv3", hid, dim, 1, res_end=True)
            else:        # TinyViTBlock
                ws = windows[li]
                s.ln(b + "attn.norm", dim)
                s.linear(b + "attn.qkv", 3 * dim, dim)
                s[b + "attn.proj.weight"] = ((dim, dim), "linear_res")
                s[b + "attn.proj.bias"] = ((dim,), "bias")
                s[b + "attn.attention_biases"] = ((heads[li], ws * ws), "attn_bias")
                conv_bn(b + "local_conv", dim, dim, 3, groups=dim)
                s.ln(b + "mlp.norm", dim)
                s.linear(b + "mlp.fc1", 4 * dim, dim)
                s[b + "mlp.fc2.weight"] = ((dim, 4 * dim), "linear_res")
                s[b + "mlp.fc2.bias"] = ((dim,), "bias")
        if li < len(dims) - 1:
            out = dims[li + 1]
            conv_bn(q + "downsample.conv1", dim, out, 1)
            conv_bn(q + "downsample.conv2", out, out, 3, groups=out)
            conv_bn(q + "downsample.conv3", out, out, 1)
    return s


def student_head_schema(c_backbone: int) -> _Schema:
    """ImageStudentEncoder.head (model_builder.py:770-775)."""
    s = _Schema()
    s.conv(TRUNK + "head.0", EMBED_DIM, c_backbone, 1, bias=False)
    s.bn(TRUNK + "head.1", EMBED_DIM)
    s.conv(TRUNK + "head.3", EMBED_DIM, EMBED_DIM, 3, bias=True)
    return s


def neck_schema(which: str) -> _Schema:
    """One SimpleFPN neck, ``which`` in {"convs", "sam2_convs"} (necks.py:36-98)."""
    s = _Schema()
    p = NECK + which + "."
    dim = EMBED_DIM
    s.convT(p + "0.dconv_2x2_0", dim, dim // 2, 2)
    s.convT(p + "0.dconv_2x2_1", dim // 2, dim // 4, 2)
    s.conv(p + "0.conv_1x1", D_MODEL, dim // 4, 1, bias=True)
    s.conv(p + "0.conv_3x3", D_MODEL, D_MODEL, 3, bias=True)
    s.convT(p + "1.dconv_2x2", dim, dim // 2, 2)
    s.conv(p + "1.conv_1x1", D_MODEL, dim // 2, 1, bias=True)
    s.conv(p + "1.conv_3x3", D_MODEL, D_MODEL, 3, bias=True)
    for lvl in (2, 3):
        s.conv(p + f"{lvl}.conv_1x1", D_MODEL, dim, 1, bias=True)
        s.conv(p + f"{lvl}.conv_3x3", D_MODEL, D_MODEL, 3, bias=True)
    return s


def _attention_schema(s: _Schema, p: str, embed: int, internal: int):
    for n in ("q_proj", "k_proj", "v_proj"):
        s.linear(p + n, internal, embed)
    s.linear(p + "out_proj", embed, internal)


def _mlp_schema(s: _Schema, p: str, cin: int, hidden: int, cout: int, n: int = 3):
    dims = [cin] + [hidden] * (n - 1) + [cout]
    for i in range(n):
        s.linear(p + f"layers.{i}", dims[i + 1], dims[i])


def sam_heads_schema() -> _Schema:
    """Prompt encoder + mask decoder + no_mem_embed (sam3_tracker_base.py:110,179-218)."""
    s = _Schema()
    s[SAM + "no_mem_embed"] = ((1, 1, D_MODEL), "embed")
    pe = SAM + "sam_prompt_encoder."
    s[pe + "pe_layer.positional_encoding_gaussian_matrix"] = ((2, D_MODEL // 2), "gauss")
    for i in range(4):
        s[pe + f"point_embeddings.{i}.weight"] = ((1, D_MODEL), "embed")
    s[pe + "not_a_point_embed.weight"] = ((1, D_MODEL), "embed")
    # mask_downscaling (mask_in_chans = 16): conv k2s2 1->4, LN2d, GELU, conv k2s2 4->16, LN2d, GELU, conv1x1 16->256
    s.conv(pe + "mask_downscaling.0", 4, 1, 2, bias=True)
    s.ln(pe + "mask_downscaling.1", 4)
    s.conv(pe + "mask_downscaling.3", 16, 4, 2, bias=True)
    s.ln(pe + "mask_downscaling.4", 16)
    s.conv(pe + "mask_downscaling.6", D_MODEL, 16, 1, bias=True)
    s[pe + "no_mask_embed.weight"] = ((1, D_MODEL), "embed")

    md = SAM + "sam_mask_decoder."
    for li in range(2):
        q = md + f"transformer.layers.{li}."
        _attention_schema(s, q + "self_attn.", D_MODEL, D_MODEL)
        s.ln(q + "norm1", D_MODEL)
        _attention_schema(s, q + "cross_attn_token_to_image.", D_MODEL, D_MODEL // 2)
        s.ln(q + "norm2", D_MODEL)
        s.linear(q + "mlp.lin1", 2048, D_MODEL)
        s.linear(q + "mlp.lin2", D_MODEL, 2048)
        s.ln(q + "norm3", D_MODEL)
        s.ln(q + "norm4", D_MODEL)
        _attention_schema(s, q + "cross_attn_image_to_token.", D_MODEL, D_MODEL // 2)
    _attention_schema(s, md + "transformer.final_attn_token_to_image.", D_MODEL, D_MODEL // 2)
    s.ln(md + "transformer.norm_final_attn", D_MODEL)
    s[md + "iou_token.weight"] = ((1, D_MODEL), "embed")
    s[md + "mask_tokens.weight"] = ((4, D_MODEL), "embed")
    s[md + "obj_score_token.weight"] = ((1, D_MODEL), "embed")
    s.convT(md + "output_upscaling.0", D_MODEL, D_MODEL // 4, 2)
    s.ln(md + "output_upscaling.1", D_MODEL // 4)
    s.convT(md + "output_upscaling.3", D_MODEL // 4, D_MODEL // 8, 2)
    s.conv(md + "conv_s0", D_MODEL // 8, D_MODEL, 1, bias=True)
    s.conv(md + "conv_s1", D_MODEL // 4, D_MODEL, 1, bias=True)
    for i in range(4):
        _mlp_schema(s, md + f"output_hypernetworks_mlps.{i}.", D_MODEL, D_MODEL, D_MODEL // 8)
    _mlp_schema(s, md + "iou_prediction_head.", D_MODEL, 256, 4)
    _mlp_schema(s, md + "pred_obj_score_head.", D_MODEL, D_MODEL, 1)
    return s


TEXT = "backbone.language_backbone."
TEXT_ENCODER_CFG = {
    # name: (dim, n_transformer_layers, heads, variant, causal_masking)   model_builder.py:499-560
    "MobileCLIP-S0": (512, 4, 8, "mct", False),
    "MobileCLIP-S1": (512, 12, 8, "base", False),
    "MobileCLIP2-S0": (512, 12, 8, "base", False),
    "MobileCLIP2-S2": (512, 12, 8, "base", False),
    "MobileCLIP-B": (512, 12, 8, "base", True),
    "MobileCLIP2-S3": (768, 12, 12, "base", False),
    "MobileCLIP2-S4": (768, 12, 12, "base", False),
    "MobileCLIP2-L": (768, 12, 12, "base", False),
}


def text_encoder_schema(kind: str = "MobileCLIP-S0", context_length: int = 16) -> _Schema:
    """TextStudentEncoder (text_encoder_student.py:9-58) around MobileCLIPTextTransformer
    (mobile_clip.py:709-901): token + learned positional embeddings, RepMixerBlock, N x
    TransformerEncoder, RepMixerBlock ("mct" variant), final LayerNorm, projector 512 -> 256.
    ``context_length`` is the length of the positional table in the state dict (the reference
    builds at 77 and truncates to the requested length after loading, model_builder.py:1035-1047)."""
    dim, n_layers, heads, variant, _causal = TEXT_ENCODER_CFG[kind]
    first = 1 if variant == "mct" else 0   # "base": transformer.0 .. transformer.N-1 are all TransformerEncoder layers
    s = _Schema()
    e = TEXT + "encoder."
    s[e + "projection_layer"] = ((dim, dim), "embed")  # present in the reference, unused by this path
    s[e + "embedding_layer.weight"] = ((49408, dim), "embed")
    s[e + "positional_embedding.pos_embed.pos_embed"] = ((1, 1, context_length, dim), "embed")

    def bn(name):
        s.bn(name, dim)

    def repmixer(q):
        s[q + "layer_scale"] = ((dim, 1, 1), "layer_scale")
        s[q + "token_mixer.layer_scale"] = ((dim, 1, 1), "layer_scale")
        bn(q + "token_mixer.norm.rbr_skip")
        bn(q + "token_mixer.mixer.rbr_skip")
        s[q + "token_mixer.mixer.rbr_conv.0.conv.weight"] = ((dim, 1, 1, 11), "dw")
        bn(q + "token_mixer.mixer.rbr_conv.0.bn")
        s[q + "convffn.conv.conv.weight"] = ((dim, 1, 1, 11), "dw")
        bn(q + "convffn.conv.bn")
        s.conv(q + "convffn.fc1", 4 * dim, dim, 1, bias=True)
        s.conv(q + "convffn.fc2", dim, 4 * dim, 1, bias=True)

    if variant == "mct":
        repmixer(e + "transformer.0.")
    for i in range(first, n_layers + first):
        q = e + f"transformer.{i}."
        s.ln(q + "pre_norm_mha.0", dim)
        s.linear(q + "pre_norm_mha.1.qkv_proj", 3 * dim, dim)
        s[q + "pre_norm_mha.1.out_proj.weight"] = ((dim, dim), "linear_res")
        s[q + "pre_norm_mha.1.out_proj.bias"] = ((dim,), "bias")
        s.ln(q + "pre_norm_ffn.0", dim)
        s.linear(q + "pre_norm_ffn.1", 4 * dim, dim)
        s[q + "pre_norm_ffn.4.weight"] = ((dim, 4 * dim), "linear_res")
        s[q + "pre_norm_ffn.4.bias"] = ((dim,), "bias")
    if variant == "mct":
        repmixer(e + f"transformer.{n_layers + 1}.")
    s.ln(e + "final_layer_norm", dim)
    s.linear(TEXT + "projector", D_MODEL, dim)
    return s


def synthetic_text_state_dict(kind: str = "MobileCLIP-S0", context_length: int = 16, seed: int = 0):
    return init_state_dict(text_encoder_schema(kind, context_length), seed)


VIT_TRUNK = NECK + "trunk."
VIT_CFG = dict(embed_dim=1024, depth=32, heads=16, mlp_hidden=4736, patch=14, pretrain_grid=24, window=24,
               global_blocks=(7, 15, 23, 31))  # _create_vit_backbone, model_builder.py:70-97


def vit_schema() -> _Schema:
    """ViT-H teacher trunk (vitdet.py:616-859 as configured by model_builder.py:70-97): patch embed
    14x14 s14 without bias, abs-pos table of the 336-px pre-training grid (24x24 + cls), ln_pre, 32
    blocks (LN, qkv, proj, LN, Mlp 1024->4736->1024).  The complex ``freqs_cis`` buffers are derived
    constants and not part of this schema."""
    c = VIT_CFG
    d = c["embed_dim"]
    s = _Schema()
    p = VIT_TRUNK
    s[p + "pos_embed"] = ((1, c["pretrain_grid"] ** 2 + 1, d), "pos_small")
    s.conv(p + "patch_embed.proj", d, 3, c["patch"])
    s[p + "ln_pre.weight"] = ((d,), "ln_w_small")  # keeps the residual stream (and the mask logits) O(1)
    s[p + "ln_pre.bias"] = ((d,), "ln_b")
    for i in range(c["depth"]):
        q = p + f"blocks.{i}."
        s.ln(q + "norm1", d)
        s.linear(q + "attn.qkv", 3 * d, d)
        s[q + "attn.proj.weight"] = ((d, d), "linear_res")
        s[q + "attn.proj.bias"] = ((d,), "bias")
        s.ln(q + "norm2", d)
        s.linear(q + "mlp.fc1", c["mlp_hidden"], d)
        s[q + "mlp.fc2.weight"] = ((d, c["mlp_hidden"]), "linear_res")
        s[q + "mlp.fc2.bias"] = ((d,), "bias")
    return s


def pcs_schema() -> _Schema:
    """Text-grounding (PCS) detector of Sam3Image (model_builder.py:116-300): geometry encoder, fusion
    encoder (6 layers), DETR decoder (6 layers, 200 queries, box RPB, presence token), segmentation head,
    dot-product scoring.  Key layout = the reference's module tree."""
    s = _Schema()
    d, ff = D_MODEL, 2048

    def mha(p):
        s[p + "in_proj_weight"] = ((3 * d, d), "linear")
        s[p + "in_proj_bias"] = ((3 * d,), "bias")
        s[p + "out_proj.weight"] = ((d, d), "linear_res")
        s[p + "out_proj.bias"] = ((d,), "bias")

    def mlp(p, dims):
        for i in range(len(dims) - 1):
            s.linear(p + f"layers.{i}", dims[i + 1], dims[i])

    def enc_layer(p):
        mha(p + "self_attn.")
        mha(p + "cross_attn_image.")
        s.linear(p + "linear1", ff, d)
        s[p + "linear2.weight"] = ((d, ff), "linear_res")
        s[p + "linear2.bias"] = ((d,), "bias")
        for n in ("norm1", "norm2", "norm3"):
            s.ln(p + n, d)

    g = "geometry_encoder."
    s[g + "label_embed.weight"] = ((2, d), "embed")
    s[g + "cls_embed.weight"] = ((1, d), "embed")
    s.linear(g + "points_direct_project", d, 2)
    s.linear(g + "points_pool_project", d, d)
    s.linear(g + "points_pos_enc_project", d, d)
    s.linear(g + "boxes_direct_project", d, 4)
    s.conv(g + "boxes_pool_project", d, d, 7, bias=True)
    s.linear(g + "boxes_pos_enc_project", d, d + 2)
    s.linear(g + "final_proj", d, d)
    s.ln(g + "norm", d)
    s.ln(g + "img_pre_norm", d)
    for i in range(3):
        enc_layer(g + f"encode.{i}.")
    s.ln(g + "encode_norm", d)

    for i in range(6):
        enc_layer(f"transformer.encoder.layers.{i}.")
    t = "transformer.decoder."
    for i in range(6):
        q = t + f"layers.{i}."
        mha(q + "cross_attn.")
        s.ln(q + "norm1", d)
        mha(q + "ca_text.")
        s.ln(q + "catext_norm", d)
        mha(q + "self_attn.")
        s.ln(q + "norm2", d)
        s.linear(q + "linear1", ff, d)
        s[q + "linear2.weight"] = ((d, ff), "linear_res")
        s[q + "linear2.bias"] = ((d,), "bias")
        s.ln(q + "norm3", d)
    s.ln(t + "norm", d)
    mlp(t + "bbox_embed.", [d, d, d, 4])
    s[t + "query_embed.weight"] = ((200, d), "embed")
    s[t + "reference_points.weight"] = ((200, 4), "gauss")
    mlp(t + "boxRPB_embed_x.", [2, d, 8])
    mlp(t + "boxRPB_embed_y.", [2, d, 8])
    s[t + "presence_token.weight"] = ((1, d), "embed")
    mlp(t + "presence_token_head.", [d, d, d, 1])
    s.ln(t + "presence_token_out_norm", d)
    mlp(t + "ref_point_head.", [2 * d, d, d])

    h = "segmentation_head."
    for i in range(3):
        s.conv(h + f"pixel_decoder.conv_layers.{i}", d, d, 3, bias=True)
        s.ln(h + f"pixel_decoder.norms.{i}", d)
    mlp(h + "mask_predictor.mask_embed.", [d, d, d, d])
    mha(h + "cross_attend_prompt.")
    s.ln(h + "cross_attn_norm", d)
    s.conv(h + "semantic_seg_head", 1, d, 1, bias=True)
    s.conv(h + "instance_seg_head", d, d, 1, bias=True)

    p = "dot_prod_scoring."
    s.linear(p + "prompt_mlp.layers.0", ff, d)
    s.linear(p + "prompt_mlp.layers.1", d, ff)
    s.ln(p + "prompt_mlp.out_norm", d)
    s.linear(p + "prompt_proj", d, d)
    s.linear(p + "hs_proj", d, d)
    return s


def synthetic_pcs_state_dict(seed: int = 0):
    return init_state_dict(pcs_schema(), seed)


def image_path_schema(backbone_type: str = "efficientvit", model_name: str = "b1",
                      enable_inst_interactivity: bool = True) -> _Schema:
    """All tensors read by set_image + predict_inst for a student model."""
    s = _Schema()
    if backbone_type == "efficientvit":
        s.update(efficientvit_schema(model_name))
        s.update(student_head_schema(EFFICIENTVIT_CFG[model_name][0][-1]))
    elif backbone_type == "repvit":
        model_name = model_name.replace("_", ".")
        s.update(repvit_schema(model_name))
        s.update(student_head_schema(repvit_out_channels(model_name)))
    elif backbone_type == "tinyvit":
        s.update(tinyvit_schema(model_name))
        s.update(student_head_schema(TINYVIT_CFG[model_name][0][-1]))
    elif backbone_type == "sam3":  # ViT-H teacher (build_sam3_image_model): no student head
        s.update(vit_schema())
    else:
        raise NotImplementedError(f"backbone_type={backbone_type!r}")
    s.update(neck_schema("convs"))
    if enable_inst_interactivity:
        s.update(neck_schema("sam2_convs"))
        s.update(sam_heads_schema())
    return s


# ----------------------------------------------------------------------------
# seeded "realistic" initialiser
# ----------------------------------------------------------------------------
# Which activation follows a conv decides its variance-preserving gain: layers feeding
# Hardswish / GELU / ReLU get ~2/fan_in, linear layers 1/fan_in.  BatchNorms that close a
# residual branch get a small gamma (as trained networks have) so that residual stacks do
# not blow the activation scale up.
_ACT_FOLLOWS = ("inverted_conv.conv", "depth_conv.conv", "input_stem.op_list.0.conv", "features.0.0.c",
                "channel_mixer.m.0.c", "token_mixer.1.fc1", "patch_embed.seq.0.c", "conv1.c", "conv2.c", "convffn.fc1",
                "dconv_2x2_0", "output_upscaling.0", "output_upscaling.3", "mask_downscaling.0",
                "mask_downscaling.3")


def _conv_gain(name: str) -> float:
    if any(k in name for k in _ACT_FOLLOWS):
        return 2.0
    if "head.0." in name:
        return 1.0
    return 1.0


def _linear_gain(name: str) -> float:
    if "mlp.lin1" in name or ".layers.0." in name or ".layers.1." in name or "mlp.fc1" in name \
            or "pre_norm_ffn.1." in name:
        return 2.0  # followed by ReLU
    return 1.0


def _is_residual_branch_end(name: str) -> bool:
    return ("point_conv.norm" in name or "proj.norm" in name) and "stages.0.op_list.0." not in name \
        and "stages.1.op_list.0." not in name


def init_state_dict(schema: _Schema, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic fp32 CPU state dict for ``schema``.

    One ``torch.Generator`` seeded with ``seed`` is consumed in schema order, so
    the result depends only on (schema, seed, torch's CPU Philox/MT stream).
    Weights are fan-in scaled so that every layer roughly preserves variance.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def randn(shape):
        return torch.randn(shape, generator=g, dtype=torch.float32)

    def rand(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    for name, (shape, kind) in schema.items():
        if kind in ("conv", "dw"):
            fan_in = shape[1] * shape[2] * shape[3]
            t = randn(shape) * math.sqrt(_conv_gain(name) / fan_in)
        elif kind == "convT":
            # ConvTranspose2d weight is (cin, cout, k, k); with k == stride each
            # output pixel sees exactly cin inputs.
            t = randn(shape) * math.sqrt(_conv_gain(name) / shape[0])
        elif kind == "linear":
            t = randn(shape) * math.sqrt(_linear_gain(name) / shape[1])
        elif kind == "linear_res":  # last linear of a transformer residual branch
            t = randn(shape) * math.sqrt(0.03 / shape[1])
        elif kind == "ln_w_small":
            t = rand(shape, 0.3, 0.5)
        elif kind == "pos_small":   # absolute position table
            t = randn(shape) * 0.2
        elif kind == "layer_scale":  # MobileCLIP RepMixer layer scales (trained values are O(0.1))
            t = rand(shape, 0.1, 0.5)
        elif kind == "attn_bias":   # TinyViT relative-offset attention biases
            t = randn(shape) * 0.5
        elif kind == "bias":
            t = randn(shape) * 0.1
        elif kind == "bn_w":
            t = rand(shape, 0.25, 0.6) if _is_residual_branch_end(name) else rand(shape, 0.7, 1.3)
        elif kind == "dw1":      # RepVGGDW's per-channel 1x1 branch next to the identity
            t = randn(shape) * 0.3
        elif kind == "bn_w_repdw":  # BN over (dw3x3 + dw1x1 + identity): ~2x the input variance comes in
            t = rand(shape, 0.45, 0.72)
        elif kind == "bn_w_res":    # last BN of a residual branch in a 24-block stack
            t = rand(shape, 0.1, 0.25)
        elif kind == "se_bias":     # gates mostly open, as in trained networks
            t = randn(shape) * 0.3 + 2.0
        elif kind == "bn_b":
            t = randn(shape) * 0.1
        elif kind == "bn_m":
            t = randn(shape) * 0.1
        elif kind == "bn_v":
            t = rand(shape, 0.5, 1.5)
        elif kind == "bn_n":
            t = torch.tensor(1000, dtype=torch.int64)
        elif kind == "ln_w":
            t = rand(shape, 0.8, 1.2)
        elif kind == "ln_b":
            t = randn(shape) * 0.05
        elif kind == "embed":
            t = randn(shape) * 0.5
        elif kind == "gauss":
            t = randn(shape)
        else:  # pragma: no cover
            raise ValueError(kind)
        sd[name] = t
    return sd


def synthetic_state_dict(backbone_type: str = "efficientvit", model_name: str = "b1",
                         seed: int = 0, enable_inst_interactivity: bool = True):
    return init_state_dict(
        image_path_schema(backbone_type, model_name, enable_inst_interactivity), seed)


def shapes(schema: _Schema) -> Dict[str, Shape]:
    return {k: v[0] for k, v in schema.items()}


def param_count(schema: _Schema) -> int:
    n = 0
    for shape, kind in schema.values():
        if kind != "bn_n":
            n += int(math.prod(shape)) if shape else 1
    return n
