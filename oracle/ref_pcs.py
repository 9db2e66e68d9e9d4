"""ORACLE (test infrastructure only): fp32 CPU restatement of the PCS text-grounding path of the reference
for the prompt ``Sam3Processor.set_text_prompt`` issues (one text, the dummy geometric prompt):
``Sam3Image.forward_grounding`` (sam3/sam3/model/sam3_image.py:442-493) = _encode_prompt
(:169-216; SequenceGeometryEncoder.forward, geometry_encoders.py:732-853, with an empty prompt) ->
TransformerEncoderFusion (encoder.py:139-201,513-577) -> TransformerDecoder (decoder.py:33-191,417-618)
-> DotProductScoring (model_misc.py:37-91) + box heads (sam3_image.py:300-375) ->
UniversalSegmentationHead (maskformer_segmentation.py:172-323), and the processor's post-processing
(sam3_image_processor.py:219-259).  Plain functions over the reference's state dict; pinned against
the real reference by oracle/gen_golden_pcs.py.  Only tests may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
D = 256
HEADS = 8


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def mlp(sd: SD, p: str, x: torch.Tensor, n: int) -> torch.Tensor:
    """model_misc.MLP (model_misc.py:160-195) without residual / out_norm."""
    for i in range(n):
        x = _lin(sd, p + f"layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def mha(sd: SD, p: str, q, k, v, key_padding_mask: Optional[torch.Tensor] = None,
        attn_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.MultiheadAttention forward, batch-first tensors [B, L, 256], 8 heads; key_padding_mask [B, Lk]
    True = ignore; attn_bias additive [B, heads, Lq, Lk]."""
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    e = q.shape[-1]
    hd = e // HEADS

    def split(t):
        return t.view(t.shape[0], t.shape[1], HEADS, hd).transpose(1, 2)

    qq = split(F.linear(q, w[:e], b[:e]))
    kk = split(F.linear(k, w[e:2 * e], b[e:2 * e]))
    vv = split(F.linear(v, w[2 * e:], b[2 * e:]))
    a = (qq @ kk.transpose(-1, -2)) * (hd ** -0.5)
    if attn_bias is not None:
        a = a + attn_bias
    if key_padding_mask is not None:
        a = a.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    o = (a.softmax(-1) @ vv).transpose(1, 2).reshape(q.shape[0], q.shape[1], e)
    return _lin(sd, p + "out_proj", o)


def encoder_layer(sd: SD, p: str, tgt, memory, mem_kpm, query_pos, mem_pos, pos_at_attn: bool,
                  pos_at_keys: bool, tgt_kpm=None) -> torch.Tensor:
    """TransformerEncoderLayer.forward_pre (encoder.py:139-201), batch-first."""
    t2 = _ln(sd, p + "norm1", tgt)
    qk = t2 + query_pos if pos_at_attn else t2
    tgt = tgt + mha(sd, p + "self_attn.", qk, qk, t2, key_padding_mask=tgt_kpm)
    t2 = _ln(sd, p + "norm2", tgt)
    tgt = tgt + mha(sd, p + "cross_attn_image.", t2, memory + mem_pos if pos_at_keys else memory, memory,
                    key_padding_mask=mem_kpm)
    t2 = _ln(sd, p + "norm3", tgt)
    return tgt + _lin(sd, p + "linear2", F.relu(_lin(sd, p + "linear1", t2)))


def geometry_dummy(sd: SD, img_feat: torch.Tensor, img_pos: torch.Tensor):
    """SequenceGeometryEncoder.forward for the dummy prompt (no points, no boxes): the sequence is the CLS
    token alone -> final_proj + norm -> 3 encoder layers that cross-attend to the image (keys carry the
    position encoding) -> encode_norm.  img_feat / img_pos: [B, HW, 256].  -> ([B, 1, 256], mask [B, 1])."""
    g = "geometry_encoder."
    b = img_feat.shape[0]
    x = sd[g + "cls_embed.weight"].view(1, 1, D).repeat(b, 1, 1)
    x = _ln(sd, g + "norm", _lin(sd, g + "final_proj", x))
    zero = torch.zeros_like(x)
    for i in range(3):
        x = encoder_layer(sd, g + f"encode.{i}.", x, img_feat, None, zero, img_pos, False, True)
    return _ln(sd, g + "encode_norm", x), torch.zeros((b, 1), dtype=torch.bool)


def concat_padded(seq1, mask1, seq2, mask2):
    """concat_padded_sequences (geometry_encoders.py:22-79), batch-first: [B, L1, C] + [B, L2, C] right-padded
    sequences -> one right-padded [B, L1+L2, C]; masks True = padding."""
    b, l1, c = seq1.shape
    l2 = seq2.shape[1]
    n1, n2 = (~mask1).sum(-1), (~mask2).sum(-1)
    out = torch.zeros((b, l1 + l2, c), dtype=seq2.dtype)
    out[:, :l1] = seq1
    for i in range(b):
        out[i, int(n1[i]):int(n1[i]) + l2] = seq2[i]
    mask = torch.arange(l1 + l2)[None, :] >= (n1 + n2)[:, None]
    return out, mask


def roi_align_7(feat: torch.Tensor, boxes_xyxy: torch.Tensor, out: int = 7) -> torch.Tensor:
    """torchvision.ops.roi_align(feat, boxes, 7) with its defaults spatial_scale=1, sampling_ratio=-1 (adaptive:
    ceil(roi / 7) samples per bin and axis), aligned=False (roi sides clamped to >= 1) - torchvision is a
    third-party dependency absent from /root/reference; this follows its published CPU kernel
    (roi_align_kernel.cpp).  feat [B, C, H, W]; boxes_xyxy [B, Nb, 4] in feature pixels -> [B, Nb, C, 7, 7]."""
    b, c, h, w = feat.shape
    nb = boxes_xyxy.shape[1]
    res = torch.zeros((b, nb, c, out, out), dtype=feat.dtype)

    def axis(start, size, n, grid):
        # sample coordinates of the `grid` points of each of the `out` bins, then the two taps and weights
        bins = torch.arange(out, dtype=torch.float32)[:, None]
        g = torch.arange(grid, dtype=torch.float32)[None, :]
        t = start + bins * (size / out) + (g + 0.5) * (size / out) / grid      # [out, grid]
        valid = (t >= -1.0) & (t <= n)
        t = t.clamp(min=0.0)
        lo = t.floor().long()
        top = lo >= n - 1
        lo = torch.where(top, torch.full_like(lo, n - 1), lo)
        hi = torch.where(top, lo, lo + 1)
        t = torch.where(top, lo.float(), t)
        fr = t - lo.float()
        return lo, hi, fr, valid

    for i in range(b):
        for j in range(nb):
            x1, y1, x2, y2 = [float(v) for v in boxes_xyxy[i, j]]
            rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
            gh, gw = int(math.ceil(rh / out)), int(math.ceil(rw / out))
            ylo, yhi, yf, yv = axis(y1, rh, h, gh)
            xlo, xhi, xf, xv = axis(x1, rw, w, gw)
            f = feat[i]
            acc = torch.zeros((c, out, out), dtype=feat.dtype)
            for a in range(gh):
                for e in range(gw):
                    yl, yh_, fy = ylo[:, a], yhi[:, a], yf[:, a]
                    xl, xh_, fx = xlo[:, e], xhi[:, e], xf[:, e]
                    v = ((1 - fy)[None, :, None] * (1 - fx)[None, None, :] * f[:, yl][:, :, xl]
                         + (1 - fy)[None, :, None] * fx[None, None, :] * f[:, yl][:, :, xh_]
                         + fy[None, :, None] * (1 - fx)[None, None, :] * f[:, yh_][:, :, xl]
                         + fy[None, :, None] * fx[None, None, :] * f[:, yh_][:, :, xh_])
                    ok = (yv[:, a][:, None] & xv[:, e][None, :]).float()
                    acc = acc + v * ok[None]
            res[i, j] = acc / max(gh * gw, 1)
    return res


def encode_xy(x: torch.Tensor, y: torch.Tensor):
    """PositionEmbeddingSine._encode_xy (position_encoding.py:53-70), num_pos_feats 128, scale 2*pi."""
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / 128)

    def enc(v):
        pos = (v * (2 * math.pi))[..., None] / dim_t
        return torch.stack((pos[..., 0::2].sin(), pos[..., 1::2].cos()), dim=-1).flatten(-2)

    return enc(x), enc(y)


def geometry_encoder(sd: SD, img_feat, img_pos, h: int, w: int, points, point_labels, point_mask, boxes, box_labels,
                     box_mask):
    """SequenceGeometryEncoder.forward (geometry_encoders.py:732-853) as model_builder.py:270-285 configures it
    (direct + pooled + position projections for points and boxes, CLS token, final_proj + norm, 3 layers):
    points [B, Np, 2] normalised xy, boxes [B, Nb, 4] normalised cxcywh, labels [B, N] {0,1}, masks [B, N] True = pad;
    img_feat / img_pos [B, HW, 256]  ->  ([B, Np+Nb+1, 256], mask)."""
    g = "geometry_encoder."
    b = img_feat.shape[0]
    nchw = _ln(sd, g + "img_pre_norm", img_feat).transpose(1, 2).reshape(b, D, h, w)
    # points (:600-643): direct projection + bilinear sample of the normalised feature map + sine encoding
    pe = _lin(sd, g + "points_direct_project", points)
    if points.shape[1] > 0:
        grid = (points * 2 - 1).unsqueeze(2)                        # [B, Np, 1, 2]
        samp = F.grid_sample(nchw, grid, align_corners=False)       # [B, C, Np, 1]
        pe = pe + _lin(sd, g + "points_pool_project", samp.squeeze(-1).transpose(1, 2))
    ex, ey = encode_xy(points[..., 0], points[..., 1])
    pe = pe + _lin(sd, g + "points_pos_enc_project", torch.cat([ex, ey], -1))
    pe = pe + sd[g + "label_embed.weight"][point_labels.long()]
    # boxes (:645-695): direct projection + 7x7 roi_align through a 7x7 conv + sine encoding of (cy, cx, h, w)
    be = _lin(sd, g + "boxes_direct_project", boxes)
    if boxes.shape[1] > 0:
        roi = roi_align_7(nchw, box_cxcywh_to_xyxy(boxes) * torch.tensor([w, h, w, h], dtype=torch.float32))
        pooled = F.conv2d(roi.flatten(0, 1), sd[g + "boxes_pool_project.weight"], sd[g + "boxes_pool_project.bias"])
        be = be + pooled.view(b, boxes.shape[1], D)
    ex, ey = encode_xy(boxes[..., 0], boxes[..., 1])
    enc = torch.cat([ey, ex, boxes[..., 3:4], boxes[..., 2:3]], -1)
    be = be + _lin(sd, g + "boxes_pos_enc_project", enc)
    be = be + sd[g + "label_embed.weight"][box_labels.long()]
    x, mask = concat_padded(pe, point_mask, be, box_mask)
    cls = sd[g + "cls_embed.weight"].view(1, 1, D).repeat(b, 1, 1)
    x, mask = concat_padded(x, mask, cls, torch.zeros((b, 1), dtype=torch.bool))
    x = _ln(sd, g + "norm", _lin(sd, g + "final_proj", x))
    zero = torch.zeros_like(x)
    for i in range(3):
        x = encoder_layer(sd, g + f"encode.{i}.", x, img_feat, None, zero, img_pos, False, True, tgt_kpm=mask)
    return _ln(sd, g + "encode_norm", x), mask


def fusion_encoder(sd: SD, img_feat, img_pos, prompt, prompt_mask) -> torch.Tensor:
    """TransformerEncoderFusion (add_pooled_text_to_img_feat=False): 6 pre-norm layers, self-attention over
    the image tokens with the position encoding on q and k, cross-attention to the prompt tokens."""
    x = img_feat
    zero = torch.zeros_like(prompt)
    for i in range(6):
        x = encoder_layer(sd, f"transformer.encoder.layers.{i}.", x, prompt, prompt_mask, img_pos, zero, True, False)
    return x


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def sine_embed_boxes(boxes: torch.Tensor) -> torch.Tensor:
    """gen_sineembed_for_position (model_misc.py:238-275) for [..., 4] boxes -> [..., 512] (y, x, w, h)."""
    nf = D // 2
    dim_t = torch.arange(nf, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / nf)

    def enc(c):
        pos = (c * (2 * math.pi))[..., None] / dim_t
        return torch.stack((pos[..., 0::2].sin(), pos[..., 1::2].cos()), dim=-1).flatten(-2)

    return torch.cat((enc(boxes[..., 1]), enc(boxes[..., 0]), enc(boxes[..., 2]), enc(boxes[..., 3])), dim=-1)


def box_cxcywh_to_xyxy(x: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def rpb_matrix(sd: SD, ref_boxes: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """TransformerDecoder._get_rpb_matrix with boxRPB="log" (decoder.py:333-415): ref_boxes [B, Q, 4] cxcywh
    in [0,1] -> additive bias [B, 8, Q, h*w]."""
    t = "transformer.decoder."
    xyxy = box_cxcywh_to_xyxy(ref_boxes)
    ch = torch.arange(0, h, dtype=torch.float32) / h
    cw = torch.arange(0, w, dtype=torch.float32) / w
    dy = ch.view(1, 1, -1, 1) - xyxy[:, :, None, 1:4:2]  # [B, Q, h, 2]
    dx = cw.view(1, 1, -1, 1) - xyxy[:, :, None, 0:3:2]

    def logn(d):
        d = d * 8
        return torch.sign(d) * torch.log2(torch.abs(d) + 1.0) / np.log2(8)

    by = mlp(sd, t + "boxRPB_embed_y.", logn(dy), 2)  # [B, Q, h, 8]
    bx = mlp(sd, t + "boxRPB_embed_x.", logn(dx), 2)  # [B, Q, w, 8]
    bias = by[:, :, :, None, :] + bx[:, :, None, :, :]  # [B, Q, h, w, 8]
    return bias.flatten(2, 3).permute(0, 3, 1, 2).contiguous()


def decoder(sd: SD, memory, mem_pos, prompt, prompt_mask, h: int, w: int):
    """TransformerDecoder.forward at inference (no DAC): -> hs [6, B, 200, 256] (normed), reference boxes
    [6, B, 200, 4] (the boxes each layer STARTS from), presence logits [6, B, 1]."""
    t = "transformer.decoder."
    b = memory.shape[0]
    out = sd[t + "query_embed.weight"][None].repeat(b, 1, 1)
    ref = sd[t + "reference_points.weight"][None].repeat(b, 1, 1).sigmoid()
    presence = sd[t + "presence_token.weight"][None].expand(b, 1, -1)
    hs, refs, pres = [], [ref], []
    for i in range(6):
        p = t + f"layers.{i}."
        sine = sine_embed_boxes(ref)                                   # valid ratios are all ones
        qpos = mlp(sd, t + "ref_point_head.", sine, 2)
        bias = rpb_matrix(sd, ref, h, w)
        bias = torch.cat([torch.zeros_like(bias[:, :, :1]), bias], dim=2)  # the presence token sees no bias
        x = torch.cat([presence, out], dim=1)
        xpos = torch.cat([torch.zeros_like(presence), qpos], dim=1)
        qk = x + xpos
        x = _ln(sd, p + "norm2", x + mha(sd, p + "self_attn.", qk, qk, x))
        x = _ln(sd, p + "catext_norm", x + mha(sd, p + "ca_text.", x + xpos, prompt, prompt, key_padding_mask=prompt_mask))
        x = _ln(sd, p + "norm1", x + mha(sd, p + "cross_attn.", x + xpos, memory + mem_pos, memory, attn_bias=bias))
        x = _ln(sd, p + "norm3", x + _lin(sd, p + "linear2", F.relu(_lin(sd, p + "linear1", x))))
        presence, out = x[:, :1], x[:, 1:]
        normed = _ln(sd, t + "norm", out)
        ref = (mlp(sd, t + "bbox_embed.", normed, 3) + inverse_sigmoid(ref)).sigmoid()
        if i != 5:
            refs.append(ref)
        hs.append(normed)
        pres.append(mlp(sd, t + "presence_token_head.", _ln(sd, t + "presence_token_out_norm", presence), 3).squeeze(-1))
    return torch.stack(hs), torch.stack(refs), torch.stack(pres)


def dot_product_scoring(sd: SD, hs, prompt, prompt_mask) -> torch.Tensor:
    """DotProductScoring.forward (model_misc.py:66-91); prompt [B, S, 256], mask True = padding."""
    p = "dot_prod_scoring."
    x = _lin(sd, p + "prompt_mlp.layers.1", F.relu(_lin(sd, p + "prompt_mlp.layers.0", prompt))) + prompt
    x = _ln(sd, p + "prompt_mlp.out_norm", x)
    valid = (~prompt_mask).float()[..., None]
    pooled = (x * valid).sum(1) / valid.sum(1).clamp(min=1.0)
    pp = _lin(sd, p + "prompt_proj", pooled)            # [B, 256]
    ph = _lin(sd, p + "hs_proj", hs)                    # [L, B, Q, 256]
    s = torch.matmul(ph, pp[None, :, :, None]) * (1.0 / math.sqrt(D))
    return s.clamp(min=-12.0, max=12.0)


def segmentation_head(sd: SD, fpn, enc_out, prompt, prompt_mask, queries):
    """UniversalSegmentationHead.forward: fpn = sam3 backbone_fpn [B,256,288,288], [B,256,144,144], (72x72 level
    replaced by the encoder output); enc_out [B, 5184, 256]; queries [B, 200, 256] (last decoder layer)."""
    h = "segmentation_head."
    x = enc_out + mha(sd, h + "cross_attend_prompt.", _ln(sd, h + "cross_attn_norm", enc_out), prompt, prompt,
                      key_padding_mask=prompt_mask)
    b = x.shape[0]
    prev = x.transpose(1, 2).reshape(b, D, fpn[1].shape[-2] // 2, fpn[1].shape[-1] // 2)
    for li, feat in enumerate((fpn[1], fpn[0])):
        prev = feat + F.interpolate(prev, size=feat.shape[-2:], mode="nearest")
        prev = F.conv2d(prev, sd[h + f"pixel_decoder.conv_layers.{li}.weight"], sd[h + f"pixel_decoder.conv_layers.{li}.bias"],
                        padding=1)
        prev = F.relu(F.group_norm(prev, 8, sd[h + f"pixel_decoder.norms.{li}.weight"], sd[h + f"pixel_decoder.norms.{li}.bias"]))
    inst = F.conv2d(prev, sd[h + "instance_seg_head.weight"], sd[h + "instance_seg_head.bias"])
    masks = torch.einsum("bqc,bchw->bqhw", mlp(sd, h + "mask_predictor.mask_embed.", queries, 3), inst)
    sem = F.conv2d(prev, sd[h + "semantic_seg_head.weight"], sd[h + "semantic_seg_head.bias"])
    return masks, sem


def forward_grounding(sd: SD, backbone_fpn, pos72: torch.Tensor, language_features: torch.Tensor,
                      language_mask: torch.Tensor, taps: Optional[dict] = None, geo: Optional[dict] = None) -> dict:
    """backbone_fpn: the sam3 neck's three NCHW levels for the B images; pos72 [B,256,72,72] sine position
    encoding of the last level; language_features [S, B, 256] / language_mask [B, S] (one text per image);
    geo: optional geometric prompt {points, point_labels, point_mask, boxes, box_labels, box_mask} (batch-first)."""
    feat = backbone_fpn[-1]
    b, _, h, w = feat.shape
    img = feat.flatten(2).transpose(1, 2)
    pos = pos72.flatten(2).transpose(1, 2)
    txt = language_features.transpose(0, 1)
    if geo is None:
        geo, geo_mask = geometry_dummy(sd, img, pos)
    else:
        geo, geo_mask = geometry_encoder(sd, img, pos, h, w, geo["points"], geo["point_labels"], geo["point_mask"],
                                         geo["boxes"], geo["box_labels"], geo["box_mask"])
    prompt = torch.cat([txt, geo], dim=1)
    prompt_mask = torch.cat([language_mask, geo_mask], dim=1)
    memory = fusion_encoder(sd, img, pos, prompt, prompt_mask)
    hs, refs, pres = decoder(sd, memory, pos, prompt, prompt_mask, h, w)
    logits = dot_product_scoring(sd, hs, prompt, prompt_mask)
    boxes = (inverse_sigmoid(refs) + mlp(sd, "transformer.decoder.bbox_embed.", hs, 3)).sigmoid()
    masks, sem = segmentation_head(sd, backbone_fpn, memory, prompt, prompt_mask, hs[-1])
    if taps is not None:
        taps.update(prompt=prompt, memory=memory, hs=hs, refs=refs)
    return {"pred_logits": logits[-1], "pred_boxes": boxes[-1], "pred_boxes_xyxy": box_cxcywh_to_xyxy(boxes[-1]),
            "presence_logit_dec": pres[-1], "pred_masks": masks, "semantic_seg": sem}


def postprocess_grounding(out: dict, orig_hw, confidence_threshold: float = 0.5) -> dict:
    """Sam3Processor._forward_grounding after the model call (sam3_image_processor.py:227-259), batch of one."""
    probs = (out["pred_logits"].sigmoid() * out["presence_logit_dec"].sigmoid().unsqueeze(1)).squeeze(-1)
    keep = probs > confidence_threshold
    h, w = orig_hw
    boxes = box_cxcywh_to_xyxy(out["pred_boxes"][keep]) * torch.tensor([w, h, w, h], dtype=torch.float32)[None]
    ml = F.interpolate(out["pred_masks"][keep].unsqueeze(1), (h, w), mode="bilinear", align_corners=False).sigmoid()
    return {"masks_logits": ml, "masks": ml > 0.5, "boxes": boxes, "scores": probs[keep]}
