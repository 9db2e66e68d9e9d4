"""``build_efficientsam3_image_model`` with the reference's signature
(sam3/sam3/model_builder.py:944-1053); extra keyword-only arguments select the engine's
activation precision and, because no checkpoint exists offline, a seeded synthetic
state dict (the reference likewise leaves the model randomly initialised when
``checkpoint_path`` is None)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import schema
from .sam3_image import Sam3Image

# eval/eval_coco.py:158-162 size aliases
SIZE_ALIASES = {
    "efficientvit": {"s": "b0", "m": "b1", "l": "b2"},
    "repvit": {"s": "m0.9", "m": "m1.1", "l": "m2.3"},
    "tinyvit": {"s": "5m", "m": "11m", "l": "21m"},
}


def _clean_checkpoint_keys(ckpt: Dict[str, torch.Tensor], interactive: bool) -> Dict[str, torch.Tensor]:
    """Key remap rules of _load_checkpoint (model_builder.py:584-630), see checkpoint.clean_checkpoint_keys."""
    from .checkpoint import clean_checkpoint_keys
    return clean_checkpoint_keys(ckpt, interactive)


def build_efficientsam3_image_model(
    bpe_path=None,
    device="cuda",
    eval_mode=True,
    checkpoint_path=None,
    load_from_HF=False,
    enable_segmentation=True,
    enable_inst_interactivity=False,
    compile=False,
    backbone_type="efficientvit",
    model_name="b0",
    efficientvit_model=None,
    text_encoder_type=None,
    text_encoder_context_length=77,
    *,
    dtype: str = "bf16",
    state_dict: Optional[Dict[str, torch.Tensor]] = None,
    synthetic_seed: int = 0,
    dual_neck: bool = True,
    fuse_linear_chains: bool = True,
) -> Sam3Image:
    """Build an EfficientSAM3 image model whose encode/decode run as HIP kernels on MI355X.

    Reference arguments keep their meaning.  ``compile`` is accepted and ignored (there is no
    tracing compiler; the engine *is* the compiled graph).  ``text_encoder_type="MobileCLIP-S0"`` adds
    the student text encoder (``model.backbone.language_backbone`` / ``forward_text``; ``bpe_path`` is
    the reference's merge table); ``enable_segmentation`` configures the PCS grounding head, which
    this build does not run yet.
    ``dtype``: "bf16" (throughput) or "f32" (validation: exact-f32 MFMA).
    ``fuse_linear_chains``: compose the neck's ConvT->1x1 and 3x3->conv_s0/s1 weight chains at
    load time (exact algebra, same outputs, fewer FLOPs); False runs the reference's layer list.
    """
    if efficientvit_model is not None:
        backbone_type, model_name = "efficientvit", efficientvit_model
    if str(device).startswith("cpu"):
        raise RuntimeError("EfficientSAM3-AMD has no CPU path; pass a HIP device (device='cuda')")
    model = Sam3Image(backbone_type, model_name, bool(enable_inst_interactivity), dtype=dtype,
                      device=device, dual_neck=dual_neck, fuse_linear_chains=fuse_linear_chains,
                      text_encoder_type=text_encoder_type, text_encoder_context_length=text_encoder_context_length,
                      bpe_path=bpe_path)
    if state_dict is None and checkpoint_path is not None:
        with open(checkpoint_path, "rb") as f:
            ckpt = torch.load(f, map_location="cpu", weights_only=True)
        state_dict = _clean_checkpoint_keys(ckpt, bool(enable_inst_interactivity))
    if state_dict is None:
        state_dict = schema.synthetic_state_dict(backbone_type, model_name, seed=synthetic_seed,
                                                 enable_inst_interactivity=bool(enable_inst_interactivity))
        if text_encoder_type is not None:
            state_dict.update(schema.synthetic_text_state_dict(text_encoder_type, text_encoder_context_length,
                                                               seed=synthetic_seed))
            state_dict.update(schema.synthetic_pcs_state_dict(seed=synthetic_seed))
    model.load_state_dict(state_dict, strict=False)
    return model


def build_sam3_image_model(
    bpe_path=None,
    device="cuda",
    eval_mode=True,
    checkpoint_path=None,
    load_from_HF=False,
    enable_segmentation=True,
    enable_inst_interactivity=False,
    compile=False,
    enable_text_encoder=True,
    enable_vision_encoder=True,
    text_encoder_type=None,
    text_encoder_context_length=77,
    *,
    dtype: str = "bf16",
    state_dict: Optional[Dict[str, torch.Tensor]] = None,
    synthetic_seed: int = 0,
    dual_neck: bool = True,
    fuse_linear_chains: bool = True,
) -> Sam3Image:
    """``build_sam3_image_model`` (model_builder.py:643-750): the ViT-H teacher trunk + the same dual neck and
    SAM heads as the students.  The image path (set_image / predict_inst) runs on the HIP engine; the 354 M
    teacher text encoder and the PCS grounding head are not built (``text_encoder_type="MobileCLIP-S0"`` gives
    the LiteText student encoder)."""
    if str(device).startswith("cpu"):
        raise RuntimeError("EfficientSAM3-AMD has no CPU path; pass a HIP device (device='cuda')")
    if not enable_vision_encoder:
        raise NotImplementedError("enable_vision_encoder=False")
    model = Sam3Image("sam3", "vit_h", bool(enable_inst_interactivity), dtype=dtype, device=device, dual_neck=dual_neck,
                      fuse_linear_chains=fuse_linear_chains, text_encoder_type=text_encoder_type,
                      text_encoder_context_length=text_encoder_context_length, bpe_path=bpe_path)
    if state_dict is None and checkpoint_path is not None:
        with open(checkpoint_path, "rb") as f:
            ckpt = torch.load(f, map_location="cpu", weights_only=True)
        state_dict = _clean_checkpoint_keys(ckpt, bool(enable_inst_interactivity))
    if state_dict is None:
        state_dict = schema.synthetic_state_dict("sam3", "vit_h", seed=synthetic_seed,
                                                 enable_inst_interactivity=bool(enable_inst_interactivity))
        if text_encoder_type is not None:
            state_dict.update(schema.synthetic_text_state_dict(text_encoder_type, text_encoder_context_length,
                                                               seed=synthetic_seed))
            state_dict.update(schema.synthetic_pcs_state_dict(seed=synthetic_seed))
    model.load_state_dict(state_dict, strict=False)
    return model
