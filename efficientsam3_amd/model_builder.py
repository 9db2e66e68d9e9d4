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


def resolve_bpe_path(bpe_path=None, required: bool = False):
    """The CLIP BPE merge table: explicit argument, else $ESAM3_BPE_PATH, else ``assets/bpe_simple_vocab_16e6.txt.gz``
    next to this package or next to an importable reference ``sam3`` package (the reference's own default,
    sam3/sam3/model_builder.py:676-680).  An explicit path must exist; otherwise None is returned when nothing is found
    (``required=False``) so that models fed with token ids can still be built."""
    import os
    if bpe_path is not None:
        if not os.path.exists(bpe_path):
            raise FileNotFoundError(f"bpe_path {bpe_path!r} does not exist (the reference ships it as "
                                    "assets/bpe_simple_vocab_16e6.txt.gz)")
        return bpe_path
    cands = [os.environ.get("ESAM3_BPE_PATH"),
             os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "bpe_simple_vocab_16e6.txt.gz")]
    try:
        import importlib.util
        spec = importlib.util.find_spec("sam3")
        for loc in (list(spec.submodule_search_locations) if spec and spec.submodule_search_locations else []):
            cands += [os.path.join(loc, "..", "assets", "bpe_simple_vocab_16e6.txt.gz"),
                      os.path.join(loc, "assets", "bpe_simple_vocab_16e6.txt.gz")]
    except (ImportError, ValueError):
        pass
    for c in cands:
        if c and os.path.exists(c):
            return os.path.abspath(c)
    if required:
        raise FileNotFoundError("no CLIP BPE merge table found: pass bpe_path= or set ESAM3_BPE_PATH "
                                "(the reference ships it as assets/bpe_simple_vocab_16e6.txt.gz)")
    return None


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
    tracing compiler; the engine *is* the compiled graph).  ``text_encoder_type`` (MobileCLIP-S0 / -S1 / -B,
    MobileCLIP2-S0/S2/S3/S4/L) adds the student text encoder (``model.backbone.language_backbone`` /
    ``forward_text``) and the PCS text-grounding detector (``Sam3Processor.set_text_prompt``,
    ``add_geometric_prompt``); ``bpe_path`` is the CLIP merge table the reference ships as
    ``assets/bpe_simple_vocab_16e6.txt.gz`` -- resolved here, at build time (``resolve_bpe_path``): an explicit path
    that does not exist raises, no path at all leaves the token-id entry points (``language_backbone.encode_tokens``,
    ``engine.encode_text``) usable and makes string prompts fail with the same message.
    ``load_state_dict`` is stricter than the reference's ``strict=False`` load: a key the selected graph needs and the
    checkpoint lacks raises KeyError (the reference warns and keeps the random initialisation).
    ``dtype``: "bf16" (throughput) or "f32" (validation: exact-f32 MFMA).
    ``fuse_linear_chains``: compose the neck's ConvT->1x1 and 3x3->conv_s0/s1 weight chains at
    load time (exact algebra, same outputs, fewer FLOPs); False runs the reference's layer list.
    """
    if efficientvit_model is not None:
        backbone_type, model_name = "efficientvit", efficientvit_model
    if str(device).startswith("cpu"):
        raise RuntimeError("EfficientSAM3-AMD has no CPU path; pass a HIP device (device='cuda')")
    if text_encoder_type is not None:
        bpe_path = resolve_bpe_path(bpe_path)
    model = Sam3Image(backbone_type, model_name, bool(enable_inst_interactivity), dtype=dtype,
                      device=device, dual_neck=dual_neck, fuse_linear_chains=fuse_linear_chains,
                      text_encoder_type=text_encoder_type, text_encoder_context_length=text_encoder_context_length,
                      bpe_path=bpe_path)
    if state_dict is None and checkpoint_path is not None:
        from .checkpoint import load_state_dict_file
        state_dict = _clean_checkpoint_keys(load_state_dict_file(checkpoint_path), bool(enable_inst_interactivity))
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
    SAM heads as the students, on the HIP engine.  With ``text_encoder_type`` (the LiteText students, e.g.
    "MobileCLIP-S0") the text encoder and the PCS grounding detector run as well (BASELINE config 4); the 354 M
    teacher text encoder (``text_encoder_type=None`` with ``enable_text_encoder``) is out of scope."""
    if str(device).startswith("cpu"):
        raise RuntimeError("EfficientSAM3-AMD has no CPU path; pass a HIP device (device='cuda')")
    if not enable_vision_encoder:
        raise NotImplementedError("enable_vision_encoder=False")
    if text_encoder_type is not None:
        bpe_path = resolve_bpe_path(bpe_path)
    model = Sam3Image("sam3", "vit_h", bool(enable_inst_interactivity), dtype=dtype, device=device, dual_neck=dual_neck,
                      fuse_linear_chains=fuse_linear_chains, text_encoder_type=text_encoder_type,
                      text_encoder_context_length=text_encoder_context_length, bpe_path=bpe_path)
    if state_dict is None and checkpoint_path is not None:
        from .checkpoint import load_state_dict_file
        state_dict = _clean_checkpoint_keys(load_state_dict_file(checkpoint_path), bool(enable_inst_interactivity))
    if state_dict is None:
        state_dict = schema.synthetic_state_dict("sam3", "vit_h", seed=synthetic_seed,
                                                 enable_inst_interactivity=bool(enable_inst_interactivity))
        if text_encoder_type is not None:
            state_dict.update(schema.synthetic_text_state_dict(text_encoder_type, text_encoder_context_length,
                                                               seed=synthetic_seed))
            state_dict.update(schema.synthetic_pcs_state_dict(seed=synthetic_seed))
    model.load_state_dict(state_dict, strict=False)
    return model
