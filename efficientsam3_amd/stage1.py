"""Stage-1 image distillation, forward pieces on the device: the loss between a student embedding
and the saved teacher embedding (stage1/train_image_encoder_stage1.py:271-307) and the teacher-embedding payload
(stage1/save_embedding_image_stage1.py:92-96: int32 augmentation seed ‖ fp16 [C, H, W]).  The trunks that produce the
embeddings are the engine's encoders (``engine.encode(..., want_trunk=True)``); the backward pass is not built."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def valid_mask(img_size: int, sizes_before_pad: Sequence[Tuple[int, int]], target_hw: Tuple[int, int]) -> np.ndarray:
    """build_valid_mask in closed form: the indicator of the (h, w) rectangle is separable, so its bilinear resize
    (align_corners=False, no antialias) is the product of two 1-D resizes -> uint8 [B, H*W] (1 = valid)."""
    def axis(n_valid: int, n_out: int) -> np.ndarray:
        scale = img_size / n_out
        src = np.maximum((np.arange(n_out, dtype=np.float32) + 0.5) * np.float32(scale) - 0.5, 0.0).astype(np.float32)
        i0 = np.minimum(np.floor(src).astype(np.int64), img_size - 1)
        i1 = np.minimum(i0 + 1, img_size - 1)
        lam = (src - i0).astype(np.float32)
        return (1.0 - lam) * (i0 < n_valid) + lam * (i1 < n_valid)
    th, tw = target_hw
    out = np.zeros((len(sizes_before_pad), th * tw), np.uint8)
    for i, (h, w) in enumerate(sizes_before_pad):
        out[i] = (np.outer(axis(h, th), axis(w, tw)).astype(np.float32) > 0.5).reshape(-1)
    return out


def distill_loss(preds: torch.Tensor, teacher: torch.Tensor, valid: torch.Tensor):
    """preds [B, HW, C] fp32 / bf16 and teacher [B, HW, C] fp32 / bf16 / fp16 (token-major, on the GPU),
    valid uint8 [B, HW] -> (masked_mse, masked_cosine_loss, per_image [B, 2])."""
    assert preds.is_cuda and teacher.is_cuda and valid.is_cuda
    assert preds.dim() == 3 and preds.shape == teacher.shape and preds.dtype in (torch.float32, torch.bfloat16)
    assert teacher.dtype in _DT and valid.dtype == torch.uint8 and tuple(valid.shape) == tuple(preds.shape[:2])
    b, hw, c = preds.shape
    preds, teacher, valid = preds.contiguous(), teacher.contiguous(), valid.contiguous()
    per = torch.empty((b, 2), dtype=torch.float32, device=preds.device)
    scratch = torch.empty((b * hw * 2,), dtype=torch.float32, device=preds.device)
    lib = _lib.load()
    with torch.cuda.device(preds.device):
        _lib.check(lib.esam3_distill_loss(_DT[preds.dtype], preds.data_ptr(), _DT[teacher.dtype], teacher.data_ptr(),
                                          valid.data_ptr(), b, hw, c, per.data_ptr(), scratch.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "esam3_distill_loss")
    m = per.mean(dim=0)
    return m[0], m[1], per


def distill_loss_backward(preds: torch.Tensor, teacher: torch.Tensor, valid: torch.Tensor, cosine_weight: float = 0.0,
                          grad_scale: float = 1.0) -> torch.Tensor:
    """dL/dpreds of ``masked_mse + cosine_weight * masked_cosine_loss`` times ``grad_scale`` (= 1 / ACCUMULATION_STEPS in
    stage1/train_image_encoder_stage1.py:186-210): [B, HW, C] in the dtype of ``preds``, zero at masked pixels -- the tensor
    ``loss.backward()`` hands to the student trunk."""
    assert preds.is_cuda and teacher.is_cuda and valid.is_cuda
    assert preds.dim() == 3 and preds.shape == teacher.shape and preds.dtype in (torch.float32, torch.bfloat16)
    assert teacher.dtype in _DT and valid.dtype == torch.uint8 and tuple(valid.shape) == tuple(preds.shape[:2])
    b, hw, c = preds.shape
    preds, teacher, valid = preds.contiguous(), teacher.contiguous(), valid.contiguous()
    grad = torch.empty_like(preds)
    scratch = torch.empty((b,), dtype=torch.float32, device=preds.device)
    lib = _lib.load()
    with torch.cuda.device(preds.device):
        _lib.check(lib.esam3_distill_loss_backward(_DT[preds.dtype], preds.data_ptr(), _DT[teacher.dtype], teacher.data_ptr(),
                                                   valid.data_ptr(), b, hw, c, float(cosine_weight), float(grad_scale),
                                                   grad.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "esam3_distill_loss_backward")
    return grad


def pack_embedding(seed: int, embedding_chw: np.ndarray) -> bytes:
    return np.int32(seed).tobytes() + np.ascontiguousarray(embedding_chw, dtype=np.float16).tobytes()


def unpack_embedding(payload: bytes, shape_chw: Tuple[int, int, int]):
    """-> (seed, fp16 [C, H, W]) as dataset_wrapper.py:50-62 parses it."""
    seed = int(np.frombuffer(payload[:4], dtype=np.int32)[0])
    n = int(np.prod(shape_chw))
    if len(payload) < 4 + 2 * n:
        raise ValueError(f"payload of {len(payload)} bytes is too short for an embedding of shape {shape_chw}")
    return seed, np.frombuffer(payload[4:4 + 2 * n], dtype=np.float16).copy().reshape(shape_chw)


def paired_forward(teacher, student, images: torch.Tensor, sizes_before_pad: Sequence[Tuple[int, int]]):
    """BASELINE config 5, forward half: the ViT-H teacher trunk and a student trunk on the same normalised images
    [B, 3, 1008, 1008], then the distillation loss between the two [B, 1024, 72, 72] embeddings
    (stage1/model.py:237 ``model.backbone.vision_backbone.trunk(x)`` on both sides,
    stage1/train_image_encoder_stage1.py:196-212).  ``teacher`` / ``student`` are models built by
    ``build_sam3_image_model`` / ``build_efficientsam3_image_model`` on the same device.
    -> dict(mse, cosine, per_image [B, 2], teacher [B, 5184, 1024], student [B, 5184, 1024])."""
    x = images.to(student.device, torch.float32).contiguous()
    t = teacher.engine.encode(x, want_sam3=False, want_sam2=False, want_trunk=True)["trunk"]
    s = student.engine.encode(x, want_sam3=False, want_sam2=False, want_trunk=True)["trunk"]
    b, h, w, c = s.shape
    assert tuple(t.shape) == (b, h, w, c), (t.shape, s.shape)
    valid = torch.from_numpy(valid_mask(x.shape[-1], sizes_before_pad, (h, w))).to(x.device)
    tt, ss = t.reshape(b, h * w, c), s.reshape(b, h * w, c)
    mse, cos, per = distill_loss(ss, tt, valid)
    return {"mse": mse, "cosine": cos, "per_image": per, "teacher": tt, "student": ss, "valid": valid}
