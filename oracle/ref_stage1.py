"""ORACLE (test infrastructure only): CPU restatement of the stage-1 image-distillation loss and of the saved
teacher-embedding payload.  Follows stage1/train_image_encoder_stage1.py:271-307
(build_valid_mask, masked_mse, masked_cosine_loss) and stage1/save_embedding_image_stage1.py:92-96 /
stage1/data/augmentation/dataset_wrapper.py:50-62 (payload = int32 seed bytes ‖ fp16 embedding).  Pinned against
the reference's own functions by oracle/gen_golden_stage1.py.  Only tests may import this module."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def build_valid_mask(img_size: int, sizes_before_pad: Sequence[Tuple[int, int]], target_hw: Tuple[int, int]) -> torch.Tensor:
    """1 inside the un-padded (h, w) top-left rectangle of each img_size^2 input, bilinearly resized to the
    embedding grid and thresholded at 0.5 -> [B, 1, H, W] float."""
    valid = torch.zeros(len(sizes_before_pad), 1, img_size, img_size)
    for i, (h, w) in enumerate(sizes_before_pad):
        valid[i, :, :h, :w] = 1
    valid = F.interpolate(valid, size=tuple(target_hw), mode="bilinear", align_corners=False)
    return (valid > 0.5).float()


def masked_mse(preds: torch.Tensor, teacher: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    diff = (preds - teacher) * mask
    denom = mask.sum(dim=(1, 2, 3)).clamp(min=1.0)
    return (diff.square().sum(dim=(1, 2, 3)) / denom).mean()


def masked_cosine_loss(preds: torch.Tensor, teacher: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    loss = (1.0 - F.cosine_similarity(preds, teacher, dim=1)) * mask.squeeze(1)
    denom = mask.squeeze(1).sum(dim=(1, 2)).clamp(min=1.0)
    return (loss.sum(dim=(1, 2)) / denom).mean()


def pack_embedding(seed: int, embedding_chw: np.ndarray) -> bytes:
    return np.int32(seed).tobytes() + np.ascontiguousarray(embedding_chw, dtype=np.float16).tobytes()


def unpack_embedding(payload: bytes, shape_chw: Tuple[int, int, int]):
    seed = int(np.frombuffer(payload[:4], dtype=np.int32)[0])
    n = int(np.prod(shape_chw))
    emb = np.frombuffer(payload[4:4 + 2 * n], dtype=np.float16).copy().reshape(shape_chw)
    return seed, emb
