"""Geometric prompt container of the PCS detector: the host-side mirror of ``Prompt``
(sam3/sam3/model/geometry_encoders.py:82-400) for what ``Sam3Processor.add_geometric_prompt`` /
``add_point_prompt`` build: right-padded point and box sequences with positive / negative labels.

Same tensor conventions as the reference (sequence first): ``box_embeddings [Nb, B, 4]`` normalised cxcywh,
``box_labels [Nb, B]``, ``box_mask [B, Nb]`` (True = padding), and the same for points (``[Np, B, 2]`` xy).
"""
from __future__ import annotations

from typing import Optional

import torch


def _concat_padded(seq1, mask1, seq2, mask2):
    """Two right-padded sequences -> one right-padded sequence (geometry_encoders.py:22-79)."""
    l1, b = seq1.shape[:2]
    l2 = seq2.shape[0]
    n1, n2 = (~mask1).sum(-1), (~mask2).sum(-1)
    out = torch.zeros((l1 + l2, b) + tuple(seq2.shape[2:]), dtype=seq2.dtype, device=seq2.device)
    out[:l1] = seq1
    for i in range(b):
        out[int(n1[i]):int(n1[i]) + l2, i] = seq2[:, i]
    mask = torch.arange(l1 + l2, device=seq2.device)[None, :] >= (n1 + n2)[:, None]
    return out, mask


_NO_MASK_PROMPTS = ("mask prompts of the geometry encoder (SequenceGeometryEncoder._encode_masks, sam3/model/geometry_encoders.py:697-745, "
                    "815-823) are not built: Sam3Processor never issues them (it appends boxes and points only, "
                    "sam3_image_processor.py:130-190); pass boxes / points, or use the interactive predictor's mask_input")


class Prompt:
    def __init__(self, box_embeddings: Optional[torch.Tensor] = None, box_mask: Optional[torch.Tensor] = None,
                 point_embeddings: Optional[torch.Tensor] = None, point_mask: Optional[torch.Tensor] = None,
                 box_labels: Optional[torch.Tensor] = None, point_labels: Optional[torch.Tensor] = None,
                 mask_embeddings: Optional[torch.Tensor] = None, mask_mask: Optional[torch.Tensor] = None,
                 mask_labels: Optional[torch.Tensor] = None):
        if mask_embeddings is not None or mask_mask is not None or mask_labels is not None:
            raise NotImplementedError(_NO_MASK_PROMPTS)
        ref = box_embeddings if box_embeddings is not None else point_embeddings
        if ref is None:
            raise ValueError("Prompt needs box_embeddings or point_embeddings (use zero-length tensors for none)")
        b, dev = ref.shape[1], ref.device
        self.box_embeddings = box_embeddings if box_embeddings is not None else torch.zeros(0, b, 4, device=dev)
        self.point_embeddings = point_embeddings if point_embeddings is not None else torch.zeros(0, b, 2, device=dev)
        nb, np_ = self.box_embeddings.shape[0], self.point_embeddings.shape[0]
        self.box_mask = box_mask if box_mask is not None else torch.zeros(b, nb, dtype=torch.bool, device=dev)
        self.point_mask = point_mask if point_mask is not None else torch.zeros(b, np_, dtype=torch.bool, device=dev)
        self.box_labels = box_labels if box_labels is not None else torch.ones(nb, b, dtype=torch.long, device=dev)
        self.point_labels = point_labels if point_labels is not None else torch.ones(np_, b, dtype=torch.long, device=dev)
        assert self.box_embeddings.shape[-1] == 4 and self.point_embeddings.shape[-1] == 2
        assert tuple(self.box_mask.shape) == (b, nb) and tuple(self.point_mask.shape) == (b, np_)
        assert tuple(self.box_labels.shape) == (nb, b) and tuple(self.point_labels.shape) == (np_, b)

    @property
    def n_prompts(self) -> int:
        return int(self.box_embeddings.shape[0] + self.point_embeddings.shape[0])

    def append_boxes(self, boxes: torch.Tensor, labels: torch.Tensor, mask: Optional[torch.Tensor] = None) -> None:
        """boxes [n, B, 4] cxcywh in [0, 1], labels [n, B] (geometry_encoders.py:331-352)."""
        b = self.box_embeddings.shape[1]
        assert boxes.shape[1] == labels.shape[1] == b and list(boxes.shape[:2]) == list(labels.shape[:2])
        if mask is None:
            mask = torch.zeros(b, boxes.shape[0], dtype=torch.bool, device=boxes.device)
        lab, _ = _concat_padded(self.box_labels.long().unsqueeze(-1), self.box_mask, labels.long().unsqueeze(-1), mask)
        self.box_labels = lab.squeeze(-1)
        self.box_embeddings, self.box_mask = _concat_padded(self.box_embeddings, self.box_mask, boxes.float(), mask)

    def append_points(self, points: torch.Tensor, labels: torch.Tensor, mask: Optional[torch.Tensor] = None) -> None:
        """points [n, B, 2] xy in [0, 1], labels [n, B] (geometry_encoders.py:354-375)."""
        b = self.point_embeddings.shape[1]
        assert points.shape[1] == labels.shape[1] == b and list(points.shape[:2]) == list(labels.shape[:2])
        if mask is None:
            mask = torch.zeros(b, points.shape[0], dtype=torch.bool, device=points.device)
        lab, _ = _concat_padded(self.point_labels.long().unsqueeze(-1), self.point_mask, labels.long().unsqueeze(-1), mask)
        self.point_labels = lab.squeeze(-1)
        self.point_embeddings, self.point_mask = _concat_padded(self.point_embeddings, self.point_mask, points.float(), mask)

    def append_masks(self, masks, labels=None, attn_mask=None) -> None:
        """geometry_encoders.py:377-400 -- refused by name instead of being dropped silently"""
        raise NotImplementedError(_NO_MASK_PROMPTS)

    def batch_first(self) -> dict:
        """The engine's layout: points [B, Np, 2], labels [B, N] int32, masks [B, N] uint8."""
        return {"points": self.point_embeddings.transpose(0, 1).float().contiguous(),
                "point_labels": self.point_labels.transpose(0, 1).to(torch.int32).contiguous(),
                "point_mask": self.point_mask.to(torch.uint8).contiguous(),
                "boxes": self.box_embeddings.transpose(0, 1).float().contiguous(),
                "box_labels": self.box_labels.transpose(0, 1).to(torch.int32).contiguous(),
                "box_mask": self.box_mask.to(torch.uint8).contiguous()}
