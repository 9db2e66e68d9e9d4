"""Box format helpers the reference's callers import next to the predictor (``sam3.model.box_ops``,
sam3/sam3/model/box_ops.py:11-45,47-58,91-117; used by efficientsam3_image_predictor_example.py:29 to turn an XYWH
box into the normalised CXCYWH box ``Sam3Processor.add_geometric_prompt`` takes).  Plain tensor arithmetic, any
leading dimensions, last dimension 4."""
from __future__ import annotations

import torch


def _cat(*cols):
    return torch.stack(cols, dim=-1)


def box_cxcywh_to_xyxy(x: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = x.unbind(-1)
    return _cat(cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h)


def box_cxcywh_to_xywh(x: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = x.unbind(-1)
    return _cat(cx - 0.5 * w, cy - 0.5 * h, w, h)


def box_xywh_to_xyxy(x: torch.Tensor) -> torch.Tensor:
    x0, y0, w, h = x.unbind(-1)
    return _cat(x0, y0, x0 + w, y0 + h)


def box_xywh_to_cxcywh(x: torch.Tensor) -> torch.Tensor:
    x0, y0, w, h = x.unbind(-1)
    return _cat(x0 + 0.5 * w, y0 + 0.5 * h, w, h)


def box_xyxy_to_xywh(x: torch.Tensor) -> torch.Tensor:
    x0, y0, x1, y1 = x.unbind(-1)
    return _cat(x0, y0, x1 - x0, y1 - y0)


def box_xyxy_to_cxcywh(x: torch.Tensor) -> torch.Tensor:
    x0, y0, x1, y1 = x.unbind(-1)
    return _cat((x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0)


def box_area(boxes: torch.Tensor) -> torch.Tensor:
    """Area of XYXY boxes."""
    x0, y0, x1, y1 = boxes.unbind(-1)
    return (x1 - x0) * (y1 - y0)


def box_iou(boxes1: torch.Tensor, boxes2: torch.Tensor):
    """Pairwise IoU and union of XYXY boxes [N, 4] x [M, 4] -> ([N, M], [N, M])."""
    a1, a2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[None, :, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2[None, :] - inter
    return inter / union, union
