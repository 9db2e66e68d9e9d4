import torch
import torch.nn as nn

from .roi_align import RoIAlign, roi_align  # noqa: F401


class StochasticDepth(nn.Module):
    """Identity in eval mode (row-mode stochastic depth in training)."""

    def __init__(self, p: float, mode: str):
        super().__init__()
        self.p = p
        self.mode = mode

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = 1.0 - self.p
        size = [x.shape[0]] + [1] * (x.ndim - 1) if self.mode == "row" else [1] * x.ndim
        noise = torch.empty(size, dtype=x.dtype, device=x.device).bernoulli_(keep)
        if keep > 0.0:
            noise.div_(keep)
        return x * noise


def masks_to_boxes(masks):
    if masks.numel() == 0:
        return torch.zeros((0, 4), device=masks.device, dtype=torch.float)
    n = masks.shape[0]
    out = torch.zeros((n, 4), device=masks.device, dtype=torch.float)
    for i, m in enumerate(masks):
        y, x = torch.where(m != 0)
        out[i, 0], out[i, 1], out[i, 2], out[i, 3] = x.min(), y.min(), x.max(), y.max()
    return out
