"""Shim transforms: scriptable Resize/Normalize + ToTensor (sam1_utils.py:12,31-36)."""
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional, v2  # noqa: F401


class Resize(nn.Module):
    def __init__(self, size: List[int]):
        super().__init__()
        self.size = [int(size[0]), int(size[1])]

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        if img.shape[-2] == self.size[0] and img.shape[-1] == self.size[1]:
            return img
        squeeze = img.dim() == 3
        x = img.unsqueeze(0) if squeeze else img
        x = F.interpolate(x, size=self.size, mode="bilinear", align_corners=False, antialias=True)
        return x.squeeze(0) if squeeze else x


class Normalize(nn.Module):
    def __init__(self, mean: List[float], std: List[float]):
        super().__init__()
        self.mean = [float(m) for m in mean]
        self.std = [float(s) for s in std]

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        mean = torch.tensor(self.mean, dtype=img.dtype, device=img.device).view(-1, 1, 1)
        std = torch.tensor(self.std, dtype=img.dtype, device=img.device).view(-1, 1, 1)
        return (img - mean) / std


class ToTensor:
    def __call__(self, pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x
