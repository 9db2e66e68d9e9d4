"""Shim of transforms.v2 as used by sam3_image_processor.py:24-31,57-58.

v2.Resize on uint8 uses an antialiased bilinear kernel upstream; that path is
NOT pinned here (parity unpinned) -- golden vectors are generated from inputs
that are already 1008x1008 so Resize short-circuits exactly as upstream does.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional  # noqa: F401


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToDtype:
    def __init__(self, dtype, scale=False):
        self.dtype = dtype
        self.scale = scale

    def __call__(self, x):
        if x.dtype == self.dtype:
            return x
        if self.dtype == torch.uint8:
            if x.is_floating_point():
                return (x * 255.999).to(torch.uint8) if self.scale else x.to(torch.uint8)
            return x.to(torch.uint8)
        if self.dtype.is_floating_point:
            if x.dtype == torch.uint8 and self.scale:
                return x.to(self.dtype) / 255.0
            return x.to(self.dtype)
        return x.to(self.dtype)


class Resize:
    def __init__(self, size, antialias=True):
        self.size = tuple(size)

    def __call__(self, x):
        if tuple(x.shape[-2:]) == self.size:
            return x
        was_u8 = x.dtype == torch.uint8
        y = F.interpolate(x[None].float(), size=self.size, mode="bilinear",
                          align_corners=False, antialias=True)[0]
        if was_u8:
            y = y.round().clamp(0, 255).to(torch.uint8)
        return y


class Normalize:
    def __init__(self, mean, std):
        self.mean = mean
        self.std = std

    def __call__(self, x):
        mean = torch.tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        std = torch.tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - mean) / std
