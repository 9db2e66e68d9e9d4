"""Shim roi_align: torchvision semantics (aligned=False, sampling_ratio=-1 adaptive).

Only needed so geometry_encoders imports and the zero-box call returns an empty
tensor; the >=1-box path is implemented per torchvision's documented algorithm
but is NOT pinned against upstream (parity unpinned for box-geometry prompts).
"""
import math

import torch
import torch.nn as nn


def _bilinear(feat, y, x):
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return feat.new_zeros(C)
    y = max(y, 0.0)
    x = max(x, 0.0)
    y0, x0 = int(y), int(x)
    if y0 >= H - 1:
        y0 = y1 = H - 1
        y = float(y0)
    else:
        y1 = y0 + 1
    if x0 >= W - 1:
        x0 = x1 = W - 1
        x = float(x0)
    else:
        x1 = x0 + 1
    ly, lx = y - y0, x - x0
    hy, hx = 1.0 - ly, 1.0 - lx
    return (hy * hx * feat[:, y0, x0] + hy * lx * feat[:, y0, x1]
            + ly * hx * feat[:, y1, x0] + ly * lx * feat[:, y1, x1])


def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    if isinstance(boxes, (list, tuple)):
        rois = []
        for i, b in enumerate(boxes):
            idx = torch.full((b.shape[0], 1), float(i), dtype=b.dtype, device=b.device)
            rois.append(torch.cat([idx, b], dim=1))
        rois = torch.cat(rois, dim=0) if rois else input.new_zeros((0, 5))
    else:
        rois = boxes
    ph, pw = output_size
    K = rois.shape[0]
    C = input.shape[1]
    out = input.new_zeros((K, C, ph, pw))
    off = 0.5 if aligned else 0.0
    for k in range(K):
        b = int(rois[k, 0].item())
        x1, y1, x2, y2 = [float(v) * spatial_scale - off for v in rois[k, 1:5]]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph, rw / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        cnt = max(gh * gw, 1)
        for i in range(ph):
            for j in range(pw):
                acc = input.new_zeros(C)
                for iy in range(gh):
                    yy = y1 + i * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        xx = x1 + j * bw + (ix + 0.5) * bw / gw
                        acc = acc + _bilinear(input[b], yy, xx)
                out[k, :, i, j] = acc / cnt
    return out


class RoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=False):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.aligned = aligned

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale,
                         self.sampling_ratio, self.aligned)
