"""Masks -> COCO RLE on the device: the host-side mirror of ``rle_encode`` / ``robust_rle_encode``
(sam3/sam3/train/masks_ops.py:161-250) and of the ``pycocotools.mask.encode`` call of the evaluation writers
(sam3/scripts/eval/gold/eval_efficientsam3_all_subsets.py:124-135).  The run lengths come from the HIP kernels
(``esam3_rle_encode``), the compressed "counts" string from the library's host codec (``esam3_rle_to_string``);
there is no CPU fallback for the device part."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch

from . import _lib


def rle_counts_device(masks: torch.Tensor, capacity: int = 0):
    """masks [N, H, W] bool / uint8 on a GPU -> (counts uint32 ndarray, offsets int32 ndarray [N+1])."""
    assert masks.is_cuda and masks.dim() == 3, "masks must be a [N, H, W] device tensor"
    assert masks.dtype in (torch.bool, torch.uint8), "masks must have dtype bool or uint8"
    n, h, w = masks.shape
    if n == 0:
        return np.zeros((0,), np.uint32), np.zeros((1,), np.int32)
    lib = _lib.load()
    m = masks.contiguous().view(torch.uint8) if masks.dtype == torch.bool else masks.contiguous()
    cap = int(capacity) if capacity > 0 else max(1024, n * (h + w) * 8)
    dev = masks.device
    while True:
        counts = torch.empty((cap,), dtype=torch.int32, device=dev)      # uint32 payload
        offsets = torch.empty((n + 1,), dtype=torch.int32, device=dev)
        nbytes = int(lib.esam3_rle_scratch_bytes(n, h, w, cap))
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.esam3_rle_encode(m.data_ptr(), n, h, w, counts.data_ptr(), cap, offsets.data_ptr(),
                                            scratch.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
                       "esam3_rle_encode")
        offs = offsets.cpu().numpy()
        total = int(offs[-1])
        if total <= cap:
            return counts[:total].cpu().numpy().view(np.uint32), offs
        cap = total  # the counts were truncated: retry with the exact size


def counts_to_string(counts: np.ndarray) -> str:
    """cocoapi's compressed string of one mask's run lengths (library host codec)."""
    lib = _lib.load()
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    buf = C.create_string_buffer(max(16, 7 * c.size))
    n = int(lib.esam3_rle_to_string(c.ctypes.data_as(C.c_void_p), c.size, buf, len(buf)))
    if n < 0:
        raise _lib.Esam3Error((lib.esam3_last_error() or b'rle string codec failed').decode())
    return buf.raw[:n].decode("ascii")


def string_to_counts(s: str) -> np.ndarray:
    lib = _lib.load()
    raw = s.encode("ascii")
    out = np.empty((max(1, len(raw)),), dtype=np.uint32)
    n = int(lib.esam3_rle_from_string(raw, len(raw), out.ctypes.data_as(C.c_void_p), out.size))
    if n < 0:
        raise _lib.Esam3Error((lib.esam3_last_error() or b'rle string codec failed').decode())
    return out[:n].copy()


def rle_encode(orig_mask: torch.Tensor, return_areas: bool = False) -> List[dict]:
    """Same contract as the reference's ``rle_encode`` (masks_ops.py:161-230): masks [N, H, W] bool ->
    list of {"size": [H, W], "counts": str[, "area": int]}."""
    assert orig_mask.ndim == 3, "Mask must be of shape (N, H, W)"
    assert orig_mask.dtype == torch.bool, "Mask must have dtype=torch.bool"
    if orig_mask.numel() == 0:
        return []
    counts, offs = rle_counts_device(orig_mask)
    h, w = int(orig_mask.shape[1]), int(orig_mask.shape[2])
    out = []
    for i in range(orig_mask.shape[0]):
        c = counts[offs[i]:offs[i + 1]]
        r = {"size": [h, w], "counts": counts_to_string(c)}
        if return_areas:
            r["area"] = int(c[1::2].sum(dtype=np.int64))
        out.append(r)
    return out


robust_rle_encode = rle_encode
