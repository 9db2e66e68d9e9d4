"""ORACLE (test infrastructure only): CPU restatement of the mask -> COCO RLE step of the reference's evaluation
writers.  Two parts:

* ``rle_counts``: the uncompressed run lengths, following ``rle_encode`` of the reference
  (sam3/sam3/train/masks_ops.py:161-230: transpose to Fortran order, mark value changes, difference the change
  positions) — pinned against that function run here (oracle/gen_golden_rle.py, tests/golden/rle/).
* ``counts_to_string`` / ``string_to_counts``: the compressed "counts" string of pycocotools.  pycocotools
  (cocoapi, unpinned in the reference's pyproject) is a third-party dependency that is absent from
  /root/reference and from this image: this restates its published algorithm (cocoapi common/maskApi.c
  rleToString / rleFrString).  **Parity unpinned** for the string form; the tests check the round trip and the
  uncompressed counts.

Only tests may import this module."""
from __future__ import annotations

from typing import List

import numpy as np


def rle_counts(mask: np.ndarray) -> List[int]:
    """mask [H, W] (non-zero = foreground) -> run lengths in column-major order, starting with the zeros run."""
    flat = (np.asarray(mask) != 0).T.reshape(-1)          # Fortran order (masks_ops.py:181-186)
    n = flat.size
    diff = np.ones(n + 1, dtype=bool)                      # masks_ops.py:190-194
    diff[1:-1] = flat[:-1] != flat[1:]
    diff[0] = flat[0]
    idx = np.nonzero(diff)[0]
    counts = idx.copy()
    counts[1:] -= idx[:-1]                                 # masks_ops.py:203-208
    return [int(c) for c in counts]


def counts_to_string(counts: List[int]) -> str:
    """cocoapi rleToString: delta against the count two places back (from the 4th on), 5 bits per character."""
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5                                        # arithmetic shift, like C on a signed long
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def string_to_counts(s: str) -> List[int]:
    """cocoapi rleFrString."""
    counts: List[int] = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def decode(counts: List[int], h: int, w: int) -> np.ndarray:
    """Run lengths -> mask [H, W] uint8 (cocoapi rleDecode)."""
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, v = 0, 0
    for c in counts:
        flat[pos:pos + c] = v
        pos += c
        v ^= 1
    assert pos == h * w, (pos, h, w)
    return flat.reshape(w, h).T.copy()


def encode(mask: np.ndarray) -> dict:
    """pycocotools.mask.encode equivalent for one [H, W] mask: {"size": [H, W], "counts": str}."""
    h, w = mask.shape
    return {"size": [int(h), int(w)], "counts": counts_to_string(rle_counts(mask))}
