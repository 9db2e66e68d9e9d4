"""Shim of pycocotools.mask (absent third-party dependency) for the pieces the reference's hot-path neighbours
call: ``frPyObjects`` on an uncompressed RLE dict (sam3/train/masks_ops.py:223) and ``encode`` / ``decode`` of
Fortran-ordered uint8 masks.  The string codec is the oracle's restatement of cocoapi's published algorithm
(oracle/ref_rle.py); the last uncompressed RLE seen is kept in ``LAST_UNCOMPRESSED`` so that
oracle/gen_golden_rle.py can pin the oracle's run lengths against the reference's own ``rle_encode``."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import ref_rle  # noqa: E402

LAST_UNCOMPRESSED = []


def frPyObjects(obj, h, w):
    if isinstance(obj, dict) and isinstance(obj.get("counts"), list):
        LAST_UNCOMPRESSED.append({"counts": [int(c) for c in obj["counts"]], "size": [int(h), int(w)]})
        return {"size": [int(h), int(w)], "counts": ref_rle.counts_to_string(obj["counts"]).encode("ascii")}
    raise NotImplementedError("shim: only uncompressed RLE dicts")


def encode(mask):
    m = np.asarray(mask)
    if m.ndim == 3:
        return [encode(m[:, :, i]) for i in range(m.shape[2])]
    r = ref_rle.encode(m)
    r["counts"] = r["counts"].encode("ascii")
    return r


def decode(rle):
    if isinstance(rle, list):
        return np.stack([decode(r) for r in rle], axis=-1)
    s = rle["counts"].decode("ascii") if isinstance(rle["counts"], bytes) else rle["counts"]
    h, w = rle["size"]
    return np.asfortranarray(ref_rle.decode(ref_rle.string_to_counts(s), h, w))


def area(rle):
    s = rle["counts"].decode("ascii") if isinstance(rle["counts"], bytes) else rle["counts"]
    return int(sum(ref_rle.string_to_counts(s)[1::2]))
