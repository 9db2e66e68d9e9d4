"""Shim of skimage.measure.label for 2-D inputs: default connectivity == ndim
(8-connected in 2-D), backed by scipy.ndimage.label so the CPU oracle performs
hole filling instead of silently skipping it (sam1_utils.py:106-117)."""
import numpy as np
from scipy import ndimage


def label(values, return_num=False, connectivity=None, background=0):
    assert values.ndim == 2
    structure = np.ones((3, 3), dtype=np.int32) if connectivity in (None, 2) else None
    labels, num = ndimage.label(values != background, structure=structure)
    labels = labels.astype(np.int64)
    return (labels, num) if return_num else labels
