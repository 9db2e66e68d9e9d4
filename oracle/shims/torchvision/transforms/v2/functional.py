import numpy as np
import torch


def to_image(x):
    """PIL / HWC ndarray -> CHW tensor; tensors pass through."""
    if isinstance(x, torch.Tensor):
        return x
    arr = np.asarray(x)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()
