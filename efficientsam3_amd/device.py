"""Device helpers with the reference's names (sam3/sam3/device.py:6-35)."""
import torch


def get_device() -> torch.device:
    if torch.cuda.is_available():  # ROCm exposes HIP devices through the cuda namespace
        return torch.device("cuda")
    raise RuntimeError("no HIP device visible: EfficientSAM3-AMD has no CPU path")


def get_autocast_device_type(device=None) -> str:
    return "cuda"


def get_autocast_dtype(device=None) -> torch.dtype:
    return torch.bfloat16
