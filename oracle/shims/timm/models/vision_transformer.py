from timm.layers import trunc_normal_  # noqa: F401
