from timm.layers import *  # noqa: F401,F403
from timm.layers import DropPath, Mlp, trunc_normal_, to_2tuple, SqueezeExcite  # noqa: F401
