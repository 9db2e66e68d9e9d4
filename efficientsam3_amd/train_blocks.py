"""Forward AND backward of the layers an EfficientViT MBConv block is made of, on the HIP kernels (building
blocks of the stage-1 student-trunk backward; ``stage1_train.Stage1Trainer`` composes them, with ``train_repvit`` / ``train_tinyvit``, into the
training step of all nine students).  The reference's block is
``backbones/efficientvit/nn/ops.py:39-81`` (ConvLayer: Conv2d without bias -> BatchNorm2d -> activation) and ``:310-360`` (MBConv:
1x1 expand + Hardswish, depthwise 3x3 + Hardswish, 1x1 project, BatchNorm after each; ResidualBlock adds the input), run under
``model.train()`` by ``stage1/train_image_encoder_stage1.py:165``.  Everything is NHWC on the GPU; weight gradients come back fp32.

This module is a thin composition: it owns no arithmetic.  It sequences the gradient kernels (``esam3_act_backward``,
``esam3_bn_train_backward``, ``esam3_linear_wgrad``, ``esam3_dwconv_wgrad``, ...) in the order and with the tensors a real block hands
them; every block is checked against ``torch.autograd`` (tests/test_train_blocks.py).  Weights may be host tensors (the block tests:
packed on the CPU by the ``esam3_op_*`` test entry points) or DEVICE fp32 tensors -- views of the optimizer's flat arena
(``stage1.Stage1Updater``), the form ``stage1_train.Stage1Trainer`` uses: then the ``esam3_train_*`` entry points re-pack them on the
device every call (no host copy, no synchronisation) and a parameter update is seen by the next forward without any reload."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from . import stage1 as _s1
from .stage1 import bn_apply, bn_backward_apply, bn_backward_sums, bn_stats

ACT = {None: 0, "relu": 1, "gelu": 2, "hswish": 3, "sigmoid": 4}
_DT = {torch.float32: 0, torch.bfloat16: 1}
DEVICE = "cuda"   # where the block classes keep their BatchNorm parameters (the host-logic test runs the compositions on "cpu" doubles)


# SyncBatchNorm (stage1/train_image_encoder_stage1.py:62-63 `--use-sync-bn`: torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)): None = every
# rank normalises with its own batch statistics (nn.BatchNorm2d); a torch.distributed process group (or True for the default group) = the
# statistics of all its ranks, torch.nn.SyncBatchNorm's protocol with one collective each way.  Set by ``Stage1Trainer(sync_bn=True)``.
SYNC_BN = None          # process-wide default (tests, scripts); a trainer sets `sync` on ITS BatchNorm layers instead (round 6), so two
GLOBAL = object()       # trainers in one process -- with and without SyncBatchNorm, or on different groups -- do not change each other


def _sync(sync):
    """a layer's `sync` attribute (or the GLOBAL marker: the module default above) -> None | True | process group"""
    return SYNC_BN if sync is GLOBAL else sync


def _sync_group(sync=GLOBAL):
    v = _sync(sync)
    return None if v is True else v


def bn_train_forward(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: torch.Tensor = None, running_var: torch.Tensor = None,
                     momentum: float = 0.1, eps: float = 1e-5, sync=GLOBAL):
    """training-mode BatchNorm2d on NHWC rows -> (y, mean, rstd).  Under ``SYNC_BN``: every rank's (mean, biased variance, rows) are gathered
    (one all_gather of 2 C + 1 doubles), combined as torch's batch_norm_gather_stats_with_counts does -- the count-weighted mean, and the
    count-weighted mean of var + (mean_r - mean)^2 --, the running statistics take the all-rank values (unbiased variance with the all-rank
    count), and the map runs with them"""
    if _sync(sync) is None:
        return _s1.bn_train_forward(x, gamma, beta, running_mean, running_var, momentum, eps)
    import torch.distributed as dist
    c = x.shape[-1]
    mean_l, _, var_l = bn_stats(x, eps)
    rows = torch.full((1,), float(x.numel() // c), dtype=torch.float64, device=x.device)
    packed = torch.cat([mean_l.double(), var_l.double(), rows])
    gathered = [torch.empty_like(packed) for _ in range(dist.get_world_size(_sync_group(sync)))]
    dist.all_gather(gathered, packed, group=_sync_group(sync))
    g = torch.stack(gathered)
    cnt = g[:, -1:]
    n = cnt.sum()
    mean = (g[:, :c] * cnt).sum(0) / n
    var = ((g[:, c:2 * c] + (g[:, :c] - mean) ** 2) * cnt).sum(0) / n
    rstd = torch.rsqrt(var + eps)
    if running_mean is not None:
        running_mean.mul_(1.0 - momentum).add_((momentum * mean).float())
        running_var.mul_(1.0 - momentum).add_((momentum * var * n / (n - 1.0).clamp(min=1.0)).float())
    mean32, rstd32 = mean.float(), rstd.float()
    return bn_apply(x, gamma, beta, mean32, rstd32), mean32, rstd32


def bn_train_backward(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, save_mean: torch.Tensor, save_rstd: torch.Tensor, sync=GLOBAL):
    """its autograd backward -> (dx, dgamma, dbeta).  Under ``SYNC_BN``: the ranks' (sum dy xhat, sum dy, rows) are summed (one all_reduce),
    dx uses the all-rank means, dgamma / dbeta stay THIS rank's sums (the gradient all-reduce averages them like every other gradient), as
    torch.nn.SyncBatchNorm's backward does"""
    if _sync(sync) is None:
        return _s1.bn_train_backward(x, dy, gamma, save_mean, save_rstd)
    import torch.distributed as dist
    c = x.shape[-1]
    sdyx, sdy = bn_backward_sums(x, dy, save_mean, save_rstd)
    rows = torch.full((1,), float(x.numel() // c), dtype=torch.float64, device=x.device)
    packed = torch.cat([sdyx.double(), sdy.double(), rows])
    dist.all_reduce(packed, group=_sync_group(sync))
    means = (packed[:2 * c] / packed[-1]).float()
    dx = bn_backward_apply(x, dy, gamma, save_mean, save_rstd, means[:c].contiguous(), means[c:].contiguous())
    return dx, sdyx, sdy


# The ConvLayer's activation inside the BatchNorm kernels' own passes (esam3_bn_act_train_*): one elementwise pass less each way.  Off under
# SyncBatchNorm (its halves have no fused form) and in the CPU composition tests (they stand in for the unfused kernels).
FUSE_BN_ACT = True


def bn_act_forward(x: torch.Tensor, gamma, beta, running_mean, running_var, momentum: float, eps: float, act, sync=GLOBAL):
    """BatchNorm2d (training mode) then ``act`` -> (pre = the BatchNorm's output, act(pre), mean, rstd); pre is None in the fused form: its
    backward recomputes it from x (one tensor write here, two tensor reads there, and one kept tensor per layer less)"""
    if act is not None and FUSE_BN_ACT and _sync(sync) is None:
        return _s1.bn_act_train_forward(x, gamma, beta, running_mean, running_var, momentum, eps, act, keep_pre=False)
    pre, mean, rstd = bn_train_forward(x, gamma, beta, running_mean, running_var, momentum, eps, sync=sync)
    return pre, (act_forward(pre, act) if act else pre), mean, rstd


def bn_act_backward(x: torch.Tensor, dy: torch.Tensor, pre: torch.Tensor, act, gamma, mean, rstd, sync=GLOBAL, beta=None):
    """its backward: dy = the gradient of act(pre) -> (dx, dgamma, dbeta); ``pre`` None (the fused forward's): recomputed, needs ``beta``"""
    if act is not None and FUSE_BN_ACT and _sync(sync) is None:
        return _s1.bn_act_train_backward(x, dy, pre, act, gamma, mean, rstd, beta=beta)
    return bn_train_backward(x, act_backward(pre, dy, act) if act else dy, gamma, mean, rstd, sync=sync)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _host(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.detach().float().cpu().numpy())


def act_forward(x: torch.Tensor, act) -> torch.Tensor:
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().esam3_act_forward(_DT[x.dtype], x.data_ptr(), y.data_ptr(), x.numel(), ACT[act], _stream()), "esam3_act_forward")
    return y


def act_backward(x: torch.Tensor, dy: torch.Tensor, act) -> torch.Tensor:
    """dx = dy * act'(x), x = the activation's INPUT"""
    dx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().esam3_act_backward(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), ACT[act], _stream()),
                   "esam3_act_backward")
    return dx


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _dev_f32(t: torch.Tensor) -> torch.Tensor:
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "device-resident weights are contiguous fp32 (arena views)"
    return t


def linear_forward(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor = None) -> torch.Tensor:
    """x [..., K] @ w[N, K]^T (+ bias): a 1x1 conv on NHWC rows"""
    k, n = x.shape[-1], w.shape[0]
    m = x.numel() // k
    out = torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    if w.is_cuda:   # device-resident master weight: packed on the device, no host round trip
        ws = _ws(lib.esam3_train_pack_bytes(_DT[x.dtype], n, k), x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.esam3_train_linear(_DT[x.dtype], x.data_ptr(), _dev_f32(w).data_ptr(), None if bias is None else _dev_f32(bias).data_ptr(),
                                              out.data_ptr(), m, n, k, 0, ws.data_ptr(), _stream()), "esam3_train_linear")
        return out
    wh = _host(w)
    bh = None if bias is None else _host(bias)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_op_linear(_DT[x.dtype], x.data_ptr(), wh.ctypes.data, None if bh is None else bh.ctypes.data, None,
                                       out.data_ptr(), m, n, k, 0, _stream()), "esam3_op_linear")
    return out


def linear_dgrad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """dx [..., K] = dy [..., N] @ w[N, K]: the forward operator with the transposed weight"""
    if w.is_cuda:
        n, k = w.shape
        m = dy.numel() // n
        dx = torch.empty(dy.shape[:-1] + (k,), dtype=dy.dtype, device=dy.device)
        lib = _lib.load()
        ws = _ws(lib.esam3_train_pack_bytes(_DT[dy.dtype], k, n), dy.device)
        with torch.cuda.device(dy.device):
            _lib.check(lib.esam3_train_linear(_DT[dy.dtype], dy.data_ptr(), _dev_f32(w).data_ptr(), None, dx.data_ptr(), m, k, n, 1, ws.data_ptr(),
                                              _stream()), "esam3_train_linear")
        return dx
    return linear_forward(dy, w.t().contiguous())


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dw [N, K] fp32 = sum over the rows of dy[row, :]^T x[row, :]"""
    n, k = dy.shape[-1], x.shape[-1]
    m = x.numel() // k
    lib = _lib.load()
    dw = torch.empty((n, k), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.esam3_linear_wgrad_workspace(m, n, k)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_linear_wgrad(_DT[x.dtype], dy.data_ptr(), x.data_ptr(), m, n, k, dw.data_ptr(), None, ws.data_ptr(), _stream()),
                   "esam3_linear_wgrad")
    return dw


def dwconv_forward(x: torch.Tensor, w: torch.Tensor, stride: int = 1, bias: torch.Tensor = None) -> torch.Tensor:
    """depthwise k x k (3 | 5), padding k / 2, on x [B, H, W, C]; w [C, 1, k, k]"""
    b, h, wd, c = x.shape
    out = torch.empty((b, (h + stride - 1) // stride, (wd + stride - 1) // stride, c), dtype=x.dtype, device=x.device)
    ks = int(w.shape[-1])
    if w.is_cuda:
        ws = _ws(4 * ks * ks * c, x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().esam3_train_dwconv(_DT[x.dtype], x.data_ptr(), _dev_f32(w).data_ptr(), None if bias is None else _dev_f32(bias).data_ptr(),
                                                      out.data_ptr(), b, h, wd, c, ks, stride, ws.data_ptr(), _stream()), "esam3_train_dwconv")
        return out
    wh = _host(w)
    bh = None if bias is None else _host(bias)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().esam3_op_dwconv(_DT[x.dtype], x.data_ptr(), wh.ctypes.data, None if bh is None else bh.ctypes.data, out.data_ptr(), b, h,
                                               wd, c, ks, stride, 0, _stream()), "esam3_op_dwconv")
    return out


def colsum(dy: torch.Tensor) -> torch.Tensor:
    """sum over all rows of dy [..., N] -> [N] fp32: the gradient of a conv bias"""
    n = dy.shape[-1]
    m = dy.numel() // n
    lib = _lib.load()
    out = torch.empty(n, dtype=torch.float32, device=dy.device)
    ws = torch.empty(int(lib.esam3_colsum_workspace(m, n)), dtype=torch.uint8, device=dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(lib.esam3_colsum(_DT[dy.dtype], dy.data_ptr(), m, n, out.data_ptr(), ws.data_ptr(), _stream()), "esam3_colsum")
    return out


def channel_scale(x: torch.Tensor, mul: torch.Tensor, bias: torch.Tensor = None, add: torch.Tensor = None, plus_one: bool = False,
                  bias_scale: float = 1.0) -> torch.Tensor:
    """out = add + x * (mul [+ 1]) + bias * bias_scale on x [B, ..., C]; mul / bias device fp32, [C] (per channel) or [B, C] (per image and
    channel); the elementwise half of RepVGGDW and SqueezeExcite, forwards and backwards (``esam3_channel_scale``)"""
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    assert x.is_contiguous() and (add is None or (add.shape == x.shape and add.dtype == x.dtype and add.is_contiguous()))
    for t in (mul, bias):
        assert t is None or (tuple(t.shape) in ((c,), (b, c)) and _dev_f32(t) is t)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().esam3_channel_scale(_DT[x.dtype], x.data_ptr(), mul.data_ptr(), int(mul.dim() == 2), 1.0 if plus_one else 0.0,
                                                   None if bias is None else bias.data_ptr(), int(bias is not None and bias.dim() == 2),
                                                   float(bias_scale), None if add is None else add.data_ptr(), out.data_ptr(), b, hw, c, _stream()),
                   "esam3_channel_scale")
    return out


_ONES = {}


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a + b (same shape and dtype; fp32 sum, rounded once): the residual additions of the blocks, as ``esam3_channel_scale`` with a unit
    multiplier -- one launch where the torch form (two casts, an add, a cast) was four"""
    c = a.shape[-1]
    key = (c, str(a.device))
    if key not in _ONES:
        _ONES[key] = torch.ones(c, dtype=torch.float32, device=a.device)
    return channel_scale(a.contiguous(), _ONES[key], add=b.contiguous())


def batched_coldot(a: torch.Tensor, b2: torch.Tensor = None, scale: float = 1.0, per_image: bool = True) -> torch.Tensor:
    """[B, C] fp32 (``per_image``) or [C]: scale * the sum over the pixels (of each image | of all images) of a * b2 (of a when ``b2`` is
    None) for a [B, ..., C] (``esam3_batched_coldot``)"""
    c = a.shape[-1]
    b = a.shape[0] if per_image else 1
    hw = a.numel() // (b * c)
    assert a.is_contiguous() and (b2 is None or (b2.shape == a.shape and b2.dtype == a.dtype and b2.is_contiguous()))
    lib = _lib.load()
    out = torch.empty((b, c) if per_image else (c,), dtype=torch.float32, device=a.device)
    ws = torch.empty(int(lib.esam3_batched_coldot_workspace(b, c)), dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.esam3_batched_coldot(_DT[a.dtype], a.data_ptr(), None if b2 is None else b2.data_ptr(), b, hw, c, float(scale), out.data_ptr(),
                                            ws.data_ptr(), _stream()), "esam3_batched_coldot")
    return out


def dwconv_dgrad(dy: torch.Tensor, w: torch.Tensor, in_hw, stride: int = 1) -> torch.Tensor:
    """dx [B, H, W, C] of the depthwise conv from dy [B, ceil(H/s), ceil(W/s), C] (for stride 2 a transposed convolution)"""
    b, c = dy.shape[0], dy.shape[-1]
    h, wd = in_hw
    dx = torch.empty((b, h, wd, c), dtype=dy.dtype, device=dy.device)
    wdev = _dev_f32(w) if w.is_cuda else w.detach().float().to(dy.device).contiguous()
    ks = int(w.shape[-1])
    ws = _ws(4 * ks * ks * c, dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(_lib.load().esam3_train_dwconv_dgrad(_DT[dy.dtype], dy.data_ptr(), wdev.data_ptr(), dx.data_ptr(), b, h, wd, c, ks, stride,
                                                        ws.data_ptr(), _stream()), "esam3_train_dwconv_dgrad")
    return dx


def dwconv_wgrad(x: torch.Tensor, dy: torch.Tensor, stride: int = 1, ksize: int = 3) -> torch.Tensor:
    b, h, wd, c = x.shape
    lib = _lib.load()
    dw = torch.empty((c, 1, ksize, ksize), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.esam3_dwconv_wgrad_workspace(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_dwconv_wgrad(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), b, h, wd, c, ksize, stride, dw.data_ptr(), ws.data_ptr(),
                                          _stream()), "esam3_dwconv_wgrad")
    return dw


class ConvLayerTrain:
    """ConvLayer (ops.py:39-81) in training mode: conv (1x1 ``kind="pw"`` or depthwise k x k ``kind="dw"``) -> BatchNorm2d when ``gamma`` /
    ``beta`` are given, else a conv bias -> activation, with the tensors the backward needs kept on the object."""

    def __init__(self, kind: str, weight: torch.Tensor, gamma: torch.Tensor = None, beta: torch.Tensor = None, act=None, eps: float = 1e-5,
                 momentum: float = 0.1, stride: int = 1, bias: torch.Tensor = None, running_mean: torch.Tensor = None,
                 running_var: torch.Tensor = None):
        assert kind in ("pw", "dw") and (stride == 1 or kind == "dw") and (gamma is None) == (beta is None)
        self.kind, self.w, self.act, self.eps, self.momentum, self.stride, self.bias = kind, weight, act, eps, momentum, stride, bias
        self.norm = gamma is not None
        if self.norm:
            # device fp32 tensors are kept AS GIVEN (views of the optimizer's arena stay views: an update is seen at once)
            self.gamma, self.beta = gamma.float().to(DEVICE).contiguous(), beta.float().to(DEVICE).contiguous()
            c = gamma.numel()
            # BatchNorm buffers: the state dict's norm.running_mean / running_var when given (updated in place, exported by the trainer)
            self.running_mean = torch.zeros(c, dtype=torch.float32, device=DEVICE) if running_mean is None else running_mean.float().to(DEVICE).contiguous()
            self.running_var = torch.ones(c, dtype=torch.float32, device=DEVICE) if running_var is None else running_var.float().to(DEVICE).contiguous()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.x = x
        self.conv_out = linear_forward(x, self.w, self.bias) if self.kind == "pw" else dwconv_forward(x, self.w, self.stride, self.bias)
        if self.norm:
            self.pre, y, self.mean, self.rstd = bn_act_forward(self.conv_out, self.gamma, self.beta, self.running_mean, self.running_var,
                                                               self.momentum, self.eps, self.act, sync=getattr(self, "sync", GLOBAL))
            return y
        self.pre = self.conv_out
        return act_forward(self.pre, self.act) if self.act else self.pre

    def backward(self, dy: torch.Tensor):
        """-> (dx, {"weight": dw, "gamma" / "beta" or "bias": ...})"""
        grads = {}
        if self.norm:
            d_conv, grads["gamma"], grads["beta"] = bn_act_backward(self.conv_out, dy, self.pre, self.act, self.gamma, self.mean, self.rstd, sync=getattr(self, "sync", GLOBAL), beta=self.beta)
        else:
            d_conv = act_backward(self.pre, dy, self.act) if self.act else dy
        if self.bias is not None:
            grads["bias"] = colsum(d_conv)
        if self.kind == "pw":
            grads["weight"] = linear_wgrad(d_conv, self.x)
            dx = linear_dgrad(d_conv, self.w)
        else:
            grads["weight"] = dwconv_wgrad(self.x, d_conv, self.stride, int(self.w.shape[-1]))
            dx = dwconv_dgrad(d_conv, self.w, self.x.shape[1:3], self.stride)
        return dx, grads


class MBConvTrain:
    """ResidualBlock(MBConv(Cin -> Cmid -> Cout, stride 1), Identity) of the EfficientViT trunks (ops.py:310-360, backbone.py:91-147):
    inverted 1x1 + Hardswish, depthwise 3x3 (stride 1, or 2 in the first block of a stage) + Hardswish, pointwise 1x1, a BatchNorm after
    each; ``y = x + block(x)`` when ``residual``."""

    def __init__(self, params: dict, residual: bool = True, act="hswish", stride: int = 1):
        """``params``: "<layer>.weight" plus either "<layer>.gamma" / ".beta" (a BatchNorm follows) or "<layer>.bias" (the fewer_norm form of
        stages 3-4 and of the EfficientViTBlock's local module: use_bias=(True, True, False), norm=(None, None, bn2d), ops.py:704-711)."""
        assert not (residual and stride != 1)
        kw = lambda n: dict(gamma=params.get(f"{n}.gamma"), beta=params.get(f"{n}.beta"), bias=params.get(f"{n}.bias"),  # noqa: E731
                            running_mean=params.get(f"{n}.running_mean"), running_var=params.get(f"{n}.running_var"))
        self.inv = ConvLayerTrain("pw", params["inverted.weight"], act=act, **kw("inverted"))
        self.dw = ConvLayerTrain("dw", params["depth.weight"], act=act, stride=stride, **kw("depth"))
        self.pw = ConvLayerTrain("pw", params["point.weight"], act=None, **kw("point"))
        self.residual = residual

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.pw.forward(self.dw.forward(self.inv.forward(x)))
        return add(x, y) if self.residual else y

    def backward(self, dy: torch.Tensor):
        d, g_pw = self.pw.backward(dy)
        d, g_dw = self.dw.backward(d)
        d, g_inv = self.inv.backward(d)
        dx = add(d, dy) if self.residual else d
        grads = {f"{n}.{k}": v for n, g in (("inverted", g_inv), ("depth", g_dw), ("point", g_pw)) for k, v in g.items()}
        return dx, grads


class DSConvTrain:
    """ResidualBlock(DSConv(C -> C), Identity) of the EfficientViT input stem (ops.py:264-307, backbone.py:60-78): depthwise 3x3 +
    BatchNorm + Hardswish, pointwise 1x1 + BatchNorm; ``y = x + block(x)``."""

    def __init__(self, params: dict, residual: bool = True, act="hswish"):
        rs = lambda n: dict(running_mean=params.get(f"{n}.running_mean"), running_var=params.get(f"{n}.running_var"))  # noqa: E731
        self.dw = ConvLayerTrain("dw", params["depth.weight"], params["depth.gamma"], params["depth.beta"], act, **rs("depth"))
        self.pw = ConvLayerTrain("pw", params["point.weight"], params["point.gamma"], params["point.beta"], None, **rs("point"))
        self.residual = residual

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.pw.forward(self.dw.forward(x))
        return add(x, y) if self.residual else y

    def backward(self, dy: torch.Tensor):
        d, g_pw = self.pw.backward(dy)
        d, g_dw = self.dw.backward(d)
        dx = add(d, dy) if self.residual else d
        return dx, {f"{n}.{k}": v for n, g in (("depth", g_dw), ("point", g_pw)) for k, v in g.items()}


def lite_mla_backward(ms: torch.Tensor, dout: torch.Tensor, groups: int, dim: int, eps: float = 1e-15):
    """backward of LiteMLA.relu_linear_att (ops.py:584-621) on the multi-scale qkv tensor ms [B, N, groups * 3 * dim] (per head group the
    channels [q | k | v]) given dout [B, N, groups * dim]: returns (d_ms, y) -- the gradient and, for free, the forward output."""
    b, n, c3 = ms.shape
    assert c3 == groups * 3 * dim and tuple(dout.shape) == (b, n, groups * dim) and ms.is_contiguous() and dout.is_contiguous()
    dms, y = torch.empty_like(ms), torch.empty_like(dout)
    lib = _lib.load()
    ws = _ws(lib.esam3_lite_mla_backward_workspace(b, n, groups, dim), ms.device)   # tokens split over workgroups, fixed-order sums (round 5)
    with torch.cuda.device(ms.device):
        _lib.check(lib.esam3_lite_mla_backward_ws(_DT[ms.dtype], ms.data_ptr(), dout.data_ptr(), dms.data_ptr(), y.data_ptr(), b, n, groups,
                                                  dim, float(eps), ws.data_ptr(), _stream()), "esam3_lite_mla_backward_ws")
    return dms, y


def lite_mla_forward(ms: torch.Tensor, groups: int, dim: int, eps: float = 1e-15) -> torch.Tensor:
    """LiteMLA.relu_linear_att (ops.py:584-621) forward only: y [B, N, groups * dim] (the backward kernels with dout = NULL: the S pass and
    one pass over the tokens; round 5 -- the training forward used to run the whole backward with a zero upstream gradient)"""
    b, n, c3 = ms.shape
    assert c3 == groups * 3 * dim and ms.is_contiguous()
    y = torch.empty((b, n, groups * dim), dtype=ms.dtype, device=ms.device)
    lib = _lib.load()
    ws = _ws(lib.esam3_lite_mla_backward_workspace(b, n, groups, dim), ms.device)
    with torch.cuda.device(ms.device):
        _lib.check(lib.esam3_lite_mla_backward_ws(_DT[ms.dtype], ms.data_ptr(), None, None, y.data_ptr(), b, n, groups, dim, float(eps),
                                                  ws.data_ptr(), _stream()), "esam3_lite_mla_backward_ws (forward only)")
    return y


def lite_mla_pair(ms0: torch.Tensor, ms1: torch.Tensor, dout, groups: int, dim: int, eps: float = 1e-15):
    """the two functions above on the multi-scale tensor AS ITS TWO HALVES (round 6: ``esam3_lite_mla_backward_ws2``): ms0 = the qkv conv's
    output, ms1 = the aggregated scale, each [B, N, groups / 2 * 3 * dim].  ``dout`` None: the forward output y; else (d_ms0, d_ms1).  Saves the
    torch.cat in front of the forward and the two slice copies behind the backward."""
    b, n, c3h = ms0.shape
    assert ms1.shape == ms0.shape and 2 * c3h == groups * 3 * dim and ms0.is_contiguous() and ms1.is_contiguous() and n > 256
    lib = _lib.load()
    ws = _ws(lib.esam3_lite_mla_backward_workspace(b, n, groups, dim), ms0.device)
    if dout is None:
        y = torch.empty((b, n, groups * dim), dtype=ms0.dtype, device=ms0.device)
        with torch.cuda.device(ms0.device):
            _lib.check(lib.esam3_lite_mla_backward_ws2(_DT[ms0.dtype], ms0.data_ptr(), ms1.data_ptr(), None, None, None, y.data_ptr(), b, n, groups, dim,
                                                       float(eps), ws.data_ptr(), _stream()), "esam3_lite_mla_backward_ws2 (forward only)")
        return y
    assert tuple(dout.shape) == (b, n, groups * dim) and dout.is_contiguous()
    d0, d1, y = torch.empty_like(ms0), torch.empty_like(ms1), torch.empty_like(dout)
    with torch.cuda.device(ms0.device):
        _lib.check(lib.esam3_lite_mla_backward_ws2(_DT[ms0.dtype], ms0.data_ptr(), ms1.data_ptr(), dout.data_ptr(), d0.data_ptr(), d1.data_ptr(),
                                                   y.data_ptr(), b, n, groups, dim, float(eps), ws.data_ptr(), _stream()), "esam3_lite_mla_backward_ws2")
    return d0, d1


_DIAG_IDX = {}


def _diag_idx(g: int, device) -> torch.Tensor:
    key = (g, str(device))
    if key not in _DIAG_IDX:
        _DIAG_IDX[key] = torch.arange(g, device=device)
    return _DIAG_IDX[key]


def _blockdiag(wg: torch.Tensor, gs: int) -> torch.Tensor:
    """grouped 1x1 weight [C, gs, 1, 1] (group size gs) -> the dense block-diagonal [C, C] matrix the engine's GEMM runs; pure data
    movement, on the weight's own device (a device-resident weight changes every step: rebuilt per forward)"""
    c = wg.shape[0]
    g = c // gs
    dense = torch.zeros((g, gs, g, gs), dtype=torch.float32, device=wg.device)
    idx = _diag_idx(g, wg.device)
    dense[idx, :, idx, :] = wg.detach().float().reshape(g, gs, gs)
    return dense.reshape(c, c)


def _blockdiag_extract(dense: torch.Tensor, gs: int) -> torch.Tensor:
    """the diagonal gs x gs blocks of a dense [C, C] matrix -> [C, gs, 1, 1] (the grouped weight's gradient inside dy^T x)"""
    c = dense.shape[0]
    g = c // gs
    idx = _diag_idx(g, dense.device)
    return dense.reshape(g, gs, g, gs)[idx, :, idx, :].reshape(c, gs, 1, 1).contiguous()


class LiteMLATrain:
    """ResidualBlock(LiteMLA(C -> C, dim, scales=(5,)), Identity): the context module of an EfficientViTBlock (ops.py:521-640, 643-690) in
    training mode.  qkv 1x1 (no norm) -> [qkv | grouped-1x1(dw5x5(qkv))] -> ReLU linear attention -> proj 1x1 + BatchNorm -> + x."""

    def __init__(self, params: dict, dim: int, eps: float = 1e-15):
        self.p, self.dim, self.eps = params, dim, eps
        self.c = params["qkv.weight"].shape[1]
        self.proj = ConvLayerTrain("pw", params["proj.weight"], params["proj.gamma"], params["proj.beta"], None,
                                   running_mean=params.get("proj.running_mean"), running_var=params.get("proj.running_var"))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, h, w, c = x.shape
        self.x = x
        self.qkv = linear_forward(x, self.p["qkv.weight"])                       # [B, H, W, 3C]
        self.agg1 = dwconv_forward(self.qkv, self.p["aggreg.dw.weight"], 1)     # depthwise 5x5
        self.wg_dense = _blockdiag(self.p["aggreg.pw.weight"], self.dim)         # from the CURRENT grouped weight
        agg2 = linear_forward(self.agg1, self.wg_dense)                          # grouped 1x1 as a block-diagonal GEMM
        self.groups = 2 * (c // self.dim)
        self.pair = x.is_cuda and h * w > 256       # the kernels take the two scales as separate tensors (no concatenated copy)
        if self.pair:
            self.agg2 = agg2
            att = lite_mla_pair(self.qkv.reshape(b, h * w, 3 * c), agg2.reshape(b, h * w, 3 * c), None, self.groups, self.dim, self.eps)
        else:
            self.ms = torch.cat([self.qkv, agg2], dim=-1).reshape(b, h * w, 6 * c).contiguous()
            att = lite_mla_forward(self.ms, self.groups, self.dim, self.eps)
        self.att = att.reshape(b, h, w, 2 * c)
        y = self.proj.forward(self.att)
        return add(x, y)

    def backward(self, dy: torch.Tensor):
        b, h, w, c = self.x.shape
        d_att, g_proj = self.proj.backward(dy)
        if self.pair:
            d0, d1 = lite_mla_pair(self.qkv.reshape(b, h * w, 3 * c), self.agg2.reshape(b, h * w, 3 * c),
                                   d_att.reshape(b, h * w, 2 * c).contiguous(), self.groups, self.dim, self.eps)
            d_qkv_direct, d_agg2 = d0.reshape(b, h, w, 3 * c), d1.reshape(b, h, w, 3 * c)
        else:
            d_ms, _ = lite_mla_backward(self.ms, d_att.reshape(b, h * w, 2 * c).contiguous(), self.groups, self.dim, self.eps)
            d_ms = d_ms.reshape(b, h, w, 6 * c)
            d_qkv_direct, d_agg2 = d_ms[..., :3 * c].contiguous(), d_ms[..., 3 * c:].contiguous()
        dense = linear_wgrad(d_agg2, self.agg1)                                  # [3C, 3C]; only its diagonal blocks are the grouped weight's
        dwg = _blockdiag_extract(dense, self.dim)
        d_agg1 = linear_dgrad(d_agg2, self.wg_dense)
        dwd = dwconv_wgrad(self.qkv, d_agg1, 1, 5)
        d_qkv = add(d_qkv_direct, dwconv_dgrad(d_agg1, self.p["aggreg.dw.weight"], (h, w), 1))
        dwq = linear_wgrad(d_qkv, self.x)
        dx = add(linear_dgrad(d_qkv, self.p["qkv.weight"]), dy)
        grads = {"qkv.weight": dwq, "aggreg.dw.weight": dwd, "aggreg.pw.weight": dwg, "proj.weight": g_proj["weight"],
                 "proj.gamma": g_proj["gamma"], "proj.beta": g_proj["beta"]}
        return dx, grads


class EfficientViTBlockTrain:
    """EfficientViTBlock (ops.py:670-730): context module ResidualBlock(LiteMLA) followed by local module ResidualBlock(MBConv with conv
    biases on its first two layers and one BatchNorm at the end) -- the repeated block of stages 3 and 4 of the EfficientViT trunks."""

    def __init__(self, context_params: dict, local_params: dict, dim: int):
        self.context = LiteMLATrain(context_params, dim)
        self.local = MBConvTrain(local_params, residual=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.local.forward(self.context.forward(x))

    def backward(self, dy: torch.Tensor):
        d, g_local = self.local.backward(dy)
        dx, g_ctx = self.context.backward(d)
        grads = {f"context.{k}": v for k, v in g_ctx.items()}
        grads.update({f"local.{k}": v for k, v in g_local.items()})
        return dx, grads


# ---- the trunk: every layer of EfficientViTBackbone (backbones/efficientvit/backbone.py:33-137) in training mode --------------------------
def stem_forward(img_nchw_f32: torch.Tensor, w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """the input stem's dense 3x3 stride-2 conv on the fp32 NCHW image -> NHWC [B, ceil(H/2), ceil(W/2), Cout] in ``dtype`` (no bias, no
    activation: the BatchNorm and Hardswish of the ConvLayer follow as separate steps in training mode)"""
    b, _, h, wd = img_nchw_f32.shape
    cout = w.shape[0]
    out = torch.empty((b, (h + 1) // 2, (wd + 1) // 2, cout), dtype=dtype, device=img_nchw_f32.device)
    if w.is_cuda:
        ws = _ws(108 * cout, img_nchw_f32.device)
        with torch.cuda.device(img_nchw_f32.device):
            _lib.check(_lib.load().esam3_train_stem(_DT[dtype], img_nchw_f32.data_ptr(), _dev_f32(w).data_ptr(), out.data_ptr(), b, h, wd, cout,
                                                    ws.data_ptr(), _stream()), "esam3_train_stem")
        return out
    wh = _host(w)
    with torch.cuda.device(img_nchw_f32.device):
        _lib.check(_lib.load().esam3_op_stem(_DT[dtype], img_nchw_f32.data_ptr(), wh.ctypes.data, None, out.data_ptr(), b, h, wd, cout, 0, _stream()),
                   "esam3_op_stem")
    return out


def stem_im2col(img_nchw_f32: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[B * OH * OW, 32] rows of the 27 image values under each output pixel of the stride-2 3x3 conv (order ci, kh, kw = the weight's), padded to
    32 columns: the `x` operand of the stem's weight gradient.  Pure data movement (torch's unfold)."""
    if img_nchw_f32.is_cuda and dtype in _DT:   # round 6: one kernel (esam3_stem_im2col) instead of unfold + permute + pad + cast
        b, _, h, wd = img_nchw_f32.shape
        img = img_nchw_f32.contiguous()
        out = torch.empty((b * ((h + 1) // 2) * ((wd + 1) // 2), 32), dtype=dtype, device=img.device)
        with torch.cuda.device(img.device):
            _lib.check(_lib.load().esam3_stem_im2col(_DT[dtype], img.data_ptr(), out.data_ptr(), b, h, wd, _stream()), "esam3_stem_im2col")
        return out
    cols = torch.nn.functional.unfold(img_nchw_f32, kernel_size=3, padding=1, stride=2)       # [B, 27, OH * OW]  (host doubles of the tests)
    cols = cols.permute(0, 2, 1).reshape(-1, 27)
    return torch.nn.functional.pad(cols, (0, 5)).to(dtype).contiguous()


class StemConvTrain:
    """input_stem.op_list.0: ConvLayer(3 -> C0, 3x3, stride 2) + BatchNorm + Hardswish on the image (backbone.py:50-58).  No input gradient."""

    def __init__(self, weight: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, dtype: torch.dtype, act="hswish", eps=1e-5, momentum=0.1,
                 running_mean: torch.Tensor = None, running_var: torch.Tensor = None):
        self.w, self.act, self.dtype, self.eps, self.momentum = weight, act, dtype, eps, momentum
        self.gamma, self.beta = gamma.float().to(DEVICE).contiguous(), beta.float().to(DEVICE).contiguous()
        c = gamma.numel()
        self.running_mean = torch.zeros(c, dtype=torch.float32, device=DEVICE) if running_mean is None else running_mean.float().to(DEVICE).contiguous()
        self.running_var = torch.ones(c, dtype=torch.float32, device=DEVICE) if running_var is None else running_var.float().to(DEVICE).contiguous()

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        self.img = img
        self.conv_out = stem_forward(img, self.w, self.dtype)
        self.pre, y, self.mean, self.rstd = bn_act_forward(self.conv_out, self.gamma, self.beta, self.running_mean, self.running_var, self.momentum,
                                                           self.eps, self.act, sync=getattr(self, "sync", GLOBAL))
        return y

    def backward(self, dy: torch.Tensor):
        d_conv, dgamma, dbeta = bn_act_backward(self.conv_out, dy, self.pre, self.act, self.gamma, self.mean, self.rstd, sync=getattr(self, "sync", GLOBAL), beta=self.beta)
        dw = linear_wgrad(d_conv, stem_im2col(self.img, self.dtype))[:, :27].reshape(self.w.shape)
        return None, {"weight": dw, "gamma": dgamma, "beta": dbeta}


class EfficientViTTrunkTrain:
    """EfficientViTBackbone (backbone.py:33-137) in training mode from a state dict in the reference's names: the stem conv, depth_list[0] DSConv
    blocks, two stages of MBConv blocks (the first of each with stride 2), two stages of a stride-2 MBConv in its conv-bias form followed by
    EfficientViTBlocks.  ``forward(image)`` returns the last stage's output (NHWC); ``backward(dy)`` returns the gradient of every parameter under
    its state-dict name (``...conv.weight``, ``...conv.bias``, ``...norm.weight``, ``...norm.bias``)."""

    def __init__(self, sd: dict, width_list, depth_list, dim: int, dtype: torch.dtype = torch.float32, prefix: str = ""):
        g = lambda k: sd[prefix + k]  # noqa: E731
        self.layers = []   # (state-dict prefix, block object, {block grad key: state-dict suffix})

        def conv_keys(base, name, kind):   # parameters of one ConvLayer -> block params + the map back to state-dict names
            p, back = {}, {}
            w = g(f"{base}.{name}.conv.weight")
            p[f"{kind}.weight"] = w.reshape(w.shape[0], w.shape[1]) if w.shape[-1] == 1 and kind != "depth" else w
            back[f"{kind}.weight"] = f"{base}.{name}.conv.weight"
            if prefix + f"{base}.{name}.conv.bias" in sd:
                p[f"{kind}.bias"] = g(f"{base}.{name}.conv.bias"); back[f"{kind}.bias"] = f"{base}.{name}.conv.bias"
            if prefix + f"{base}.{name}.norm.weight" in sd:
                p[f"{kind}.gamma"] = g(f"{base}.{name}.norm.weight"); back[f"{kind}.gamma"] = f"{base}.{name}.norm.weight"
                p[f"{kind}.beta"] = g(f"{base}.{name}.norm.bias"); back[f"{kind}.beta"] = f"{base}.{name}.norm.bias"
                for stat in ("running_mean", "running_var"):   # BatchNorm buffers: loaded when the state dict has them
                    if prefix + f"{base}.{name}.norm.{stat}" in sd:
                        p[f"{kind}.{stat}"] = g(f"{base}.{name}.norm.{stat}")
            return p, back

        def mbconv(base, residual, stride):
            p, back = {}, {}
            for name, kind in (("inverted_conv", "inverted"), ("depth_conv", "depth"), ("point_conv", "point")):
                a, b_ = conv_keys(base, name, kind)
                p.update(a); back.update(b_)
            return MBConvTrain(p, residual=residual, stride=stride), back

        self.shapes = {k[len(prefix):]: tuple(v.shape) for k, v in sd.items() if k.startswith(prefix)}
        self.stem = StemConvTrain(g("input_stem.op_list.0.conv.weight"), g("input_stem.op_list.0.norm.weight"), g("input_stem.op_list.0.norm.bias"), dtype,
                                  running_mean=sd.get(prefix + "input_stem.op_list.0.norm.running_mean"),
                                  running_var=sd.get(prefix + "input_stem.op_list.0.norm.running_var"))
        for i in range(1, depth_list[0] + 1):
            base = f"input_stem.op_list.{i}.main"
            p, back = {}, {}
            for name, kind in (("depth_conv", "depth"), ("point_conv", "point")):
                a, b_ = conv_keys(base, name, kind)
                p.update(a); back.update(b_)
            self.layers.append((DSConvTrain(p), back))
        for s_, d in enumerate(depth_list[1:3]):
            for i in range(d):
                self.layers.append(mbconv(f"stages.{s_}.op_list.{i}.main", residual=i > 0, stride=2 if i == 0 else 1))
        for s_, d in enumerate(depth_list[3:], start=2):
            self.layers.append(mbconv(f"stages.{s_}.op_list.0.main", residual=False, stride=2))
            for i in range(1, d + 1):
                cb, lb = f"stages.{s_}.op_list.{i}.context_module.main", f"stages.{s_}.op_list.{i}.local_module.main"
                pc = {"qkv.weight": g(f"{cb}.qkv.conv.weight").flatten(1), "aggreg.dw.weight": g(f"{cb}.aggreg.0.0.weight"),
                      "aggreg.pw.weight": g(f"{cb}.aggreg.0.1.weight"), "proj.weight": g(f"{cb}.proj.conv.weight").flatten(1),
                      "proj.gamma": g(f"{cb}.proj.norm.weight"), "proj.beta": g(f"{cb}.proj.norm.bias")}
                for stat in ("running_mean", "running_var"):
                    if prefix + f"{cb}.proj.norm.{stat}" in sd:
                        pc[f"proj.{stat}"] = g(f"{cb}.proj.norm.{stat}")
                back = {"context.qkv.weight": f"{cb}.qkv.conv.weight", "context.aggreg.dw.weight": f"{cb}.aggreg.0.0.weight",
                        "context.aggreg.pw.weight": f"{cb}.aggreg.0.1.weight", "context.proj.weight": f"{cb}.proj.conv.weight",
                        "context.proj.gamma": f"{cb}.proj.norm.weight", "context.proj.beta": f"{cb}.proj.norm.bias"}
                pl = {}
                for name, kind in (("inverted_conv", "inverted"), ("depth_conv", "depth"), ("point_conv", "point")):
                    a, b_ = conv_keys(lb, name, kind)
                    pl.update(a); back.update({f"local.{k}": v for k, v in b_.items()})
                self.layers.append((EfficientViTBlockTrain(pc, pl, dim), back))

    def forward(self, img_nchw_f32: torch.Tensor) -> torch.Tensor:
        x = self.stem.forward(img_nchw_f32)
        for blk, _ in self.layers:
            x = blk.forward(x)
        return x

    def backward(self, dy: torch.Tensor, sink=None) -> dict:
        """gradients of every parameter under its state-dict name AND in its state-dict shape (a 1x1 conv weight comes back
        [N, K, 1, 1]); ``sink(name, grad)`` is called as soon as a gradient exists (last layer first: the order a bucketed all-reduce
        wants, ``dist.GradientAllReducer.push``)"""
        grads = {}

        def put(name, gval):
            grads[name] = gval.reshape(self.shapes[name])
            if sink is not None:
                sink(name, grads[name])

        d = dy
        for blk, back in reversed(self.layers):
            d, g_ = blk.backward(d)
            for k, name in back.items():
                put(name, g_[k])
        _, g_ = self.stem.backward(d)
        put("input_stem.op_list.0.conv.weight", g_["weight"])
        put("input_stem.op_list.0.norm.weight", g_["gamma"])
        put("input_stem.op_list.0.norm.bias", g_["beta"])
        return grads

    def norm_layers(self):
        """(state-dict prefix of the BatchNorm, layer object holding running_mean / running_var) of every BatchNorm of the trunk"""
        out = [("input_stem.op_list.0.norm", self.stem)]
        for blk, back in self.layers:
            subs = []
            if isinstance(blk, EfficientViTBlockTrain):
                subs = [("context.proj", blk.context.proj), ("local.inverted", blk.local.inv), ("local.depth", blk.local.dw), ("local.point", blk.local.pw)]
            elif isinstance(blk, MBConvTrain):
                subs = [("inverted", blk.inv), ("depth", blk.dw), ("point", blk.pw)]
            elif isinstance(blk, DSConvTrain):
                subs = [("depth", blk.dw), ("point", blk.pw)]
            for key, layer in subs:
                if getattr(layer, "norm", False) and f"{key}.gamma" in back:
                    out.append((back[f"{key}.gamma"][:-len(".weight")], layer))
        return out
