"""Forward AND backward of the RepViT student trunks (RV-S / RV-M / RV-L = repvit_m0_9 / m1_1 / m2_3) in training mode, on the HIP
kernels.  The reference runs them under ``model.train()`` in ``stage1/train_image_encoder_stage1.py:165-226``
through ``stage1/model.py:287-296`` (``RepViTAdapter``: every layer of ``model.features``) built by ``stage1/model.py:386-395``; the
layers are ``sam3/backbones/repvit.py``:

* ``:27-36``   ``Conv2d_BN``: Conv2d without bias -> BatchNorm2d                         -> ``train_blocks.ConvLayerTrain`` (1x1, depthwise)
  and ``Conv3x3S2Train`` below (the dense stride-2 3x3 of the patch embedding);
* ``:84-93``   ``RepVGGDW``: ``bn(conv_bn_dw3x3(x) + conv1(x) + x)``, conv1 a depthwise 1x1 WITH bias   -> ``RepVGGDWTrain``;
* timm ``SqueezeExcite(C, 0.25)`` (``:136,150``): ``x * sigmoid(fc2(relu(fc1(mean_hw(x)))))``, 1x1 convs with bias -> ``SqueezeExciteTrain``;
* ``:125-161`` ``RepViTBlock``: token mixer (RepVGGDW [+ SE] | depthwise 3x3 stride 2 [+ SE] + 1x1) then
  ``Residual(1x1 C -> 2C, GELU, 1x1 2C -> C)`` (``:51-63``; drop 0)                                       -> ``RepViTBlockTrain``;
* ``:226-231`` the patch embedding: ``Conv2d_BN(3, C/2, 3, 2, 1)``, GELU, ``Conv2d_BN(C/2, C, 3, 2, 1)``        -> ``StemConvTrain`` + ``Conv3x3S2Train``.

Like ``train_blocks`` this module owns no arithmetic: it sequences kernels (``esam3_train_*``,
``esam3_bn_train_*``, ``esam3_channel_scale``, ``esam3_batched_coldot``, the gradient kernels) on NHWC tensors; parameters are fp32 device views
of the optimizer's arena.  The tiny SqueezeExcite MLP ([B, C] rows) runs in fp32 whatever the activation dtype.  The composition is checked
against torch.autograd on the CPU with kernel stand-ins (tests/test_train_repvit_host.py), the kernels and blocks against autograd on the
GPU (tests/test_train_blocks.py), a whole training step against the reference's own run (tests/test_stage1_step.py)."""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import _lib
from . import train_blocks as tb
from .schema import REPVIT_CFG

_DT = tb._DT


# ---- the dense 3x3 of the patch embedding -------------------------------------------------------------------------------------------------
def _conv3x3(x: torch.Tensor, w: torch.Tensor, out_channels: int, dgrad: bool) -> torch.Tensor:
    b, h, wd, cin = x.shape
    out = torch.empty((b, h, wd, out_channels), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    ws = tb._ws(lib.esam3_train_pack_bytes(_DT[x.dtype], out_channels, 9 * cin), x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_train_conv3x3(_DT[x.dtype], x.data_ptr(), tb._dev_f32(w).data_ptr(), None, out.data_ptr(), b, h, wd, cin, out_channels,
                                           1 if dgrad else 0, ws.data_ptr(), tb._stream()), "esam3_train_conv3x3")
    return out


def conv3x3_s2_forward(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """dense 3x3, stride 2, padding 1, NHWC; w [Cout, Cin, 3, 3] device fp32 -> [B, ceil(H/2), ceil(W/2), Cout]"""
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    out = torch.empty((b, (h + 1) // 2, (wd + 1) // 2, cout), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    ws = tb._ws(lib.esam3_train_pack_bytes(_DT[x.dtype], cout, 9 * cin), x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_train_conv3x3_s2(_DT[x.dtype], x.data_ptr(), tb._dev_f32(w).data_ptr(), None, out.data_ptr(), b, h, wd, cin, cout,
                                              ws.data_ptr(), tb._stream()), "esam3_train_conv3x3_s2")
    return out


def conv3x3_s2_dgrad(dy: torch.Tensor, w: torch.Tensor, in_hw) -> torch.Tensor:
    """dx [B, H, W, Cin] of that conv: dy sits on the even pixels of a zero H x W grid (data movement), then the stride-1 data-gradient conv
    (the 3x3 conv with the rotated, channel-transposed weight, packed that way on the device)"""
    b, oh, ow, cout = dy.shape
    h, wd = in_hw
    up = torch.zeros((b, h, wd, cout), dtype=dy.dtype, device=dy.device)
    up[:, ::2, ::2] = dy
    return _conv3x3(up, w, int(w.shape[1]), dgrad=True)


def conv3x3_s2_wgrad(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dw [Cout, Cin, 3, 3] fp32: tap (ky, kx) is the 1x1 weight gradient of dy against the input pixels (2 oy + ky - 1, 2 ox + kx - 1) (zero
    outside the image): ``esam3_conv3x3_wgrad`` with stride 2, one launch (the operand gathered in the kernel)"""
    b, h, wd, cin = x.shape
    cout = dy.shape[-1]
    assert x.is_contiguous() and dy.is_contiguous() and dy.dtype == x.dtype
    lib = _lib.load()
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    ws = tb._ws(lib.esam3_conv3x3_wgrad_workspace(b, h, wd, cin, cout, 2), x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_conv3x3_wgrad(_DT[x.dtype], dy.data_ptr(), x.data_ptr(), b, h, wd, cin, cout, 2, dw.data_ptr(), ws.data_ptr(), tb._stream()),
                   "esam3_conv3x3_wgrad")
    return dw


class Conv3x3S2Train:
    """``Conv2d_BN(Cin, Cout, 3, 2, 1)`` (repvit.py:27-36,230): dense 3x3 stride 2 -> BatchNorm2d, no activation."""

    def __init__(self, weight: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, momentum: float = 0.1,
                 running_mean: torch.Tensor = None, running_var: torch.Tensor = None):
        self.w, self.eps, self.momentum = weight, eps, momentum
        self.gamma, self.beta = gamma.float().to(tb.DEVICE).contiguous(), beta.float().to(tb.DEVICE).contiguous()
        c = gamma.numel()
        self.running_mean = torch.zeros(c, dtype=torch.float32, device=tb.DEVICE) if running_mean is None else running_mean.float().to(tb.DEVICE).contiguous()
        self.running_var = torch.ones(c, dtype=torch.float32, device=tb.DEVICE) if running_var is None else running_var.float().to(tb.DEVICE).contiguous()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.x = x
        self.conv_out = conv3x3_s2_forward(x, self.w)
        y, self.mean, self.rstd = tb.bn_train_forward(self.conv_out, self.gamma, self.beta, self.running_mean, self.running_var, self.momentum, self.eps, sync=getattr(self, "sync", tb.GLOBAL))
        return y

    def backward(self, dy: torch.Tensor):
        d_conv, dgamma, dbeta = tb.bn_train_backward(self.conv_out, dy, self.gamma, self.mean, self.rstd, sync=getattr(self, "sync", tb.GLOBAL))
        dw = conv3x3_s2_wgrad(d_conv, self.x)
        dx = conv3x3_s2_dgrad(d_conv, self.w, self.x.shape[1:3])
        return dx, {"weight": dw, "gamma": dgamma, "beta": dbeta}


# ---- RepVGGDW -----------------------------------------------------------------------------------------------------------------------------------
class RepVGGDWTrain:
    """``bn(conv(x) + conv1(x) + x)`` (repvit.py:84-93): ``conv`` = depthwise 3x3 + BatchNorm, ``conv1`` = depthwise 1x1 with bias (a per-channel
    scale and shift), then a BatchNorm over the sum.  ``params``: conv.weight / .gamma / .beta (+ .running_*), conv1.weight [C, 1, 1, 1] /
    conv1.bias, bn.gamma / .beta (+ .running_*)."""

    def __init__(self, params: dict, eps: float = 1e-5, momentum: float = 0.1):
        p = params
        self.conv = tb.ConvLayerTrain("dw", p["conv.weight"], p["conv.gamma"], p["conv.beta"], None, eps=eps, momentum=momentum,
                                      running_mean=p.get("conv.running_mean"), running_var=p.get("conv.running_var"))
        self.w1, self.b1, self.eps, self.momentum = p["conv1.weight"], p["conv1.bias"], eps, momentum
        self.gamma, self.beta = p["bn.gamma"].float().to(tb.DEVICE).contiguous(), p["bn.beta"].float().to(tb.DEVICE).contiguous()
        c = self.gamma.numel()
        rm, rv = p.get("bn.running_mean"), p.get("bn.running_var")
        self.running_mean = torch.zeros(c, dtype=torch.float32, device=tb.DEVICE) if rm is None else rm.float().to(tb.DEVICE).contiguous()
        self.running_var = torch.ones(c, dtype=torch.float32, device=tb.DEVICE) if rv is None else rv.float().to(tb.DEVICE).contiguous()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.x = x
        a = self.conv.forward(x)
        self.s = tb.channel_scale(x, self.w1.reshape(-1), bias=self.b1, add=a, plus_one=True)        # conv(x) + (w1 x + b1) + x
        y, self.mean, self.rstd = tb.bn_train_forward(self.s, self.gamma, self.beta, self.running_mean, self.running_var, self.momentum, self.eps, sync=getattr(self, "sync", tb.GLOBAL))
        return y

    def backward(self, dy: torch.Tensor):
        ds, dgamma, dbeta = tb.bn_train_backward(self.s, dy, self.gamma, self.mean, self.rstd, sync=getattr(self, "sync", tb.GLOBAL))
        dx_conv, g = self.conv.backward(ds)
        dw1 = tb.batched_coldot(ds, self.x, per_image=False)                                           # sum over all pixels of ds x
        db1 = tb.colsum(ds)
        dx = tb.channel_scale(ds, self.w1.reshape(-1), add=dx_conv, plus_one=True)                     # dx_conv + ds (w1 + 1)
        return dx, {"conv.weight": g["weight"], "conv.gamma": g["gamma"], "conv.beta": g["beta"], "conv1.weight": dw1.reshape(self.w1.shape),
                    "conv1.bias": db1, "bn.gamma": dgamma, "bn.beta": dbeta}


# ---- SqueezeExcite ------------------------------------------------------------------------------------------------------------------------------
class SqueezeExciteTrain:
    """timm ``SqueezeExcite(C, rd_ratio=0.25)`` as RepViT uses it: ``x * sigmoid(fc2(relu(fc1(x.mean((2, 3))))))``, fc1 / fc2 1x1 convs with
    bias (``params``: fc1.weight [R, C, 1, 1], fc1.bias, fc2.weight [C, R, 1, 1], fc2.bias).  The mean, the two tiny GEMMs and the gate are
    fp32 [B, C] / [B, R] rows."""

    def __init__(self, params: dict):
        self.w1, self.b1, self.w2, self.b2 = params["fc1.weight"], params["fc1.bias"], params["fc2.weight"], params["fc2.bias"]
        self.r, self.c = int(self.w1.shape[0]), int(self.w1.shape[1])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.x = x
        self.hw = x.shape[1] * x.shape[2]
        self.pooled = tb.batched_coldot(x, None, scale=1.0 / self.hw)                                   # [B, C] fp32
        self.z1 = tb.linear_forward(self.pooled, self.w1.reshape(self.r, self.c), self.b1)
        self.h = tb.act_forward(self.z1, "relu")
        self.z2 = tb.linear_forward(self.h, self.w2.reshape(self.c, self.r), self.b2)
        self.gate = tb.act_forward(self.z2, "sigmoid")
        return tb.channel_scale(x, self.gate)

    def backward(self, dy: torch.Tensor):
        d_gate = tb.batched_coldot(dy, self.x)                                                          # [B, C]: sum over the pixels of dy x
        d_z2 = tb.act_backward(self.z2, d_gate, "sigmoid")
        grads = {"fc2.weight": tb.linear_wgrad(d_z2, self.h).reshape(self.w2.shape), "fc2.bias": tb.colsum(d_z2)}
        d_z1 = tb.act_backward(self.z1, tb.linear_dgrad(d_z2, self.w2.reshape(self.c, self.r)), "relu")
        grads["fc1.weight"] = tb.linear_wgrad(d_z1, self.pooled).reshape(self.w1.shape)
        grads["fc1.bias"] = tb.colsum(d_z1)
        d_pooled = tb.linear_dgrad(d_z1, self.w1.reshape(self.r, self.c))
        dx = tb.channel_scale(dy, self.gate, bias=d_pooled, bias_scale=1.0 / self.hw)                  # dy gate + the mean's share
        return dx, grads


# ---- RepViTBlock ---------------------------------------------------------------------------------------------------------------------------------
def _conv_bn(get: Callable[[str], torch.Tensor], has: Callable[[str], bool], base: str, kind: str, act=None, stride: int = 1):
    """``Conv2d_BN`` under ``base`` (``<base>.c.weight``, ``<base>.bn.*``) -> (ConvLayerTrain, {block grad key: state-dict name})"""
    w = get(base + ".c.weight")
    layer = tb.ConvLayerTrain(kind, w.reshape(w.shape[0], w.shape[1]) if kind == "pw" else w, get(base + ".bn.weight"), get(base + ".bn.bias"), act,
                              stride=stride, running_mean=get(base + ".bn.running_mean") if has(base + ".bn.running_mean") else None,
                              running_var=get(base + ".bn.running_var") if has(base + ".bn.running_var") else None)
    return layer, {"weight": base + ".c.weight", "gamma": base + ".bn.weight", "beta": base + ".bn.bias"}


class RepViTBlockTrain:
    """``RepViTBlock`` (repvit.py:125-161) under the state-dict prefix ``base`` (``features.<i>``).  ``backward`` returns the gradients under
    their state-dict names (relative to the trunk)."""

    def __init__(self, get, has, base: str, stride: int, use_se: bool):
        self.stride, self.base = stride, base
        opt = lambda k: get(k) if has(k) else None  # noqa: E731
        self.parts = []      # (layer object, {grad key: state-dict name}) in forward order
        tm = base + ".token_mixer"
        if stride == 2:
            self.parts.append(_conv_bn(get, has, tm + ".0", "dw", stride=2))
            if use_se:
                self.parts.append(self._se(get, tm + ".1"))
            self.parts.append(_conv_bn(get, has, tm + ".2", "pw"))
        else:
            q = tm + ".0"
            p = {"conv.weight": get(q + ".conv.c.weight"), "conv.gamma": get(q + ".conv.bn.weight"), "conv.beta": get(q + ".conv.bn.bias"),
                 "conv.running_mean": opt(q + ".conv.bn.running_mean"), "conv.running_var": opt(q + ".conv.bn.running_var"),
                 "conv1.weight": get(q + ".conv1.weight"), "conv1.bias": get(q + ".conv1.bias"), "bn.gamma": get(q + ".bn.weight"),
                 "bn.beta": get(q + ".bn.bias"), "bn.running_mean": opt(q + ".bn.running_mean"), "bn.running_var": opt(q + ".bn.running_var")}
            back = {"conv.weight": q + ".conv.c.weight", "conv.gamma": q + ".conv.bn.weight", "conv.beta": q + ".conv.bn.bias",
                    "conv1.weight": q + ".conv1.weight", "conv1.bias": q + ".conv1.bias", "bn.gamma": q + ".bn.weight", "bn.beta": q + ".bn.bias"}
            self.parts.append((RepVGGDWTrain({k: v for k, v in p.items() if v is not None}), back))
            if use_se:
                self.parts.append(self._se(get, tm + ".1"))
        cm = base + ".channel_mixer.m"
        self.mixer = [_conv_bn(get, has, cm + ".0", "pw", act="gelu"), _conv_bn(get, has, cm + ".2", "pw")]

    @staticmethod
    def _se(get, q: str):
        names = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")
        return SqueezeExciteTrain({n: get(f"{q}.{n}") for n in names}), {n: f"{q}.{n}" for n in names}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for layer, _ in self.parts:
            x = layer.forward(x)
        h = self.mixer[1][0].forward(self.mixer[0][0].forward(x))
        return tb.add(x, h)                                              # Residual (repvit.py:57-63, drop 0)

    def backward(self, dy: torch.Tensor):
        grads = {}

        def through(layer, back, d):
            d, g = layer.backward(d)
            grads.update({back[k]: v for k, v in g.items()})
            return d

        d = through(*self.mixer[1], dy)
        d = through(*self.mixer[0], d)
        d = tb.add(d, dy)
        for layer, back in reversed(self.parts):
            d = through(layer, back, d)
        return d, grads

    def norm_layers(self):
        """(state-dict prefix of the BatchNorm, object holding running_mean / running_var / momentum)"""
        out = []
        for layer, back in self.parts + self.mixer:
            if isinstance(layer, RepVGGDWTrain):
                out.append((back["conv.gamma"][:-len(".weight")], layer.conv))
                out.append((back["bn.gamma"][:-len(".weight")], layer))
            elif isinstance(layer, tb.ConvLayerTrain):
                out.append((back["gamma"][:-len(".weight")], layer))
        return out


# ---- the trunk ----------------------------------------------------------------------------------------------------------------------------------
class RepViTTrunkTrain:
    """``RepViTAdapter`` (stage1/model.py:287-296) = every layer of ``RepViT.features`` (repvit.py:226-238) in training mode, from a state dict
    in the reference's names (``features.0.0.c.weight`` ...).  Same interface as ``train_blocks.EfficientViTTrunkTrain``: ``forward(image)``
    -> the last block's output (NHWC), ``backward(dy, sink)`` -> the gradient of every parameter under its state-dict name and shape,
    handed to ``sink`` the moment it exists (last layer first), ``norm_layers()``."""

    def __init__(self, sd: Dict[str, torch.Tensor], model_name: str, dtype: torch.dtype = torch.float32, prefix: str = ""):
        cfgs = REPVIT_CFG[model_name]
        get = lambda k: sd[prefix + k]  # noqa: E731
        has = lambda k: (prefix + k) in sd  # noqa: E731
        opt = lambda k: sd.get(prefix + k)  # noqa: E731
        self.shapes = {k[len(prefix):]: tuple(v.shape) for k, v in sd.items() if k.startswith(prefix)}
        self.stem1 = tb.StemConvTrain(get("features.0.0.c.weight"), get("features.0.0.bn.weight"), get("features.0.0.bn.bias"), dtype, act="gelu",
                                      running_mean=opt("features.0.0.bn.running_mean"), running_var=opt("features.0.0.bn.running_var"))
        self.stem2 = Conv3x3S2Train(get("features.0.2.c.weight"), get("features.0.2.bn.weight"), get("features.0.2.bn.bias"),
                                    running_mean=opt("features.0.2.bn.running_mean"), running_var=opt("features.0.2.bn.running_var"))
        self.blocks = [RepViTBlockTrain(get, has, f"features.{i}", stride, bool(use_se))
                       for i, (_k, _t, _c, use_se, _hs, stride) in enumerate(cfgs, start=1)]

    def forward(self, img_nchw_f32: torch.Tensor) -> torch.Tensor:
        x = self.stem2.forward(self.stem1.forward(img_nchw_f32))
        for blk in self.blocks:
            x = blk.forward(x)
        return x

    def backward(self, dy: torch.Tensor, sink=None) -> dict:
        grads = {}

        def put(name, gval):
            grads[name] = gval.reshape(self.shapes[name])
            if sink is not None:
                sink(name, grads[name])

        d = dy
        for blk in reversed(self.blocks):
            d, g = blk.backward(d)
            for name, gval in g.items():
                put(name, gval)
        d, g = self.stem2.backward(d)
        for key, suffix in (("weight", "c.weight"), ("gamma", "bn.weight"), ("beta", "bn.bias")):
            put("features.0.2." + suffix, g[key])
        _, g = self.stem1.backward(d)
        for key, suffix in (("weight", "c.weight"), ("gamma", "bn.weight"), ("beta", "bn.bias")):
            put("features.0.0." + suffix, g[key])
        return grads

    def norm_layers(self):
        out = [("features.0.0.bn", self.stem1), ("features.0.2.bn", self.stem2)]
        for blk in self.blocks:
            out.extend(blk.norm_layers())
        return out
