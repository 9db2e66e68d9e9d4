"""Forward AND backward of the TinyViT student trunks (TV-S / TV-M / TV-L = tiny_vit_5m / 11m / 21m) in training mode, on the HIP kernels
(SURVEY.md 8(f).3, round 5).  The reference runs them under ``model.train()`` in ``stage1/train_image_encoder_stage1.py:165-226`` through
``stage1/model.py:299-324`` (``TinyViTAdapter``: patch_embed, every layer, tokens back to a map; head and norm_head removed) built by
``stage1/model.py:397-406``; the layers are ``sam3/backbones/tiny_vit.py``:

* ``:67-84``   ``PatchEmbed``: Conv2d_BN(3, C/2, 3, 2, 1), GELU, Conv2d_BN(C/2, C, 3, 2, 1)          -> ``StemConvTrain`` + ``train_repvit.Conv3x3S2Train``
* ``:87-125``  ``MBConv``: 1x1 + BN, GELU, depthwise 3x3 + BN, GELU, 1x1 + BN, drop_path, + shortcut, GELU   -> ``MBConvTrain``
* ``:128-154`` ``PatchMerging``: 1x1 + BN, GELU, depthwise 3x3 stride 2 + BN, GELU, 1x1 + BN                     -> ``PatchMergingTrain``
* ``:196-216`` ``Mlp``: LayerNorm, fc1, GELU, fc2;  ``:219-293`` ``Attention``: LayerNorm, qkv Linear, per-head
  softmax(q k^T scale + attention_biases[:, idxs]) v, proj Linear                                          -> ``WindowAttentionTrain``
* ``:296-386`` ``TinyViTBlock``: zero-pad to whole windows, window partition, attention, reverse, crop, residual (drop_path), depthwise
  3x3 + BN (local_conv), residual MLP (drop_path)                                                            -> ``TinyViTBlockTrain``

Activations stay NHWC [B, H, W, C] between blocks (the reference's [B, L, C] token view of the same memory).  Padding tokens are real rows
here as in the reference: they are zeros BEFORE the attention's LayerNorm, so they enter the attention as LayerNorm(0) = its bias -- keys and
values every real token attends to, and a path the gradients of norm.bias and qkv.* take; only their input gradient is dropped by the crop.

Stochastic depth (``DropPath``, timm: a per-sample factor 0 or 1 / keep on the residual branch, rates linspace(0, 0.0 | 0.1 | 0.2, blocks) for
5m | 11m | 21m, tiny_vit.py:491,663-689) is drawn on the host by ``drop_path_sampler(name, call, batch)`` -- by default from a seeded torch
generator -- and applied by ``esam3_channel_scale``; tests inject the factors the reference's own run drew.

Like ``train_blocks`` / ``train_repvit`` this module owns no arithmetic: kernels (``esam3_ln_train_*``, ``esam3_win_attn_train_*``,
``esam3_attn_bias_gather_sum``, ``esam3_channel_scale``, the convolution / BatchNorm / activation / gradient kernels) and data movement
(padding, window partition: torch views and copies)."""
from __future__ import annotations

import itertools
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import _lib
from . import train_blocks as tb
from .schema import TINYVIT_CFG
from .train_repvit import Conv3x3S2Train

_DT = tb._DT
DROP_PATH_RATE = {"5m": 0.0, "11m": 0.1, "21m": 0.2}     # tiny_vit.py:663,676,689
HEAD_DIM = 32                                             # embed_dims / num_heads of every TinyViT stage (tiny_vit.py:657-692)


# ---- kernel wrappers ----------------------------------------------------------------------------------------------------------------------------
def layernorm_forward(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """nn.LayerNorm over the last dimension of x [..., C] -> (y, mean [rows], rstd [rows]) (``esam3_ln_train_forward``)"""
    c = x.shape[-1]
    m = x.numel() // c
    y = torch.empty_like(x)
    mean = torch.empty(m, dtype=torch.float32, device=x.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().esam3_ln_train_forward(_DT[x.dtype], x.data_ptr(), y.data_ptr(), m, c, tb._dev_f32(gamma).data_ptr(),
                                                      tb._dev_f32(beta).data_ptr(), float(eps), mean.data_ptr(), rstd.data_ptr(), tb._stream()),
                   "esam3_ln_train_forward")
    return y, mean, rstd


def layernorm_backward(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor):
    """-> (dx, dgamma [C] fp32, dbeta [C] fp32)"""
    c = x.shape[-1]
    m = x.numel() // c
    assert dy.shape == x.shape and dy.dtype == x.dtype and dy.is_contiguous() and x.is_contiguous()
    lib = _lib.load()
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = tb._ws(lib.esam3_ln_train_workspace(c), x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_ln_train_backward(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), tb._dev_f32(gamma).data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), m, c, ws.data_ptr(), tb._stream()),
                   "esam3_ln_train_backward")
    return dx, dgamma, dbeta


def _tab_ok(tab, ws, n, heads) -> bool:
    return tab is not None and ws * ws == n and ws <= 16 and tab.is_cuda and tab.dtype == torch.float32 and tab.is_contiguous() and tuple(tab.shape) == (heads, n)


def win_attn_forward(qkv: torch.Tensor, bias: torch.Tensor, heads: int, scale: float, tab: torch.Tensor = None, ws: int = 0):
    """qkv [windows, N, heads * 96] (per head q | k | v), bias [heads, N, N] fp32 -> (out [windows, N, heads * 32], lse [windows, heads, N]).
    ``tab`` [heads, ws ws] = the attention_biases parameter ``bias`` was gathered from (N = ws ws): the bf16 kernels index it from LDS"""
    nw, n, _ = qkv.shape
    out = torch.empty((nw, n, heads * HEAD_DIM), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((nw, heads, n), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        if _tab_ok(tab, ws, n, heads):
            _lib.check(_lib.load().esam3_win_attn_train_forward_tab(_DT[qkv.dtype], qkv.data_ptr(), bias.data_ptr(), tab.data_ptr(), ws, out.data_ptr(),
                                                                    lse.data_ptr(), nw, heads, float(scale), tb._stream()), "esam3_win_attn_train_forward_tab")
            return out, lse
        _lib.check(_lib.load().esam3_win_attn_train_forward(_DT[qkv.dtype], qkv.data_ptr(), bias.data_ptr(), out.data_ptr(), lse.data_ptr(), nw, n, heads,
                                                            float(scale), tb._stream()), "esam3_win_attn_train_forward")
    return out, lse


DS_CHUNK_BYTES = 256 << 20   # bound on the transient logits-gradient tensor of win_attn_backward


def win_attn_backward(qkv: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, lse: torch.Tensor, dout: torch.Tensor, heads: int, scale: float,
                      tab: torch.Tensor = None, ws: int = 0):
    """-> (dqkv like qkv, dbias [heads, N, N] fp32 = the logits' gradient summed over the windows)"""
    nw, n, _ = qkv.shape
    assert dout.shape == out.shape and dout.dtype == qkv.dtype and dout.is_contiguous()
    dqkv = torch.empty_like(qkv)
    # The logits' gradient dS [windows, heads N N] fp32 exists only to be summed over the windows (the bias gradient).  For TinyViT-11M at
    # 1008^2 and batch 32 the whole tensor is 0.98 GB per stage-2 block (1.5 GB for 21M): the windows are walked in chunks of at most
    # DS_CHUNK_BYTES and the chunk sums added in chunk order (a fixed order: repeats stay bit-identical).  One chunk = the old path.
    per_window = heads * n * n * 4
    chunk = max(1, min(nw, DS_CHUNK_BYTES // per_window))
    ds = torch.empty((chunk, heads * n * n), dtype=torch.float32, device=qkv.device)
    dbias = None
    lib = _lib.load()
    with torch.cuda.device(qkv.device):
        for a in range(0, nw, chunk):
            e = min(nw, a + chunk)
            if _tab_ok(tab, ws, n, heads):
                _lib.check(lib.esam3_win_attn_train_backward_tab(_DT[qkv.dtype], qkv[a:e].data_ptr(), bias.data_ptr(), tab.data_ptr(), ws, out[a:e].data_ptr(),
                                                                 lse[a:e].data_ptr(), dout[a:e].data_ptr(), dqkv[a:e].data_ptr(), ds.data_ptr(), e - a, heads,
                                                                 float(scale), tb._stream()), "esam3_win_attn_train_backward_tab")
            else:
                _lib.check(lib.esam3_win_attn_train_backward(_DT[qkv.dtype], qkv[a:e].data_ptr(), bias.data_ptr(), out[a:e].data_ptr(), lse[a:e].data_ptr(),
                                                             dout[a:e].data_ptr(), dqkv[a:e].data_ptr(), ds.data_ptr(), e - a, n, heads, float(scale),
                                                             tb._stream()), "esam3_win_attn_train_backward")
            part = tb.colsum(ds[:e - a])
            dbias = part if dbias is None else dbias.add_(part)
    return dqkv, dbias.reshape(heads, n, n)


def attention_bias_idxs(ws: int) -> np.ndarray:
    """``Attention.attention_bias_idxs`` (tiny_vit.py:240-254): [N, N] index of the offset (|dy|, |dx|) of every token pair, offsets numbered in
    order of first appearance"""
    points = list(itertools.product(range(ws), range(ws)))
    offsets, idxs = {}, []
    for p1 in points:
        for p2 in points:
            off = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
            if off not in offsets:
                offsets[off] = len(offsets)
            idxs.append(offsets[off])
    return np.asarray(idxs, dtype=np.int64).reshape(len(points), len(points))


_BIAS_TABLES: Dict[tuple, tuple] = {}


def _bias_tables(ws: int, device):
    """(idxs LongTensor [N, N], CSR start int32 [n_off + 1], CSR items int32 [N N]) on ``device``"""
    key = (ws, str(device))
    if key not in _BIAS_TABLES:
        idxs = attention_bias_idxs(ws)
        flat = idxs.reshape(-1)
        order = np.argsort(flat, kind="stable")
        counts = np.bincount(flat, minlength=ws * ws)
        start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        _BIAS_TABLES[key] = (torch.from_numpy(idxs).to(device), torch.from_numpy(start).to(device), torch.from_numpy(order.astype(np.int32)).to(device))
    return _BIAS_TABLES[key]


def attn_bias_gather(biases: torch.Tensor, ws: int) -> torch.Tensor:
    """``attention_biases[:, attention_bias_idxs]`` -> [heads, N, N] fp32 (a gather: data movement)"""
    idxs, _, _ = _bias_tables(ws, biases.device)
    return biases[:, idxs].contiguous()


def attn_bias_grad(dbias_full: torch.Tensor, ws: int) -> torch.Tensor:
    """the adjoint of that gather: [heads, N, N] -> [heads, ws * ws], every offset's pairs summed in a fixed order"""
    heads, n = dbias_full.shape[0], dbias_full.shape[1]
    _, start, items = _bias_tables(ws, dbias_full.device)
    out = torch.empty((heads, ws * ws), dtype=torch.float32, device=dbias_full.device)
    with torch.cuda.device(dbias_full.device):
        _lib.check(_lib.load().esam3_attn_bias_gather_sum(dbias_full.data_ptr(), start.data_ptr(), items.data_ptr(), out.data_ptr(), heads, n * n, ws * ws,
                                                          tb._stream()), "esam3_attn_bias_gather_sum")
    return out


# ---- layers -------------------------------------------------------------------------------------------------------------------------------------
class LayerNormTrain:
    def __init__(self, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
        self.gamma, self.beta, self.eps = gamma, beta, eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.x = x
        y, self.mean, self.rstd = layernorm_forward(x, self.gamma, self.beta, self.eps)
        return y

    def backward(self, dy: torch.Tensor):
        dx, dgamma, dbeta = layernorm_backward(self.x, dy, self.gamma, self.mean, self.rstd)
        return dx, {"weight": dgamma, "bias": dbeta}


class LinearTrain:
    """nn.Linear on [..., K] rows: weight [N, K], bias [N] (device fp32 views)"""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor):
        self.w, self.b = weight, bias

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.x = x
        return tb.linear_forward(x, self.w, self.b)

    def backward(self, dy: torch.Tensor):
        return tb.linear_dgrad(dy, self.w), {"weight": tb.linear_wgrad(dy, self.x), "bias": tb.colsum(dy)}


class WindowAttentionTrain:
    """``Attention(dim, key_dim = 32, num_heads, attn_ratio = 1, resolution = (ws, ws))`` (tiny_vit.py:219-293) on window rows [windows, N, C].
    ``params``: norm.weight / .bias, qkv.weight / .bias, proj.weight / .bias, attention_biases [heads, ws * ws]."""

    def __init__(self, params: dict, heads: int, ws: int):
        self.norm = LayerNormTrain(params["norm.weight"], params["norm.bias"])
        self.qkv = LinearTrain(params["qkv.weight"], params["qkv.bias"])
        self.proj = LinearTrain(params["proj.weight"], params["proj.bias"])
        self.biases, self.heads, self.ws, self.scale = params["attention_biases"], heads, ws, HEAD_DIM ** -0.5

    def forward(self, xw: torch.Tensor) -> torch.Tensor:
        self.qkv_out = self.qkv.forward(self.norm.forward(xw))
        self.ab = attn_bias_gather(self.biases, self.ws)
        self.out, self.lse = win_attn_forward(self.qkv_out, self.ab, self.heads, self.scale, tab=self.biases, ws=self.ws)
        return self.proj.forward(self.out)

    def backward(self, dy: torch.Tensor):
        d_out, g_proj = self.proj.backward(dy)
        dqkv, dbias_full = win_attn_backward(self.qkv_out, self.ab, self.out, self.lse, d_out, self.heads, self.scale, tab=self.biases, ws=self.ws)
        d_norm, g_qkv = self.qkv.backward(dqkv)
        dx, g_norm = self.norm.backward(d_norm)
        grads = {"attention_biases": attn_bias_grad(dbias_full, self.ws)}
        for name, g in (("norm", g_norm), ("qkv", g_qkv), ("proj", g_proj)):
            grads.update({f"{name}.{k}": v for k, v in g.items()})
        return dx, grads


def _conv_bn(get, has, base: str, kind: str, act=None, stride: int = 1):
    """``Conv2d_BN`` (tiny_vit.py:31-41) under ``base`` -> (ConvLayerTrain, {grad key: state-dict name})"""
    w = get(base + ".c.weight")
    layer = tb.ConvLayerTrain(kind, w.reshape(w.shape[0], w.shape[1]) if kind == "pw" else w, get(base + ".bn.weight"), get(base + ".bn.bias"), act,
                              stride=stride, running_mean=get(base + ".bn.running_mean") if has(base + ".bn.running_mean") else None,
                              running_var=get(base + ".bn.running_var") if has(base + ".bn.running_var") else None)
    return layer, {"weight": base + ".c.weight", "gamma": base + ".bn.weight", "beta": base + ".bn.bias"}


class _Residual:
    """x + factor[b] * branch with the per-sample DropPath factor (1 when the branch has no stochastic depth), and its two gradients"""

    def __init__(self, name: str, rate: float, sampler: Optional[Callable]):
        self.name, self.rate, self.sampler = name, float(rate), sampler
        self._ones = None

    def factors(self, call: int, like: torch.Tensor) -> torch.Tensor:
        b, c = like.shape[0], like.shape[-1]
        if self.rate == 0.0:
            if self._ones is None or tuple(self._ones.shape) != (b, c):
                self._ones = torch.ones((b, c), dtype=torch.float32, device=like.device)
            return self._ones
        f = self.sampler(self.name, call, b, 1.0 - self.rate)
        return torch.as_tensor(f, dtype=torch.float32).reshape(b, 1).expand(b, c).contiguous().to(like.device)

    def forward(self, call: int, x: torch.Tensor, branch: torch.Tensor):
        f = self.factors(call, x)
        return tb.channel_scale(branch, f, add=x), f

    @staticmethod
    def backward(dy: torch.Tensor, f: torch.Tensor) -> torch.Tensor:
        """the branch's share of dy (the shortcut's share is dy itself)"""
        return tb.channel_scale(dy, f)


def _sum(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return tb.add(a, b)


class MBConvTrain:
    """``MBConv(C, C, 4, GELU, drop_path)`` (tiny_vit.py:87-125): act3(shortcut + drop_path(conv3(act2(conv2(act1(conv1(x)))))))"""

    def __init__(self, get, has, base: str, rate: float, sampler):
        self.parts = [_conv_bn(get, has, base + ".conv1", "pw", act="gelu"), _conv_bn(get, has, base + ".conv2", "dw", act="gelu"),
                      _conv_bn(get, has, base + ".conv3", "pw")]
        self.res = _Residual(base + ".drop_path", rate, sampler)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = x
        for layer, _ in self.parts:
            h = layer.forward(h)
        self.s, self.f = self.res.forward(0, x, h)
        return tb.act_forward(self.s, "gelu")

    def backward(self, dy: torch.Tensor):
        ds = tb.act_backward(self.s, dy, "gelu")
        d = self.res.backward(ds, self.f)
        grads = {}
        for layer, back in reversed(self.parts):
            d, g = layer.backward(d)
            grads.update({back[k]: v for k, v in g.items()})
        return _sum(d, ds), grads

    def norm_layers(self):
        return [(back["gamma"][:-len(".weight")], layer) for layer, back in self.parts]


class PatchMergingTrain:
    """``PatchMerging`` (tiny_vit.py:128-154): 1x1 + BN, GELU, depthwise 3x3 stride 2 + BN, GELU, 1x1 + BN"""

    def __init__(self, get, has, base: str):
        self.parts = [_conv_bn(get, has, base + ".conv1", "pw", act="gelu"), _conv_bn(get, has, base + ".conv2", "dw", act="gelu", stride=2),
                      _conv_bn(get, has, base + ".conv3", "pw")]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for layer, _ in self.parts:
            x = layer.forward(x)
        return x

    def backward(self, dy: torch.Tensor):
        grads = {}
        for layer, back in reversed(self.parts):
            dy, g = layer.backward(dy)
            grads.update({back[k]: v for k, v in g.items()})
        return dy, grads

    norm_layers = MBConvTrain.norm_layers


class TinyViTBlockTrain:
    """``TinyViTBlock`` (tiny_vit.py:296-386) on an NHWC map [B, H, W, C] under the state-dict prefix ``base``"""

    def __init__(self, get, has, base: str, heads: int, ws: int, rate: float, sampler):
        names = ("norm.weight", "norm.bias", "qkv.weight", "qkv.bias", "proj.weight", "proj.bias", "attention_biases")
        self.base, self.ws = base, ws
        self.attn = WindowAttentionTrain({n: get(f"{base}.attn.{n}") for n in names}, heads, ws)
        self.attn_back = {n: f"{base}.attn.{n}" for n in names}
        self.local_conv, self.local_back = _conv_bn(get, has, base + ".local_conv", "dw")
        self.mlp_norm = LayerNormTrain(get(base + ".mlp.norm.weight"), get(base + ".mlp.norm.bias"))
        self.fc1 = LinearTrain(get(base + ".mlp.fc1.weight"), get(base + ".mlp.fc1.bias"))
        self.fc2 = LinearTrain(get(base + ".mlp.fc2.weight"), get(base + ".mlp.fc2.bias"))
        self.res = _Residual(base + ".drop_path", rate, sampler)

    # window partition / reverse (tiny_vit.py:350-374): data movement
    def _partition(self, x: torch.Tensor) -> torch.Tensor:
        b, h, w, c = x.shape
        ws = self.ws
        if h == ws and w == ws:
            return x.reshape(b, h * w, c)
        if x.is_cuda and x.is_contiguous() and (c * x.element_size()) % 16 == 0:   # one copy on the device (esam3_window_partition)
            nwy, nwx = (h + ws - 1) // ws, (w + ws - 1) // ws
            out = torch.empty((b * nwy * nwx, ws * ws, c), dtype=x.dtype, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().esam3_window_partition(_DT[x.dtype], x.data_ptr(), out.data_ptr(), b, h, w, c, ws, 0, tb._stream()), "esam3_window_partition")
            return out
        pad_b, pad_r = (ws - h % ws) % ws, (ws - w % ws) % ws
        if pad_b or pad_r:
            x = torch.nn.functional.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        ph, pw = h + pad_b, w + pad_r
        return x.view(b, ph // ws, ws, pw // ws, ws, c).transpose(2, 3).reshape(b * (ph // ws) * (pw // ws), ws * ws, c).contiguous()

    def _reverse(self, xw: torch.Tensor, shape) -> torch.Tensor:
        b, h, w, c = shape
        ws = self.ws
        if h == ws and w == ws:
            return xw.reshape(b, h, w, c)
        if xw.is_cuda and xw.is_contiguous() and (c * xw.element_size()) % 16 == 0:
            out = torch.empty((b, h, w, c), dtype=xw.dtype, device=xw.device)
            with torch.cuda.device(xw.device):
                _lib.check(_lib.load().esam3_window_partition(_DT[xw.dtype], xw.data_ptr(), out.data_ptr(), b, h, w, c, ws, 1, tb._stream()), "esam3_window_partition")
            return out
        ph, pw = (h + ws - 1) // ws * ws, (w + ws - 1) // ws * ws
        x = xw.view(b, ph // ws, pw // ws, ws, ws, c).transpose(2, 3).reshape(b, ph, pw, c)
        return x[:, :h, :w].contiguous()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.shape = tuple(x.shape)
        a = self._reverse(self.attn.forward(self._partition(x)), self.shape)
        x1, self.f1 = self.res.forward(0, x, a)
        x2 = self.local_conv.forward(x1)
        self.z1 = self.fc1.forward(self.mlp_norm.forward(x2))
        m = self.fc2.forward(tb.act_forward(self.z1, "gelu"))
        out, self.f2 = self.res.forward(1, x2, m)
        return out

    def backward(self, dy: torch.Tensor):
        grads = {}
        d, g = self.fc2.backward(self.res.backward(dy, self.f2))
        grads.update({f"{self.base}.mlp.fc2.{k}": v for k, v in g.items()})
        d, g = self.fc1.backward(tb.act_backward(self.z1, d, "gelu"))
        grads.update({f"{self.base}.mlp.fc1.{k}": v for k, v in g.items()})
        d, g = self.mlp_norm.backward(d)
        grads.update({f"{self.base}.mlp.norm.{k}": v for k, v in g.items()})
        d_x2 = _sum(d, dy)
        d_x1, g = self.local_conv.backward(d_x2)
        grads.update({self.local_back[k]: v for k, v in g.items()})
        d_a = self.res.backward(d_x1, self.f1)
        d_w, g = self.attn.backward(self._partition(d_a))          # padding rows of the gradient are zeros, as autograd's for F.pad
        grads.update({self.attn_back[k]: v for k, v in g.items()})
        return _sum(self._reverse(d_w, self.shape), d_x1), grads

    def norm_layers(self):
        return [(self.local_back["gamma"][:-len(".weight")], self.local_conv)]


# ---- the trunk ----------------------------------------------------------------------------------------------------------------------------------
class TinyViTTrunkTrain:
    """``TinyViTAdapter`` (stage1/model.py:299-324) in training mode from a state dict in the reference's names (``patch_embed.seq.0.c.weight``,
    ``layers.1.blocks.0.attn.qkv.weight`` ...).  Same interface as ``train_blocks.EfficientViTTrunkTrain``."""

    def __init__(self, sd: Dict[str, torch.Tensor], model_name: str, dtype: torch.dtype = torch.float32, prefix: str = "",
                 drop_path_sampler: Optional[Callable] = None, seed: Optional[int] = None):
        dims, depths, heads, windows = TINYVIT_CFG[model_name]
        get = lambda k: sd[prefix + k]  # noqa: E731
        has = lambda k: (prefix + k) in sd  # noqa: E731
        opt = lambda k: sd.get(prefix + k)  # noqa: E731
        self.shapes = {k[len(prefix):]: tuple(v.shape) for k, v in sd.items() if k.startswith(prefix)}
        # The reference seeds every rank differently (train_image_encoder_stage1.py:340: seed = config.SEED + dist.get_rank()), so the
        # data-parallel ranks draw DIFFERENT stochastic-depth masks; `seed` is the base seed, the rank is added here.
        import torch.distributed as _dist
        rank = _dist.get_rank() if (_dist.is_available() and _dist.is_initialized()) else 0
        self._gen = torch.Generator().manual_seed((0 if seed is None else int(seed)) + rank)
        self._own_sampler = drop_path_sampler is None
        self._pre = {}        # (name, call) -> device row of this step's factors, drawn and uploaded in ONE piece by forward()
        sampler = drop_path_sampler or self._draw
        rates = np.linspace(0.0, DROP_PATH_RATE[model_name], sum(depths)).tolist()       # tiny_vit.py:491
        self.stem1 = tb.StemConvTrain(get("patch_embed.seq.0.c.weight"), get("patch_embed.seq.0.bn.weight"), get("patch_embed.seq.0.bn.bias"), dtype,
                                      act="gelu", running_mean=opt("patch_embed.seq.0.bn.running_mean"), running_var=opt("patch_embed.seq.0.bn.running_var"))
        self.stem2 = Conv3x3S2Train(get("patch_embed.seq.2.c.weight"), get("patch_embed.seq.2.bn.weight"), get("patch_embed.seq.2.bn.bias"),
                                    running_mean=opt("patch_embed.seq.2.bn.running_mean"), running_var=opt("patch_embed.seq.2.bn.running_var"))
        self.blocks = []
        k = 0
        for li, depth in enumerate(depths):
            for bi in range(depth):
                base = f"layers.{li}.blocks.{bi}"
                self.blocks.append(MBConvTrain(get, has, base, rates[k], sampler) if li == 0
                                   else TinyViTBlockTrain(get, has, base, heads[li], windows[li], rates[k], sampler))
                k += 1
            if li < len(depths) - 1:
                self.blocks.append(PatchMergingTrain(get, has, f"layers.{li}.downsample"))

    def _draw(self, name: str, call: int, batch: int, keep: float) -> torch.Tensor:
        """timm.layers.drop_path with scale_by_keep: bernoulli(keep) / keep per sample"""
        pre = self._pre.pop((name, call), None)
        if pre is not None and pre.shape[0] == batch:
            return pre
        return torch.empty(batch, dtype=torch.float32).bernoulli_(keep, generator=self._gen) / keep

    def _predraw(self, batch: int, device) -> None:
        """All stochastic-depth factors of one step, drawn in forward order from the trunk's generator (the same sequence the per-block
        draws would consume) and moved to the device in ONE copy: round 5 did a synchronous pageable upload per residual."""
        self._pre = {}
        todo = []
        for blk in self.blocks:
            res = getattr(blk, "res", None)
            if res is None or res.rate == 0.0:
                continue
            for call in range(2 if isinstance(blk, TinyViTBlockTrain) else 1):
                todo.append((res.name, call, 1.0 - res.rate))
        if not todo:
            return
        host = torch.stack([torch.empty(batch, dtype=torch.float32).bernoulli_(keep, generator=self._gen) / keep for _, _, keep in todo])
        dev = host.to(device, non_blocking=True)
        for i, (name, call, _) in enumerate(todo):
            self._pre[(name, call)] = dev[i]

    def rng_state(self) -> torch.Tensor:
        """state of the stochastic-depth generator (checkpoint it next to state_dict(): a resumed run continues the mask sequence)"""
        return self._gen.get_state()

    def set_rng_state(self, state: torch.Tensor) -> None:
        self._gen.set_state(state)

    def forward(self, img_nchw_f32: torch.Tensor) -> torch.Tensor:
        if self._own_sampler:
            self._predraw(img_nchw_f32.shape[0], img_nchw_f32.device)
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
            put("patch_embed.seq.2." + suffix, g[key])
        _, g = self.stem1.backward(d)
        for key, suffix in (("weight", "c.weight"), ("gamma", "bn.weight"), ("beta", "bn.bias")):
            put("patch_embed.seq.0." + suffix, g[key])
        return grads

    def norm_layers(self):
        out = [("patch_embed.seq.0.bn", self.stem1), ("patch_embed.seq.2.bn", self.stem2)]
        for blk in self.blocks:
            out.extend(blk.norm_layers())
        return out
