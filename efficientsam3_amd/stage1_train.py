"""One stage-1 distillation TRAINING STEP of an EfficientViT, RepViT or TinyViT student on the HIP kernels: the nine
students of stage1/model.py:386-420.

Reference: ``stage1/train_image_encoder_stage1.py:165-226`` (``train_one_epoch``: ``model.train()`` -> forward of
``ImageStudentEncoder`` (``stage1/model.py:188-211``: backbone -> Conv1x1 + BatchNorm + GELU -> Conv3x3 -> bilinear resize to the
embedding size) -> ``masked_mse`` + ``COSINE x masked_cosine_loss`` against the saved teacher embeddings, / ACCUMULATION_STEPS
-> ``loss_scaler(loss, optimizer, clip_grad, parameters, update_grad)`` (``stage1/utils.py:341-368``: scale, backward, unscale,
clip, AdamW step, scaler update) -> ``optimizer.zero_grad()``), optimizer from ``stage1/optimizer.py:6-46``.

Everything that is a tensor lives on the GPU for the whole step:

* the trainable parameters, their gradients and the two AdamW moments are four flat fp32 arenas (``stage1.Stage1Updater``); the
  layer objects hold VIEWS of the parameter arena, so the update kernel's result is what the next forward reads -- nothing is
  re-uploaded, re-packed on the host or synchronised (``train_blocks``: the ``esam3_train_*`` entry points pack on the device);
* BatchNorm running statistics are device buffers loaded from / exported to the state dict under the reference's names
  (``...norm.running_mean``, ``...running_var``, ``...num_batches_tracked``);
* every gradient is written into its view of the gradient arena the moment it exists (last layer first).  On an updating
  micro-step the backward pass's sink hands it to ``dist.GradientAllReducer.push`` right there -- the bucketed averaging
  all-reduce over RCCL that ``DistributedDataParallel`` performs for the reference (``train_image_encoder_stage1.py:67-72``):
  a bucket goes on the wire while the layers in front of it are still in their backward pass.  (The very first updating step
  learns the arrival order -- the bucket order -- and pushes after its backward pass; every later one overlaps.)
  ``Stage1Updater.step`` then un-scales, clips and applies AdamW.

The module owns no arithmetic: it sequences kernels (``train_blocks``, ``stage1``) and moves views around."""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from . import train_blocks as tb
from .stage1 import ArenaLayout, Stage1Updater, distill_loss, distill_loss_backward, valid_mask

_DT = {torch.float32: 0, torch.bfloat16: 1}
# EfficientViT-B0 / B1 / B2 = the EV-S / EV-M / EV-L students (backbones/efficientvit/efficientvit/backbone.py:169-190: width_list,
# depth_list, dim of the LiteMLA heads)
EFFICIENTVIT = {"b0": ([8, 16, 32, 64, 128], [1, 2, 2, 2, 2], 16), "b1": ([16, 32, 64, 128, 256], [1, 2, 3, 3, 4], 16),
                "b2": ([24, 48, 96, 192, 384], [1, 3, 4, 4, 6], 32)}


# the RepViT students RV-S / RV-M / RV-L (stage1/model.py:386-395: MODEL.BACKBONE repvit_m0_9 | repvit_m1_1 | repvit_m2_3) -> the key of
# schema.REPVIT_CFG; layers in train_repvit.py
REPVIT = {"repvit_m0_9": "m0.9", "repvit_m1_1": "m1.1", "repvit_m2_3": "m2.3"}
# the TinyViT students TV-S / TV-M / TV-L (stage1/model.py:397-406: tiny_vit_5m | tiny_vit_11m | tiny_vit_21m) -> the key of
# schema.TINYVIT_CFG; layers in train_tinyvit.py
TINYVIT = {"tiny_vit_5m": "5m", "tiny_vit_11m": "11m", "tiny_vit_21m": "21m"}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def conv3x3_forward(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """dense 3x3, padding 1, NHWC; w [Cout, Cin, 3, 3] device fp32"""
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    out = torch.empty((b, h, wd, cout), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    nb = lib.esam3_train_conv3x3_workspace(_DT[x.dtype], b, h, wd, cin, cout)   # packed weights + a bordered copy of x (the tile-GEMM form)
    ws = tb._ws(nb, x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_train_conv3x3_ws(_DT[x.dtype], x.data_ptr(), tb._dev_f32(w).data_ptr(), None if bias is None else tb._dev_f32(bias).data_ptr(),
                                              out.data_ptr(), b, h, wd, cin, cout, 0, ws.data_ptr(), nb, _stream()), "esam3_train_conv3x3_ws")
    return out


def conv3x3_dgrad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """dx of the same conv: the 3x3 conv of dy with the rotated, channel-transposed weight (packed that way on the device)"""
    b, h, wd, cout = dy.shape
    cin = w.shape[1]
    dx = torch.empty((b, h, wd, cin), dtype=dy.dtype, device=dy.device)
    lib = _lib.load()
    nb = lib.esam3_train_conv3x3_workspace(_DT[dy.dtype], b, h, wd, cout, cin)
    ws = tb._ws(nb, dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(lib.esam3_train_conv3x3_ws(_DT[dy.dtype], dy.data_ptr(), tb._dev_f32(w).data_ptr(), None, dx.data_ptr(), b, h, wd, cout, cin, 1,
                                              ws.data_ptr(), nb, _stream()), "esam3_train_conv3x3_ws")
    return dx


def conv3x3_wgrad(dy: torch.Tensor, x: torch.Tensor, stride: int = 1) -> torch.Tensor:
    """dw [Cout, Cin, 3, 3] fp32 of the dense 3x3 conv (padding 1, stride 1 | 2): tap (ky, kx) is the 1x1 weight gradient of dy against x shifted
    by (ky - 1, kx - 1) (zero outside the image) -- ``esam3_conv3x3_wgrad``: ONE launch of the weight-gradient GEMM with the nine taps on
    grid.z and the shifted operand gathered in the kernel (round 4 made nine shifted, zero-padded copies and nine launches here)"""
    b, h, wd, cin = x.shape
    cout = dy.shape[-1]
    assert x.is_contiguous() and dy.is_contiguous() and dy.dtype == x.dtype
    lib = _lib.load()
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    ws = tb._ws(lib.esam3_conv3x3_wgrad_workspace(b, h, wd, cin, cout, stride), x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_conv3x3_wgrad(_DT[x.dtype], dy.data_ptr(), x.data_ptr(), b, h, wd, cin, cout, stride, dw.data_ptr(), ws.data_ptr(), _stream()),
                   "esam3_conv3x3_wgrad")
    return dw


def resize_forward(x: torch.Tensor, size: int) -> torch.Tensor:
    b, h, w, c = x.shape
    out = torch.empty((b, size, size, c), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().esam3_op_resize_bilinear(_DT[x.dtype], x.data_ptr(), out.data_ptr(), b, h, w, size, size, c, _stream()),
                   "esam3_op_resize_bilinear")
    return out


def resize_backward(dy: torch.Tensor, in_hw: Tuple[int, int]) -> torch.Tensor:
    b, oh, ow, c = dy.shape
    dx = torch.empty((b, in_hw[0], in_hw[1], c), dtype=dy.dtype, device=dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(_lib.load().esam3_resize_bilinear_backward(_DT[dy.dtype], dy.data_ptr(), dx.data_ptr(), b, in_hw[0], in_hw[1], oh, ow, c, _stream()),
                   "esam3_resize_bilinear_backward")
    return dx


class HeadTrain:
    """``ImageStudentEncoder.head`` + the final resize (stage1/model.py:193-211): Conv2d(Cin, E, 1, bias=False) -> BatchNorm2d(E) -> GELU ->
    Conv2d(E, E, 3, padding=1) -> F.interpolate((S, S), bilinear) when the map is not S x S already.  Parameters under ``head.0.weight``,
    ``head.1.weight`` / ``.bias`` (+ running statistics), ``head.3.weight`` / ``.bias``."""

    def __init__(self, sd: Dict[str, torch.Tensor], embed_size: int, prefix: str = "head."):
        g = lambda k: sd[prefix + k]  # noqa: E731
        w0 = g("0.weight")
        self.l0 = tb.ConvLayerTrain("pw", w0.reshape(w0.shape[0], w0.shape[1]), g("1.weight"), g("1.bias"), act="gelu",
                                    running_mean=sd.get(prefix + "1.running_mean"), running_var=sd.get(prefix + "1.running_var"))
        self.w3, self.b3, self.embed_size, self.prefix = g("3.weight"), g("3.bias"), embed_size, prefix

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.a = self.l0.forward(x)
        c = conv3x3_forward(self.a, self.w3, self.b3)
        self.hw = tuple(c.shape[1:3])
        return c if self.hw == (self.embed_size, self.embed_size) else resize_forward(c, self.embed_size)

    def backward(self, dy: torch.Tensor, sink=None):
        d_c = dy if self.hw == (self.embed_size, self.embed_size) else resize_backward(dy, self.hw)
        grads = {}

        def put(name, gval, shape):
            grads[self.prefix + name] = gval.reshape(shape)
            if sink is not None:
                sink(self.prefix + name, grads[self.prefix + name])

        put("3.bias", tb.colsum(d_c), self.b3.shape)
        put("3.weight", conv3x3_wgrad(d_c, self.a), self.w3.shape)
        d_a = conv3x3_dgrad(d_c, self.w3)
        dx, g0 = self.l0.backward(d_a)
        put("1.weight", g0["gamma"], g0["gamma"].shape)
        put("1.bias", g0["beta"], g0["beta"].shape)
        put("0.weight", g0["weight"], tuple(self.l0.w.shape) + (1, 1))
        return dx, grads


class Stage1Trainer:
    """A stage-1 student (EfficientViT-B0 / B1 / B2, RepViT-M0.9 / M1.1 / M2.3 or TinyViT-5M / 11M / 21M backbone + head) that trains: ``step(images, teacher, sizes_before_pad)`` is one
    iteration of ``train_one_epoch`` (module docstring).  ``state_dict`` is the reference ``ImageStudentEncoder``'s
    (``backbone.model.<EfficientViTBackbone keys>``, ``head.*``), BatchNorm buffers included; ``state_dict()`` returns it back (fp32, host)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], model_name: str = "b1", embed_size: int = 72, dtype: str = "f32", device="cuda",
                 lr: float = 5e-4, weight_decay: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8, clip_grad: float = 5.0, amp: bool = False,
                 cosine_weight: float = 0.0, accumulation_steps: int = 1, init_scale: float = 65536.0, growth_interval: int = 2000,
                 bn_momentum: float = 0.1, group=None, force_collective: bool = False, drop_path_sampler=None, seed: Optional[int] = None,
                 sync_bn: bool = False, bucket_bytes: int = 25 << 20):
        from .dist import GradientAllReducer
        self.device = torch.device(device)
        tb.DEVICE = str(self.device)
        # --use-sync-bn (train_image_encoder_stage1.py:62-63): every BatchNorm of trunk and head uses the statistics of all ranks of `group`.
        # The setting is carried by THIS trainer's BatchNorm layers (`layer.sync`, set below next to the momentum), not by a module global:
        # a second trainer in the process (another group, no SyncBatchNorm, an evaluation copy) does not change this one.
        import torch.distributed as _dist
        if sync_bn and not _dist.is_initialized():
            import warnings
            warnings.warn("Stage1Trainer(sync_bn=True) without an initialised torch.distributed process group: BatchNorm uses this "
                          "rank's own batch statistics (initialise the process group before building the trainer)")
        self._sync = (group if group is not None else True) if (sync_bn and _dist.is_initialized()) else None
        self.tdtype = {"f32": torch.float32, "bf16": torch.bfloat16}[dtype]
        if model_name not in EFFICIENTVIT and model_name not in REPVIT and model_name not in TINYVIT:
            raise ValueError(f"stage-1 student {model_name!r}: the trainer covers {sorted(EFFICIENTVIT) + sorted(REPVIT) + sorted(TINYVIT)}")
        self.model_name = model_name
        self.embed_size, self.cosine_weight, self.accumulation_steps = embed_size, float(cosine_weight), int(accumulation_steps)
        is_buffer = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked"))  # noqa: E731
        named_shapes = [(k, tuple(v.shape)) for k, v in state_dict.items() if not is_buffer(k)]
        self.layout = ArenaLayout(named_shapes)
        self.updater = Stage1Updater(self.layout, self.device, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps, clip_grad=clip_grad,
                                     amp=amp, init_scale=init_scale, growth_interval=growth_interval)
        self.updater.load_params(state_dict)
        # what the layers see: parameters = views of the arena, BatchNorm buffers = device copies of the state dict's
        self.buffers = {k: torch.as_tensor(v).to(self.device, torch.float32).contiguous() for k, v in state_dict.items()
                        if k.endswith(("running_mean", "running_var"))}
        self.batches_tracked = {k: int(v) for k, v in state_dict.items() if k.endswith("num_batches_tracked")}
        views = {name: self.updater.param(name) for name, _ in named_shapes}
        views.update(self.buffers)
        if model_name in EFFICIENTVIT:
            self.widths, self.depths, self.dim = EFFICIENTVIT[model_name]
            self.trunk = tb.EfficientViTTrunkTrain(views, self.widths, self.depths, self.dim, dtype=self.tdtype, prefix="backbone.model.")
        elif model_name in REPVIT:
            from .train_repvit import RepViTTrunkTrain
            self.trunk = RepViTTrunkTrain(views, REPVIT[model_name], dtype=self.tdtype, prefix="backbone.model.")
        else:
            from .train_tinyvit import TinyViTTrunkTrain
            self.trunk = TinyViTTrunkTrain(views, TINYVIT[model_name], dtype=self.tdtype, prefix="backbone.model.", drop_path_sampler=drop_path_sampler,
                                           seed=seed)
        for _, layer in self.trunk.norm_layers():
            layer.momentum = bn_momentum
            layer.sync = self._sync
        self.head = HeadTrain(views, embed_size)
        self.head.l0.momentum = bn_momentum
        self.head.l0.sync = self._sync
        self.names = [n for n, _ in named_shapes]
        # gradients arrive head first, then the trunk from its last layer to the stem: the bucket order of the all-reduce
        self._arrival = None
        self._arrival_index = None
        self._pushed = False
        self._reducer_cls, self._group, self._force = GradientAllReducer, group, force_collective
        # DistributedDataParallel's bucket size (25 MB), not the reducer's 64 MB default: gradients arrive head first and head.3.weight alone
        # is 38 MB -- with 64 MB buckets the head shares a bucket with the last trunk stages (for the small students: with the WHOLE trunk),
        # which completes only when the stem's gradient arrives, i.e. nothing overlapped.  A gradient larger than a bucket closes the
        # bucket before it and fills one of its own, so the head goes on the wire while the trunk is still in its backward pass.
        self._bucket_bytes = int(bucket_bytes)
        self.reducer = None
        self._micro = 0   # micro-steps accumulated since the last update
        self.last = {}

    # ---- forward / backward -------------------------------------------------------------------------------------------------
    def forward(self, images_nchw_f32: torch.Tensor) -> torch.Tensor:
        """student embeddings [B, S, S, E] (NHWC) in the compute dtype"""
        feats = self.trunk.forward(images_nchw_f32)
        return self.head.forward(feats)

    def backward(self, d_preds: torch.Tensor, push: bool = False) -> None:
        """fills the gradient arena (accumulating: ``+=`` into the views, as autograd accumulates into ``.grad``); with ``push`` (an
        updating micro-step in a multi-rank job, arrival order known) every finished gradient goes to the all-reduce at once"""
        order = []
        first = self._micro == 0   # first micro-step after an update: the arena was zeroed -> plain copies; later ones accumulate
        push = push and self.reducer is not None and self._arrival_index is not None
        self._pushed = push

        # one-rank / non-pushing steps: the ~190 per-parameter copies into the arena (5 us each, ~1 ms of a B1 batch-32 step, every one its own
        # launch) are collected and issued as ONE multi-tensor copy after the backward pass; a pushing step needs each gradient in the arena the
        # moment it is handed to the all-reduce and keeps the immediate copy
        pending_dst, pending_src = [], []

        def sink(name, gval):
            dst = self.updater.grad(name)
            if first and not push and gval.dtype == dst.dtype and gval.shape == dst.shape and gval.is_contiguous():  # one strided source sends the whole list down the per-tensor path
                pending_dst.append(dst)
                pending_src.append(gval)
            elif first:
                dst.copy_(gval)
            else:
                dst.add_(gval.to(torch.float32))
            order.append(name)
            if push:
                self.reducer.push(self._arrival_index[name], dst)

        d_feats, _ = self.head.backward(d_preds, sink=sink)
        self.trunk.backward(d_feats, sink=lambda n, gv: sink("backbone.model." + n, gv))
        if pending_dst:
            torch._foreach_copy_(pending_dst, pending_src)
        if self._arrival is None:
            self._arrival = order
            self._arrival_index = {n: i for i, n in enumerate(order)}
            assert sorted(order) == sorted(self.names), (set(self.names) ^ set(order))

    def step(self, images_nchw_f32: torch.Tensor, teacher: torch.Tensor, sizes_before_pad: Sequence[Tuple[int, int]], lr: Optional[float] = None,
             update_grad: bool = True) -> dict:
        """one iteration: forward, loss, backward (scaled by the loss scale / accumulation steps), all-reduce, update.
        ``teacher`` [B, S, S, E] (NHWC; fp32 / bf16 / fp16 as saved).  Returns device scalars: loss, grad_norm (None when
        ``update_grad`` is False: an accumulation step)."""
        b = images_nchw_f32.shape[0]
        preds = self.forward(images_nchw_f32)
        s, e = self.embed_size, preds.shape[-1]
        vkey = (int(images_nchw_f32.shape[-1]), tuple(tuple(int(v) for v in hw) for hw in sizes_before_pad), s)
        if getattr(self, "_valid_key", None) != vkey:   # the mask of the last batch's sizes stays on the device (a pageable upload synchronises)
            self._valid_key, self._valid = vkey, torch.from_numpy(valid_mask(images_nchw_f32.shape[-1], sizes_before_pad, (s, s))).to(self.device)
        valid = self._valid
        p2, t2 = preds.reshape(b, s * s, e), teacher.reshape(b, s * s, e)
        mse, cos, _ = distill_loss(p2, t2, valid)
        loss = (mse + self.cosine_weight * cos) / self.accumulation_steps
        # d(loss x scale) / d(preds).  The loss scale lives in the updater's device state (it moves when a step is skipped / after
        # growth_interval clean steps).  Rounds 4-5 read it back once per iteration (the reference's loop synchronises every iteration too:
        # loss.item(), torch.cuda.synchronize(), train_image_encoder_stage1.py:206,226); on the GPU path it now stays on the device
        if self.updater.amp and p2.is_cuda:
            # round 6: the scale is read by the kernel, from the updater's state (esam3_distill_loss_backward_ds): the step has no host
            # synchronisation left (the read-back was one, with ~1 ms of idle GPU behind it); a power of two, so the gradient is bit-identical
            d = distill_loss_backward(p2, t2, valid, cosine_weight=self.cosine_weight, grad_scale=1.0 / self.accumulation_steps, scale_dev=self.updater.state)
        else:
            scale = float(self.updater.loss_scale) if self.updater.amp else 1.0
            d = distill_loss_backward(p2, t2, valid, cosine_weight=self.cosine_weight, grad_scale=scale / self.accumulation_steps)
        self.backward(d.reshape(b, s, s, e), push=update_grad and self._collective())
        for k in self.batches_tracked:
            self.batches_tracked[k] += 1
        out = {"loss": loss, "grad_norm": None}
        self._micro += 1
        if update_grad:
            self._allreduce()
            out["grad_norm"] = self.updater.step(lr=lr).clone()
            self._micro = 0
        self.last = out
        return out

    def _collective(self) -> bool:
        import torch.distributed as dist
        return bool(dist.is_initialized() and (dist.get_world_size(self._group) > 1 or self._force))

    def _allreduce(self) -> None:
        if not self._collective():
            return
        grads = [self.updater.grad(n) for n in self._arrival]
        if self.reducer is None:
            self.reducer = self._reducer_cls(grads, bucket_bytes=self._bucket_bytes, group=self._group, force_collective=self._force)
        if not self._pushed:   # the first updating step (order unknown until its backward pass ended)
            for i, g_ in enumerate(grads):
                self.reducer.push(i, g_)
        self.reducer.finish(grads)

    # ---- state ----------------------------------------------------------------------------------------------------------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {n: self.updater.param(n).detach().cpu().clone() for n in self.names}
        for prefix, layer in [("backbone.model." + p, l) for p, l in self.trunk.norm_layers()] + [("head.1", self.head.l0)]:
            sd[prefix + ".running_mean"] = layer.running_mean.detach().cpu().clone()
            sd[prefix + ".running_var"] = layer.running_var.detach().cpu().clone()
        for k, v in self.batches_tracked.items():
            sd[k] = torch.tensor(v, dtype=torch.long)
        return sd

    def rng_state(self) -> Optional[torch.Tensor]:
        """the stochastic-depth generator's state (TinyViT 11M / 21M; None for students without DropPath): save it beside state_dict() and the
        optimizer state, hand it to ``set_rng_state`` on resume -- otherwise a resumed run replays the mask sequence from its start"""
        return self.trunk.rng_state() if hasattr(self.trunk, "rng_state") else None

    def set_rng_state(self, state: Optional[torch.Tensor]) -> None:
        if state is not None and hasattr(self.trunk, "set_rng_state"):
            self.trunk.set_rng_state(state)

    def gradients(self) -> Dict[str, torch.Tensor]:
        return {n: self.updater.grad(n).detach().cpu().clone() for n in self.names}


def adamw_state_dict(updater: Stage1Updater, names: Sequence[str]) -> dict:
    """the arena's optimizer state in ``torch.optim.AdamW.state_dict()`` layout for the two groups ``set_weight_decay`` builds
    (stage1/optimizer.py:32-46: has_decay first, then no_decay, each in ``named_parameters`` order; no lr_scale split): resumable by the
    reference's ``optimizer.load_state_dict``"""
    from .stage1 import weight_decay_groups
    shapes = dict(updater.layout.named_shapes)
    decay = weight_decay_groups([(n, shapes[n]) for n in names])
    order = [n for n in names if decay[n]] + [n for n in names if not decay[n]]
    st = updater.state.cpu()
    state = {i: {"step": torch.tensor(float(st[4])), "exp_avg": updater.view(updater.exp_avg, n).detach().cpu().clone(),
                 "exp_avg_sq": updater.view(updater.exp_avg_sq, n).detach().cpu().clone()} for i, n in enumerate(order)}
    n_decay = sum(1 for n in names if decay[n])
    common = dict(lr=updater.lr, betas=tuple(updater.betas), eps=updater.eps, amsgrad=False, maximize=False, foreach=None, capturable=False,
                  differentiable=False, fused=None)
    # lr_scale: the key divide_param_groups_by_lr_scale adds to every group (stage1/utils.py:557-620; 1.0 without layer decay)
    groups = [dict(common, weight_decay=updater.weight_decay, lr_scale=1.0, params=list(range(n_decay))),
              dict(common, weight_decay=0.0, lr_scale=1.0, params=list(range(n_decay, len(order))))]
    return {"state": state, "param_groups": groups, "param_names": order}


def load_adamw_state_dict(updater: Stage1Updater, names: Sequence[str], sd: dict) -> None:
    """the inverse of ``adamw_state_dict``: resume the arena's moments and step count from ``torch.optim.AdamW.state_dict()`` of the optimizer
    ``stage1/optimizer.py: build_optimizer`` builds (what ``save_checkpoint`` stores under "optimizer", stage1/utils.py:86-101).  Parameter i of
    the state is the i-th name of [has_decay ..., no_decay ...] in ``named_parameters`` order; the groups' lr / betas / eps / weight_decay become
    the updater's hyper-parameters (the scheduler overrides lr per iteration anyway)."""
    from .stage1 import weight_decay_groups
    shapes = dict(updater.layout.named_shapes)
    decay = weight_decay_groups([(n, shapes[n]) for n in names])
    order = [n for n in names if decay[n]] + [n for n in names if not decay[n]]
    groups = sd["param_groups"]
    flat = [i for g_ in groups for i in g_["params"]]
    if len(flat) != len(order):
        raise ValueError(f"optimizer state holds {len(flat)} parameters, the model has {len(order)}")
    n_decay = sum(1 for n in names if decay[n])
    if len(groups) >= 2 and len(groups[0]["params"]) != n_decay:
        raise ValueError(f"first group holds {len(groups[0]['params'])} parameters, {n_decay} take weight decay here")
    steps = set()
    for pos, idx in enumerate(flat):
        st = sd["state"].get(idx)
        name = order[pos]
        ea, es = updater.view(updater.exp_avg, name), updater.view(updater.exp_avg_sq, name)
        if st is None:          # a parameter that never received a gradient
            ea.zero_()
            es.zero_()
            continue
        if tuple(st["exp_avg"].shape) != tuple(ea.shape):
            raise ValueError(f"{name}: moment shape {tuple(st['exp_avg'].shape)} != parameter shape {tuple(ea.shape)}")
        ea.copy_(torch.as_tensor(st["exp_avg"], dtype=torch.float32))
        es.copy_(torch.as_tensor(st["exp_avg_sq"], dtype=torch.float32))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"parameters with different step counts {sorted(steps)}: the arena keeps one")
    state = updater.state.cpu()
    state[4] = float(steps.pop()) if steps else 0.0
    updater.state.copy_(state)
    g0 = groups[0]
    updater.lr, updater.betas, updater.eps = float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"])
    updater.weight_decay = float(g0["weight_decay"])
