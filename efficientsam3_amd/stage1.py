"""Stage-1 image distillation, forward pieces on the device (BASELINE config 5): the dataset's image
pipeline (ResizeLongestSide + ImageNet mean / std + bottom-right padding, stage1/data/sa1b_dataset.py:163-228 and
stage1/data/transforms.py:13-88), the loss between a student embedding and the saved teacher embedding
(stage1/train_image_encoder_stage1.py:271-307) and the teacher-embedding payload
(stage1/save_embedding_image_stage1.py:92-96: int32 augmentation seed ‖ fp16 [C, H, W]).  The trunks that produce the
embeddings are the engine's encoders (``engine.encode(..., want_trunk=True)``)."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

# SA1BDataset defaults (stage1/data/sa1b_dataset.py:22)
PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)


def get_preprocess_shape(oldh: int, oldw: int, long_side_length: int) -> Tuple[int, int]:
    """ResizeLongestSide.get_preprocess_shape (transforms.py:81-88) -> (new_h, new_w)."""
    import ctypes as C
    nh, nw = C.c_int(0), C.c_int(0)
    _lib.load().esam3_stage1_preprocess_shape(int(oldh), int(oldw), int(long_side_length), C.byref(nh), C.byref(nw))
    return nh.value, nw.value


def apply_coords(coords: np.ndarray, original_size: Tuple[int, int], target_length: int) -> np.ndarray:
    """ResizeLongestSide.apply_coords (transforms.py:35-42): (x, y) prompt coordinates into the resized frame."""
    old_h, old_w = original_size
    new_h, new_w = get_preprocess_shape(old_h, old_w, target_length)
    out = np.array(coords, dtype=np.float64, copy=True)
    out[..., 0] = out[..., 0] * (new_w / old_w)
    out[..., 1] = out[..., 1] * (new_h / old_h)
    return out


def apply_boxes(boxes: np.ndarray, original_size: Tuple[int, int], target_length: int) -> np.ndarray:
    """ResizeLongestSide.apply_boxes (transforms.py:44-46): xyxy boxes into the resized frame."""
    return apply_coords(np.asarray(boxes).reshape(-1, 2, 2), original_size, target_length).reshape(-1, 4)


def preprocess_sa1b(images_hwc_u8, img_size: int = 1008, pixel_mean: Sequence[float] = PIXEL_MEAN,
                    pixel_std: Sequence[float] = PIXEL_STD, device=None, out: torch.Tensor = None):
    """The image path of SA1BDataset.__getitem__ for a list of uint8 HWC images (numpy arrays or torch tensors, any sizes)
    -> (x [B, 3, img_size, img_size] fp32 on the device, sizes_before_pad [(new_h, new_w)] * B): longest side resized to
    ``img_size`` (fp32, antialiased), ImageNet normalisation, zero padding at the bottom / right.  ``sizes_before_pad`` is
    what ``valid_mask`` / ``paired_forward`` take (``img_size_before_pad`` of the dataset)."""
    import ctypes as C
    if isinstance(images_hwc_u8, (np.ndarray, torch.Tensor)) and images_hwc_u8.ndim == 3:
        images_hwc_u8 = [images_hwc_u8]
    dev = torch.device(device) if device is not None else (out.device if out is not None else torch.device("cuda"))
    if dev.type != "cuda":
        raise ValueError("preprocess_sa1b runs on the GPU only (there is no CPU fallback)")
    b = len(images_hwc_u8)
    if out is None:
        out = torch.empty((b, 3, img_size, img_size), dtype=torch.float32, device=dev)
    assert tuple(out.shape) == (b, 3, img_size, img_size) and out.dtype == torch.float32 and out.is_contiguous()
    mean = (C.c_float * 3)(*[float(v) for v in pixel_mean])
    std = (C.c_float * 3)(*[float(v) for v in pixel_std])
    lib = _lib.load()
    sizes = []
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        for i, img in enumerate(images_hwc_u8):
            t = torch.from_numpy(np.ascontiguousarray(img)) if isinstance(img, np.ndarray) else img
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[-1] != 3:
                raise ValueError(f"image {i}: expected uint8 [H, W, 3], got {t.dtype} {tuple(t.shape)}")
            t = t.to(dev, non_blocking=True).contiguous()
            nh, nw = C.c_int(0), C.c_int(0)
            _lib.check(lib.esam3_stage1_preprocess_u8(t.data_ptr(), int(t.shape[0]), int(t.shape[1]), out[i].data_ptr(), int(img_size),
                                                      mean, std, C.byref(nh), C.byref(nw), stream), "esam3_stage1_preprocess_u8")
            sizes.append((nh.value, nw.value))
    return out, sizes


def valid_mask(img_size: int, sizes_before_pad: Sequence[Tuple[int, int]], target_hw: Tuple[int, int]) -> np.ndarray:
    """build_valid_mask in closed form: the indicator of the (h, w) rectangle is separable, so its bilinear resize
    (align_corners=False, no antialias) is the product of two 1-D resizes -> uint8 [B, H*W] (1 = valid)."""
    def axis(n_valid: int, n_out: int) -> np.ndarray:
        scale = img_size / n_out
        src = np.maximum((np.arange(n_out, dtype=np.float32) + 0.5) * np.float32(scale) - 0.5, 0.0).astype(np.float32)
        i0 = np.minimum(np.floor(src).astype(np.int64), img_size - 1)
        i1 = np.minimum(i0 + 1, img_size - 1)
        lam = (src - i0).astype(np.float32)
        return (1.0 - lam) * (i0 < n_valid) + lam * (i1 < n_valid)
    th, tw = target_hw
    out = np.zeros((len(sizes_before_pad), th * tw), np.uint8)
    for i, (h, w) in enumerate(sizes_before_pad):
        out[i] = (np.outer(axis(h, th), axis(w, tw)).astype(np.float32) > 0.5).reshape(-1)
    return out


def distill_loss(preds: torch.Tensor, teacher: torch.Tensor, valid: torch.Tensor):
    """preds [B, HW, C] fp32 / bf16 and teacher [B, HW, C] fp32 / bf16 / fp16 (token-major, on the GPU),
    valid uint8 [B, HW] -> (masked_mse, masked_cosine_loss, per_image [B, 2])."""
    assert preds.is_cuda and teacher.is_cuda and valid.is_cuda
    assert preds.dim() == 3 and preds.shape == teacher.shape and preds.dtype in (torch.float32, torch.bfloat16)
    assert teacher.dtype in _DT and valid.dtype == torch.uint8 and tuple(valid.shape) == tuple(preds.shape[:2])
    b, hw, c = preds.shape
    preds, teacher, valid = preds.contiguous(), teacher.contiguous(), valid.contiguous()
    per = torch.empty((b, 2), dtype=torch.float32, device=preds.device)
    scratch = torch.empty((b * hw * 2,), dtype=torch.float32, device=preds.device)
    lib = _lib.load()
    with torch.cuda.device(preds.device):
        _lib.check(lib.esam3_distill_loss(_DT[preds.dtype], preds.data_ptr(), _DT[teacher.dtype], teacher.data_ptr(),
                                          valid.data_ptr(), b, hw, c, per.data_ptr(), scratch.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "esam3_distill_loss")
    m = per.mean(dim=0)
    return m[0], m[1], per


def distill_loss_backward(preds: torch.Tensor, teacher: torch.Tensor, valid: torch.Tensor, cosine_weight: float = 0.0,
                          grad_scale: float = 1.0, scale_dev: torch.Tensor = None) -> torch.Tensor:
    """dL/dpreds of ``masked_mse + cosine_weight * masked_cosine_loss`` times ``grad_scale`` (= 1 / ACCUMULATION_STEPS in
    stage1/train_image_encoder_stage1.py:186-210): [B, HW, C] in the dtype of ``preds``, zero at masked pixels -- the tensor
    ``loss.backward()`` hands to the student trunk."""
    assert preds.is_cuda and teacher.is_cuda and valid.is_cuda
    assert preds.dim() == 3 and preds.shape == teacher.shape and preds.dtype in (torch.float32, torch.bfloat16)
    assert teacher.dtype in _DT and valid.dtype == torch.uint8 and tuple(valid.shape) == tuple(preds.shape[:2])
    b, hw, c = preds.shape
    preds, teacher, valid = preds.contiguous(), teacher.contiguous(), valid.contiguous()
    grad = torch.empty_like(preds)
    scratch = torch.empty((b,), dtype=torch.float32, device=preds.device)
    lib = _lib.load()
    with torch.cuda.device(preds.device):
        if scale_dev is not None:   # times scale_dev[0], read on the device (the AMP loss scale: no host read-back)
            assert scale_dev.is_cuda and scale_dev.dtype == torch.float32
            _lib.check(lib.esam3_distill_loss_backward_ds(_DT[preds.dtype], preds.data_ptr(), _DT[teacher.dtype], teacher.data_ptr(),
                                                          valid.data_ptr(), b, hw, c, float(cosine_weight), float(grad_scale), scale_dev.data_ptr(),
                                                          grad.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "esam3_distill_loss_backward_ds")
            return grad
        _lib.check(lib.esam3_distill_loss_backward(_DT[preds.dtype], preds.data_ptr(), _DT[teacher.dtype], teacher.data_ptr(),
                                                   valid.data_ptr(), b, hw, c, float(cosine_weight), float(grad_scale),
                                                   grad.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "esam3_distill_loss_backward")
    return grad


def pack_embedding(seed: int, embedding_chw: np.ndarray) -> bytes:
    return np.int32(seed).tobytes() + np.ascontiguousarray(embedding_chw, dtype=np.float16).tobytes()


def unpack_embedding(payload: bytes, shape_chw: Tuple[int, int, int]):
    """-> (seed, fp16 [C, H, W]) as dataset_wrapper.py:50-62 parses it."""
    seed = int(np.frombuffer(payload[:4], dtype=np.int32)[0])
    n = int(np.prod(shape_chw))
    if len(payload) < 4 + 2 * n:
        raise ValueError(f"payload of {len(payload)} bytes is too short for an embedding of shape {shape_chw}")
    return seed, np.frombuffer(payload[4:4 + 2 * n], dtype=np.float16).copy().reshape(shape_chw)


def paired_forward(teacher, student, images: torch.Tensor, sizes_before_pad: Sequence[Tuple[int, int]]):
    """BASELINE config 5, forward half: the ViT-H teacher trunk and a student trunk on the same normalised images
    [B, 3, 1008, 1008], then the distillation loss between the two [B, 1024, 72, 72] embeddings
    (stage1/model.py:237 ``model.backbone.vision_backbone.trunk(x)`` on both sides,
    stage1/train_image_encoder_stage1.py:196-212).  ``teacher`` / ``student`` are models built by
    ``build_sam3_image_model`` / ``build_efficientsam3_image_model`` on the same device.
    -> dict(mse, cosine, per_image [B, 2], teacher [B, 5184, 1024], student [B, 5184, 1024])."""
    x = images.to(student.device, torch.float32).contiguous()
    t = teacher.engine.encode(x, want_sam3=False, want_sam2=False, want_trunk=True)["trunk"]
    s = student.engine.encode(x, want_sam3=False, want_sam2=False, want_trunk=True)["trunk"]
    b, h, w, c = s.shape
    assert tuple(t.shape) == (b, h, w, c), (t.shape, s.shape)
    valid = torch.from_numpy(valid_mask(x.shape[-1], sizes_before_pad, (h, w))).to(x.device)
    tt, ss = t.reshape(b, h * w, c), s.reshape(b, h * w, c)
    mse, cos, per = distill_loss(ss, tt, valid)
    return {"mse": mse, "cosine": cos, "per_image": per, "teacher": tt, "student": ss, "valid": valid}


# ---- offline teacher embeddings (stage 1 is offline: SURVEY.md 0.5) -----------------------------------------------------------
class EmbeddingStore:
    """The on-disk format of the reference's ``TxtManager`` (stage1/data/augmentation/manager.py:7-165): a directory with,
    per writing rank, ``rank{r}-keys.txt`` (one key per line, in write order, duplicates skipped) and ``rank{r}-values.bin``
    (the payloads back to back, ``item_size`` bytes each).  Files written here are read by the reference's ``_Reader`` and
    vice versa.  Writing is synchronous (the reference hands the payloads to a worker process; the format is the same)."""

    def __init__(self, path: str, item_size: int, rank: int = 0):
        self.path, self.item_size, self.rank = path, int(item_size), int(rank)
        self._keys_file = self._values_file = None
        self._written = {}
        self._index = None      # key -> (package name, position), filled lazily by read()
        self._handles = {}

    # -- writing --
    def write(self, key: str, payload: bytes) -> bool:
        if len(payload) != self.item_size:
            raise ValueError(f"payload of {len(payload)} bytes, item_size is {self.item_size}")
        if "\n" in key:
            raise ValueError("keys are stored one per line")
        if self._keys_file is None:
            import os
            os.makedirs(self.path, exist_ok=True)
            base = os.path.join(self.path, f"rank{self.rank}")
            self._keys_file = open(base + "-keys.txt", "w")
            self._values_file = open(base + "-values.bin", "wb")
        if key in self._written:
            return True
        self._written[key] = len(self._written)
        self._keys_file.write(key + "\n")
        self._values_file.write(payload)
        return True

    def close(self) -> None:
        for f in (self._keys_file, self._values_file, *self._handles.values()):
            if f is not None:
                f.close()
        self._keys_file = self._values_file = None
        self._handles = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- reading --
    def _load_index(self) -> None:
        import os
        names = sorted(n[:-len("-values.bin")] for n in os.listdir(self.path) if n.endswith("-values.bin"))
        n = max(len(names), 1)
        names.sort(key=lambda s: (int(s[4:]) - self.rank) % n)      # this rank's own package first, as the reference does
        self._index = {}
        for name in reversed(names):                                # earlier packages win, like the reference's search order
            with open(os.path.join(self.path, name + "-keys.txt")) as f:
                for i, k in enumerate(f.readlines()):
                    self._index[k.strip()] = (name, i)

    def read(self, key: str) -> bytes:
        import os
        if self._index is None:
            self._load_index()
        name, i = self._index[key]
        h = self._handles.get(name)
        if h is None:
            h = self._handles[name] = open(os.path.join(self.path, name + "-values.bin"), "rb")
        h.seek(self.item_size * i)
        return h.read(self.item_size)


def embedding_item_size(shape_chw: Tuple[int, int, int]) -> int:
    """bytes of one payload: int32 seed + fp16 [C, H, W] (save_embedding_image_stage1.py:92-96)"""
    return 4 + 2 * int(np.prod(shape_chw))


def save_teacher_embeddings(teacher, batches, store: EmbeddingStore, img_size: int = 1008) -> int:
    """``save_embeddings_one_epoch`` (stage1/save_embedding_image_stage1.py:70-98) on the engine: for every batch
    ``(images_hwc_u8, keys, seeds)`` run the dataset's image pipeline (``preprocess_sa1b``) and the teacher trunk
    (``stage1/model.py:237``), cast the [B, C, H, W] embeddings to fp16 and write ``seed bytes + embedding bytes`` per image
    under its key.  The D2H copy goes through one pinned, reused buffer.  Returns the number of embeddings written."""
    pin = None
    n = 0
    for images, keys, seeds in batches:
        x, _ = preprocess_sa1b(images, img_size, device=teacher.device)
        t = teacher.engine.encode(x, want_sam3=False, want_sam2=False, want_trunk=True)["trunk"]       # [B, H, W, C]
        emb = t.permute(0, 3, 1, 2).to(torch.float16).contiguous()                                       # the reference's NCHW
        if pin is None or pin.shape != emb.shape:
            pin = torch.empty(emb.shape, dtype=torch.float16).pin_memory()
        pin.copy_(emb, non_blocking=True)
        torch.cuda.current_stream(emb.device).synchronize()
        arr = pin.numpy()
        for i, (key, seed) in enumerate(zip(keys, np.asarray(seeds).astype(np.int32))):
            store.write(key, pack_embedding(int(seed), arr[i]))
            n += 1
    return n


# ---- the update half of the training step ---------------------------------------------------------------------------------
CHUNK = 256  # arena granularity: every parameter starts at a multiple of CHUNK and one chunk belongs to one parameter group


def weight_decay_groups(named_shapes, skip_list=(), skip_keywords=()):
    """stage1/optimizer.py:32-53 (set_weight_decay, check_keywords_in_name): name -> True if the parameter is in the
    has_decay group; 1-D parameters, ``.bias`` and everything the model lists in no_weight_decay() / ..._keywords() are not."""
    out = {}
    for name, shape in named_shapes:
        no = len(tuple(shape)) == 1 or name.endswith(".bias") or name in skip_list or any(k in name for k in skip_keywords)
        out[name] = not no
    return out


class ArenaLayout:
    """Where each trainable parameter sits in the flat arena: ``offsets[name] = (start, numel)``, starts at multiples of CHUNK,
    padding elements stay zero (zero gradient, zero moments: AdamW leaves them at zero)."""

    def __init__(self, named_shapes, skip_list=(), skip_keywords=(), lr_scales=None):
        self.named_shapes = [(n, tuple(int(d) for d in s)) for n, s in named_shapes]
        decay = weight_decay_groups(self.named_shapes, skip_list, skip_keywords)
        lr_scales = lr_scales or {}
        self.offsets = {}
        pos, chunk_decay, chunk_scale = 0, [], []
        for name, shape in self.named_shapes:
            numel = int(np.prod(shape)) if len(shape) else 1
            nch = (numel + CHUNK - 1) // CHUNK
            self.offsets[name] = (pos, numel)
            chunk_decay += [1 if decay[name] else 0] * nch
            chunk_scale += [float(lr_scales.get(name, 1.0))] * nch   # getattr(p, 'lr_scale', 1.0), utils.py:595
            pos += nch * CHUNK
        self.n = pos
        self.chunk_decay = np.asarray(chunk_decay, dtype=np.uint8)
        self.chunk_lr_scale = np.asarray(chunk_scale, dtype=np.float32)


class Stage1Updater:
    """``loss_scaler(loss, optimizer, clip_grad, parameters, update_grad=True)`` + ``optimizer.zero_grad()`` of
    stage1/train_image_encoder_stage1.py:210-219 after the backward pass, for parameters held in one flat fp32 device arena:
    NativeScalerWithGradNormCount (stage1/utils.py:341-368) around torch.optim.AdamW as stage1/optimizer.py:6-29 builds it.
    ``grads`` is the buffer the backward pass accumulates SCALED gradients into (and what ``GradientAllReducer`` reduces);
    ``step()`` launches three kernels and never synchronises: the returned gradient norm is a device scalar."""

    def __init__(self, layout: ArenaLayout, device, lr: float = 5e-4, weight_decay: float = 0.05, betas=(0.9, 0.999),
                 eps: float = 1e-8, clip_grad: float = 5.0, amp: bool = True, init_scale: float = 65536.0,
                 growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                 keep_bf16_copy: bool = False):
        self.layout, self.device = layout, torch.device(device)
        self.lr, self.weight_decay, self.betas, self.eps, self.clip_grad = float(lr), float(weight_decay), betas, float(eps), clip_grad
        self.amp, self.growth_factor, self.backoff_factor, self.growth_interval = bool(amp), growth_factor, backoff_factor, growth_interval
        z = lambda: torch.zeros(layout.n, dtype=torch.float32, device=self.device)  # noqa: E731
        self.params, self.grads, self.exp_avg, self.exp_avg_sq = z(), z(), z(), z()
        self.bf16 = torch.zeros(layout.n, dtype=torch.bfloat16, device=self.device) if keep_bf16_copy else None
        self.state = torch.zeros(16, dtype=torch.float32, device=self.device)
        self.state[0] = init_scale if amp else 1.0
        self._decay = torch.from_numpy(layout.chunk_decay).to(self.device)
        self._scale = torch.from_numpy(layout.chunk_lr_scale).to(self.device)
        lib = _lib.load()
        self._ws = torch.empty(int(lib.esam3_stage1_update_workspace(layout.n)), dtype=torch.uint8, device=self.device)

    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        start, numel = self.layout.offsets[name]
        shape = dict(self.layout.named_shapes)[name]
        return buf[start:start + numel].view(shape)

    def param(self, name):
        return self.view(self.params, name)

    def grad(self, name):
        return self.view(self.grads, name)

    def load_params(self, state_dict) -> None:
        for name, _ in self.layout.named_shapes:
            self.param(name).copy_(torch.as_tensor(state_dict[name], dtype=torch.float32))

    @property
    def loss_scale(self) -> torch.Tensor:
        """the factor the loss is multiplied by before backward (GradScaler.scale); a device scalar"""
        return self.state[0]

    def step(self, lr: float = None, zero_grads: bool = True) -> torch.Tensor:
        """one optimizer update (``update_grad=True``); ``lr`` = this iteration's scheduler value.  Returns the total gradient
        norm (after un-scaling, before clipping) as a device scalar, like the reference's ``grad_norm``."""
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(lib.esam3_stage1_update(
                self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.layout.n,
                self._scale.data_ptr(), self._decay.data_ptr(), float(self.lr if lr is None else lr), float(self.betas[0]),
                float(self.betas[1]), self.eps, self.weight_decay, float(self.clip_grad or 0.0), self.state.data_ptr(),
                float(self.growth_factor), float(self.backoff_factor), int(self.growth_interval), int(self.amp), int(zero_grads),
                self.bf16.data_ptr() if self.bf16 is not None else None, self._ws.data_ptr(),
                torch.cuda.current_stream().cuda_stream), "esam3_stage1_update")
        return self.state[3].clone()   # a copy: state[3] is overwritten by the next step

    # NativeScalerWithGradNormCount.state_dict is GradScaler's: {"scale", "growth_factor", "backoff_factor",
    # "growth_interval", "_growth_tracker"}; the optimizer's adds the step count and the two moments
    def state_dict(self) -> dict:
        st = self.state.cpu()
        return {"amp_scaler": {"scale": float(st[0]), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                               "growth_interval": self.growth_interval, "_growth_tracker": int(st[1])},
                "optimizer": {"step": int(st[4]), "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu()}}

    def load_state_dict(self, sd: dict) -> None:
        a, o = sd["amp_scaler"], sd["optimizer"]
        self.growth_factor, self.backoff_factor, self.growth_interval = a["growth_factor"], a["backoff_factor"], a["growth_interval"]
        st = torch.zeros(16, dtype=torch.float32)
        st[0], st[1], st[4] = a["scale"], a["_growth_tracker"], o["step"]
        self.state.copy_(st)
        self.exp_avg.copy_(o["exp_avg"])
        self.exp_avg_sq.copy_(o["exp_avg_sq"])


# ---- BatchNorm2d in training mode (building blocks of the trunk backward; the engine's inference path folds BN into its convs) --------
def _bn_rows(x: torch.Tensor):
    assert x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() >= 2 and x.is_contiguous()
    c = x.shape[-1]
    return x.numel() // c, c


def bn_train_forward(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: torch.Tensor = None,
                     running_var: torch.Tensor = None, momentum: float = 0.1, eps: float = 1e-5):
    """nn.BatchNorm2d.forward in training mode on an NHWC tensor [..., C] (backbones/efficientvit/nn/ops.py:69-77): returns
    (y, save_mean, save_rstd); the running statistics are updated in place like the module's buffers."""
    rows, c = _bn_rows(x)
    lib = _lib.load()
    y = torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.esam3_bn_train_workspace(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_bn_train_forward(_DT[x.dtype], x.data_ptr(), y.data_ptr(), rows, c, gamma.data_ptr(), beta.data_ptr(),
                                              None if running_mean is None else running_mean.data_ptr(),
                                              None if running_var is None else running_var.data_ptr(), float(momentum), float(eps),
                                              mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "esam3_bn_train_forward")
    return y, mean, rstd


def bn_train_backward(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, save_mean: torch.Tensor, save_rstd: torch.Tensor):
    """the autograd backward of the above: (dx, dgamma, dbeta)"""
    rows, c = _bn_rows(x)
    assert dy.shape == x.shape and dy.dtype == x.dtype and dy.is_contiguous()
    lib = _lib.load()
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.esam3_bn_train_workspace(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_bn_train_backward(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, c, gamma.data_ptr(),
                                               save_mean.data_ptr(), save_rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                               ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "esam3_bn_train_backward")
    return dx, dgamma, dbeta


# ---- the halves of the two calls above, for SyncBatchNorm (train_blocks.bn_train_forward / bn_train_backward) -----------------------------------
def bn_stats(x: torch.Tensor, eps: float = 1e-5):
    """this rank's (mean, rstd, biased variance) per channel of an NHWC tensor [..., C] (``esam3_bn_train_stats``)"""
    rows, c = _bn_rows(x)
    lib = _lib.load()
    mean, rstd, var = (torch.empty(c, dtype=torch.float32, device=x.device) for _ in range(3))
    ws = torch.empty(int(lib.esam3_bn_train_workspace(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_bn_train_stats(_DT[x.dtype], x.data_ptr(), rows, c, float(eps), mean.data_ptr(), rstd.data_ptr(), var.data_ptr(), ws.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "esam3_bn_train_stats")
    return mean, rstd, var


def bn_apply(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor) -> torch.Tensor:
    """y = (x - mean) rstd gamma + beta with given statistics (``esam3_bn_train_apply``)"""
    rows, c = _bn_rows(x)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().esam3_bn_train_apply(_DT[x.dtype], x.data_ptr(), y.data_ptr(), rows, c, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                                    rstd.data_ptr(), torch.cuda.current_stream().cuda_stream), "esam3_bn_train_apply")
    return y


def bn_backward_sums(x: torch.Tensor, dy: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor):
    """this rank's (sum dy xhat, sum dy) per channel (``esam3_bn_train_backward_sums``): its dgamma and dbeta"""
    rows, c = _bn_rows(x)
    assert dy.shape == x.shape and dy.dtype == x.dtype and dy.is_contiguous()
    lib = _lib.load()
    sdyx, sdy = (torch.empty(c, dtype=torch.float32, device=x.device) for _ in range(2))
    ws = torch.empty(int(lib.esam3_bn_train_workspace(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_bn_train_backward_sums(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), rows, c, mean.data_ptr(), rstd.data_ptr(), sdyx.data_ptr(),
                                                    sdy.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "esam3_bn_train_backward_sums")
    return sdyx, sdy


def bn_backward_apply(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, mean_dy_xhat: torch.Tensor,
                      mean_dy: torch.Tensor) -> torch.Tensor:
    """dx = gamma rstd (dy - mean_dy - xhat mean_dy_xhat) with the all-rank sums ALREADY divided by the all-rank row count
    (``esam3_bn_train_backward_apply`` with total_rows = 1)"""
    rows, c = _bn_rows(x)
    dx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().esam3_bn_train_backward_apply(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, c, gamma.data_ptr(), mean.data_ptr(),
                                                             rstd.data_ptr(), mean_dy_xhat.data_ptr(), mean_dy.data_ptr(), 1.0,
                                                             torch.cuda.current_stream().cuda_stream), "esam3_bn_train_backward_apply")
    return dx


_ACT_CODE = {"relu": 1, "gelu": 2, "hswish": 3, "sigmoid": 4}


def bn_act_train_forward(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: torch.Tensor, running_var: torch.Tensor,
                         momentum: float, eps: float, act: str, keep_pre: bool = True):
    """``bn_train_forward`` and the ConvLayer's activation in the same passes -> (y, act(y), save_mean, save_rstd) (``esam3_bn_act_train_forward``);
    ``keep_pre`` False: y is neither written nor returned (None) -- for ``bn_act_train_backward`` with ``pre=None``, which recomputes it"""
    rows, c = _bn_rows(x)
    lib = _lib.load()
    y, y_act = (torch.empty_like(x) if keep_pre else None), torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.esam3_bn_train_workspace(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.esam3_bn_act_train_forward(_DT[x.dtype], x.data_ptr(), None if y is None else y.data_ptr(), y_act.data_ptr(), _ACT_CODE[act], rows, c, gamma.data_ptr(),
                                                  beta.data_ptr(), None if running_mean is None else running_mean.data_ptr(),
                                                  None if running_var is None else running_var.data_ptr(), float(momentum), float(eps),
                                                  mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "esam3_bn_act_train_forward")
    return y, y_act, mean, rstd


def bn_act_train_backward(x: torch.Tensor, dy: torch.Tensor, pre: torch.Tensor, act: str, gamma: torch.Tensor, save_mean: torch.Tensor,
                          save_rstd: torch.Tensor, beta: torch.Tensor = None):
    """backward of act(batch_norm(x)): dy = the gradient of the activation's OUTPUT, pre = the BatchNorm's output -> (dx, dgamma, dbeta)
    (``esam3_bn_act_train_backward``: dy act'(pre) is formed inside the two BatchNorm passes).  ``pre`` None with ``beta``: the BatchNorm's
    output is recomputed from x as the forward stored it (``esam3_bn_act_train_backward_rc``)"""
    rows, c = _bn_rows(x)
    assert dy.shape == x.shape and dy.dtype == x.dtype and dy.is_contiguous()
    assert (pre is None and beta is not None) or (pre.shape == x.shape and pre.dtype == x.dtype and pre.is_contiguous())
    lib = _lib.load()
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.esam3_bn_train_workspace(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        if pre is None:
            _lib.check(lib.esam3_bn_act_train_backward_rc(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), _ACT_CODE[act], dx.data_ptr(), rows, c, gamma.data_ptr(),
                                                          beta.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                          ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "esam3_bn_act_train_backward_rc")
            return dx, dgamma, dbeta
        _lib.check(lib.esam3_bn_act_train_backward(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), pre.data_ptr(), _ACT_CODE[act], dx.data_ptr(), rows, c,
                                                   gamma.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                   ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "esam3_bn_act_train_backward")
    return dx, dgamma, dbeta
