"""Host-side mirror of the reference's image model objects for the hot path.

Same names, argument meaning and error behaviour as the reference so callers
(``eval/eval_coco.py``, the example scripts) run unchanged:

  * ``Sam3Image.predict_inst`` / ``predict_inst_batch``  <- sam3/sam3/model/sam3_image.py:599-684
  * prompt preparation / output conventions              <- sam3/sam3/model/sam1_task_predictor.py:168-430
  * coordinate transforms                                <- sam3/sam3/model/utils/sam1_utils.py:47-75
  * ``backbone.forward_image`` result dictionary         <- sam3/sam3/model/vl_combiner.py:81-124

All tensor math happens in libesam3_hip.so (``HipEngine``); this file only shuffles
prompts (a handful of floats, on the host) and wraps device buffers.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import os

import sys

import numpy as np
import torch

from . import _lib, schema
from .engine import EMB, LOW_RES, NET_RES, HipEngine


def _sine_position_encoding(h: int, w: int, num_pos_feats: int = 256, temperature: float = 10000.0) -> np.ndarray:
    """PositionEmbeddingSine (normalize=True, scale=2*pi) -- a constant per (h, w)
    (sam3/sam3/model/position_encoding.py:92-127); fp32 numpy, [C,H,W]."""
    half = num_pos_feats // 2
    y = np.arange(1, h + 1, dtype=np.float32)[:, None].repeat(w, 1)
    x = np.arange(1, w + 1, dtype=np.float32)[None, :].repeat(h, 0)
    eps = np.float32(1e-6)
    scale = np.float32(2 * math.pi)
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = np.arange(half, dtype=np.float32)
    dim_t = (np.float32(temperature) ** (2 * (dim_t // 2) / np.float32(half))).astype(np.float32)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = np.stack((np.sin(px[:, :, 0::2]), np.cos(px[:, :, 1::2])), axis=3).reshape(h, w, -1)
    py = np.stack((np.sin(py[:, :, 0::2]), np.cos(py[:, :, 1::2])), axis=3).reshape(h, w, -1)
    return np.ascontiguousarray(np.concatenate((py, px), axis=2).transpose(2, 0, 1)).astype(np.float32)


def _nchw_view(t_nhwc: torch.Tensor) -> torch.Tensor:
    """Logical NCHW view (channels-last memory) of an NHWC buffer: zero copy."""
    return t_nhwc.permute(0, 3, 1, 2)



_WIDEN_POOL = None


def _widen_into(dst: np.ndarray, src: np.ndarray, parts: int = 0, serial_below: int = 1 << 22) -> None:
    """dst[...] = src (same shape, any dtypes numpy can cast: the uint8 masks become the float32 array the reference's contract returns);
    large arrays are split into `parts` contiguous ranges copied by the worker threads.  uint8 -> float32 (the thresholded masks) runs
    in the library's C loop (esam3_host_widen_u8_f32: numpy's casting copy reached 37 GB/s of stores on 16 threads, 3.6 ms per 32
    masks of 1024 x 1024 -- the longest host-side piece of an API-level step, profiles/r04/api_level_probe.txt)."""
    assert dst.shape == src.shape and dst.flags.c_contiguous and src.flags.c_contiguous
    d, s_ = dst.reshape(-1), src.reshape(-1)
    n = s_.size
    native = None
    if src.dtype == np.uint8 and dst.dtype == np.float32:
        fn = _lib.load().esam3_host_widen_u8_f32
        sp, dp = s_.ctypes.data, d.ctypes.data
        native = lambda a, b: fn(sp + a, dp + 4 * a, b - a)  # noqa: E731
    if n < serial_below:
        if native is not None:
            native(0, n)
        else:
            np.copyto(d, s_, casting="unsafe")
        return
    parts = parts or host_threads()
    step = -(-n // parts)
    if native is not None:
        list(_widen_pool().map(lambda i: native(i * step, min(n, (i + 1) * step)), range(parts)))
        return
    list(_widen_pool().map(lambda i: np.copyto(d[i * step:(i + 1) * step], s_[i * step:(i + 1) * step], casting="unsafe"), range(parts)))


def _pool_get(pool: list, make):
    """One result buffer (tensor, its ndarray) out of `pool`: a buffer that was handed out before comes back ONLY if no view of its
    ndarray is alive outside the pool (the tuple's slot + getrefcount's argument = 2); otherwise `make()` builds a new one.  At most
    three are kept per shape: an `out = step()` loop needs two, a consumer one step behind three."""
    for cand in pool:
        if sys.getrefcount(cand[1]) <= 2:
            return cand
    t = make()
    ent = (t, t.numpy())                         # views handed to the caller keep the ndarray (their base) alive
    pool.append(ent)
    if len(pool) > 3:
        pool.pop(0)
    return ent


def host_threads() -> int:
    """worker threads of the host-side copies (PIL staging, mask widening): ESAM3_HOST_THREADS, else min(16, cores) -- measured on the
    256-core GPU host (profiles/r06/api_host_threads.txt)"""
    try:
        n = int(os.environ.get("ESAM3_HOST_THREADS", "0"))
    except ValueError:
        n = 0
    return n if n > 0 else min(16, os.cpu_count() or 1)


def _widen_pool():
    """a few long-lived worker threads for host-side copies (numpy releases the GIL inside them)"""
    global _WIDEN_POOL
    if _WIDEN_POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _WIDEN_POOL = ThreadPoolExecutor(max_workers=host_threads(), thread_name_prefix="esam3-widen")
    return _WIDEN_POOL


class _Trunk:
    """``model.backbone.vision_backbone.trunk(x) -> [Tensor[B,1024,72,72]]`` (stage1/model.py:237)."""

    def __init__(self, owner: "Sam3Image"):
        self._o = owner
        self.channel_list = [1024]

    def __call__(self, x):
        x = x[0] if isinstance(x, list) else x
        out = self._o.engine.encode(self._o._to_input(x), want_sam3=False, want_sam2=False, want_trunk=True)
        return [_nchw_view(out["trunk"])]


class _VisionBackbone:
    def __init__(self, owner: "Sam3Image"):
        self.trunk = _Trunk(owner)


class _TextStudentEncoder:
    """Stand-in for TextStudentEncoder (text_encoder_student.py:9-58): tokenizer on the host, the
    MobileCLIP student transformer (S0 "mct", or the 12-layer "base" students) on the HIP engine.
    ``__call__(text, input_boxes, device)`` -> (mask [B,S] bool True = padding, memory [S,B,256],
    embeds [S,B,dim])."""

    def __init__(self, owner: "Sam3Image", context_length: int, bpe_path=None):
        self._o = owner
        self.context_length = context_length
        self._bpe_path = bpe_path if bpe_path is not None else os.environ.get("ESAM3_BPE_PATH")
        self._tokenizer = None

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .tokenizer import ClipBpeTokenizer
            self._tokenizer = ClipBpeTokenizer(self._bpe_path, context_length=self.context_length)
        return self._tokenizer

    def set_context_length(self, context_length: int):
        """The positional table is sliced per call; only lengths up to the loaded table are valid."""
        self.context_length = context_length

    def encode_tokens(self, tokenized: torch.Tensor):
        mem, emb = self._o.engine.encode_text(tokenized, dim=schema.TEXT_ENCODER_CFG[self._o.text_encoder_type][0])
        return (tokenized.to(mem.device) == 0), mem, emb

    def __call__(self, text, input_boxes=None, device=None):
        tokenized = torch.from_numpy(self.tokenizer(text, context_length=self.context_length))
        return self.encode_tokens(tokenized)


class _VLBackbone:
    """Stand-in for SAM3VLBackbone (vl_combiner.py:20-180)."""

    def __init__(self, owner: "Sam3Image"):
        self._o = owner
        self.vision_backbone = _VisionBackbone(owner)
        self.language_backbone = None
        self.scalp = 1

    def forward_image(self, samples: torch.Tensor) -> dict:
        return self._o._forward_image(samples)

    def forward_text(self, captions, input_boxes=None, additional_text=None, device=None):
        """_forward_text_no_ack_ckpt (vl_combiner.py:136-180)."""
        if self.language_backbone is None:
            raise NotImplementedError("model was built without a text encoder (pass text_encoder_type=, e.g. 'MobileCLIP-S0')")
        texts = list(captions) + (list(additional_text) if additional_text is not None else [])
        mask, memory, embeds = self.language_backbone(texts, input_boxes, device=device)
        out = {}
        if additional_text is not None:
            out["additional_text_features"] = memory[:, -len(additional_text):]
            out["additional_text_mask"] = mask[-len(additional_text):]
        n = len(captions)
        out["language_features"] = memory[:, :n]
        out["language_mask"] = mask[:n]
        out["language_embeds"] = embeds[:, :n]
        return out


class _InteractivePredictorInfo:
    """Marker object: ``model.inst_interactive_predictor is not None`` <=> interactivity enabled."""

    mask_threshold = 0.0
    max_hole_area = 256.0
    max_sprinkle_area = 0.0
    _bb_feat_sizes = [(288, 288), (144, 144), (72, 72)]


class Sam3Image:
    """EfficientSAM3 image model whose forward passes run on the HIP engine."""

    def __init__(self, backbone_type: str, model_name: str, enable_inst_interactivity: bool,
                 dtype: str = "bf16", device=None, dual_neck: bool = True,
                 fuse_linear_chains: bool = True, text_encoder_type: Optional[str] = None,
                 text_encoder_context_length: int = 77, bpe_path=None):
        self.backbone_type = backbone_type
        self.model_name = model_name
        self.dual_neck = dual_neck
        self.engine = HipEngine(backbone_type, model_name, dtype=dtype, device=device,
                                interactive=enable_inst_interactivity,
                                fuse_linear_chains=fuse_linear_chains)
        self.device = self.engine.device
        self.backbone = _VLBackbone(self)
        self.inst_interactive_predictor = _InteractivePredictorInfo() if enable_inst_interactivity else None
        self._schema = schema.image_path_schema(backbone_type, model_name, enable_inst_interactivity)
        self.text_encoder_type = text_encoder_type
        if text_encoder_type is not None:
            if text_encoder_type not in schema.TEXT_ENCODER_CFG:
                raise NotImplementedError(f"text_encoder_type={text_encoder_type!r}: known students are "
                                          f"{sorted(schema.TEXT_ENCODER_CFG)}")
            self.engine.set_text_causal(schema.TEXT_ENCODER_CFG[text_encoder_type][4])
            self.backbone.language_backbone = _TextStudentEncoder(self, text_encoder_context_length, bpe_path)
            self._schema.update(schema.pcs_schema())  # the grounding detector the text prompts feed
        self._sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._pos_cache: Dict[Tuple[int, int], torch.Tensor] = {}
        self._copy_stream = None                           # side stream of the chunked device-to-host mask copies
        self._host_stage: Dict[tuple, torch.Tensor] = {}   # pinned D2H staging buffers of predict_inst_batch, by (shape, dtype)
        self._host_out: Dict[tuple, list] = {}             # per shape: up to three result buffers (tensor, ndarray), each handed out again only once the caller dropped it
        self.training = False

    # ---- nn.Module-like surface -------------------------------------------------------------
    def eval(self):
        return self

    def to(self, device=None, *a, **k):
        if device is not None and torch.device(device).type != "cuda":
            raise RuntimeError("this model lives on a HIP device; there is no CPU path")
        return self

    def cuda(self, *a, **k):
        return self

    def parameters(self):
        return iter(self._sd.values())

    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """Reference key names (Appendix C of SURVEY.md).  Keys outside the hot-path schema are
        ignored like ``strict=False`` does upstream (model_builder.py:584-630)."""
        if self.engine.finalized:
            raise RuntimeError("weights are already packed on the device; build a new model to reload")
        if self.text_encoder_type is not None:
            # checkpoints carry the 77-row positional table (the reference truncates it after loading,
            # model_builder.py:1035-1047); take whatever length the state dict holds
            pk = schema.TEXT + "encoder.positional_embedding.pos_embed.pos_embed"
            rows = int(sd[pk].shape[2]) if pk in sd else self.backbone.language_backbone.context_length
            self._schema.update(schema.text_encoder_schema(self.text_encoder_type, rows))
        if self.text_encoder_type is not None:
            # a checkpoint may carry the text student without the grounding detector (stage-1 text checkpoints):
            # the text path still works, forward_grounding then fails loudly
            pcs_keys = [k for k in schema.pcs_schema() if k in self._schema]
            self._has_detector = all(k in sd for k in pcs_keys)
            if not self._has_detector:
                for k in pcs_keys:
                    del self._schema[k]
        missing = [k for k, (shape, kind) in self._schema.items() if k not in sd and kind != "bn_n"]
        unexpected = [k for k in sd if k not in self._schema]
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing keys: {missing[:5]} unexpected: {unexpected[:5]}")
        if missing:
            raise KeyError(f"checkpoint lacks {len(missing)} tensors the hot path needs, e.g. {missing[:3]}")
        for k, (shape, kind) in self._schema.items():
            if kind == "bn_n":
                continue
            t = sd[k]
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{k}: shape {tuple(t.shape)} != expected {tuple(shape)}")
            self._sd[k] = t.detach().to("cpu", torch.float32)
        self.engine.load_state_dict(self._sd)
        self.engine.finalize()
        return missing, unexpected

    # ---- image encoder ------------------------------------------------------------------------
    def _to_input(self, samples: torch.Tensor) -> torch.Tensor:
        if samples.dim() == 3:
            samples = samples[None]
        return samples.to(self.device, torch.float32)

    def _pos(self, b: int, h: int, w: int) -> torch.Tensor:
        key = (h, w)
        if key not in self._pos_cache:
            self._pos_cache[key] = torch.from_numpy(_sine_position_encoding(h, w)).to(
                self.device, self.engine.torch_dtype)
        return self._pos_cache[key][None].expand(b, -1, -1, -1)

    def _forward_image(self, samples: torch.Tensor) -> dict:
        """Same dictionary as SAM3VLBackbone.forward_image (vl_combiner.py:106-124); the sam2
        levels 0/1 already carry the conv_s0/conv_s1 projection that Sam3Processor applies
        right after (sam3_image_processor.py:62-75) -- flagged by ``_esam3_projected``."""
        x = self._to_input(samples)
        interactive = self.inst_interactive_predictor is not None
        out = self.engine.encode(x, want_sam3=self.dual_neck, want_sam2=interactive)
        return self._features_dict(out, x.shape[0])

    def _features_dict(self, out: dict, b: int) -> dict:
        """engine.encode's NHWC buffers -> the dictionary of SAM3VLBackbone.forward_image (NCHW views, position encodings)"""
        interactive = self.inst_interactive_predictor is not None
        res = {"vision_features": None, "vision_pos_enc": None, "backbone_fpn": None,
               "sam2_backbone_out": None}
        if self.dual_neck:
            fpn = [_nchw_view(t) for t in out["sam3_fpn"]]
            res.update(vision_features=fpn[-1], backbone_fpn=fpn,
                       vision_pos_enc=[self._pos(b, t.shape[-2], t.shape[-1]) for t in fpn],
                       _esam3_nhwc_sam3=out["sam3_fpn"])
        if interactive:
            fpn2 = [_nchw_view(t) for t in out["sam2_fpn"]]
            res["sam2_backbone_out"] = {
                "vision_features": fpn2[-1], "backbone_fpn": fpn2,
                "vision_pos_enc": [self._pos(b, t.shape[-2], t.shape[-1]) for t in fpn2],
                "_esam3_projected": True, "_esam3_nhwc": out["sam2_fpn"]}
        return res

    # ---- prompt handling (host side, a handful of floats) -------------------------------------
    @staticmethod
    def _prep_prompts(point_coords, point_labels, box, normalize_coords, orig_hw):
        """-> (coords [Bp,Np,2] f32 network px, labels [Bp,Np] i32) or (None, None).
        sam1_task_predictor.py:298-326,385-396 + sam1_utils.py:47-75."""
        h, w = orig_hw
        coords = labels = None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            coords = np.array(point_coords, dtype=np.float32, copy=True)
            if normalize_coords:
                coords[..., 0] = coords[..., 0] / np.float32(w)
                coords[..., 1] = coords[..., 1] / np.float32(h)
            coords = coords * np.float32(NET_RES)
            labels = np.asarray(point_labels).astype(np.int32)
            if coords.ndim == 2:
                coords, labels = coords[None], labels[None]
        if box is not None:
            bx = np.array(box, dtype=np.float32, copy=True).reshape(-1, 2, 2)
            if normalize_coords:
                bx[..., 0] = bx[..., 0] / np.float32(w)
                bx[..., 1] = bx[..., 1] / np.float32(h)
            bx = bx * np.float32(NET_RES)
            bl = np.tile(np.array([[2, 3]], dtype=np.int32), (bx.shape[0], 1))
            if coords is not None:
                coords = np.concatenate([bx, coords], axis=1)
                labels = np.concatenate([bl, labels], axis=1)
            else:
                coords, labels = bx, bl
        return coords, labels

    def _decode(self, sam2_nhwc: Sequence[torch.Tensor], coords: Optional[np.ndarray], labels: Optional[np.ndarray],
                prompt_image: np.ndarray, multimask_output: bool, mask_input: Optional[np.ndarray] = None):
        dev = self.device
        c = l = m = None
        if coords is not None:
            c = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.float32)).to(dev)
            l = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).to(dev)
        if mask_input is not None:
            m = torch.from_numpy(np.ascontiguousarray(mask_input, dtype=np.float32)).to(dev)
        pi = torch.from_numpy(np.ascontiguousarray(prompt_image, dtype=np.int32)).to(dev)
        return self.engine.decode(sam2_nhwc, pi, c, l, multimask_output, mask_input=m)

    @staticmethod
    def _prep_mask_input(mask_input, bp: Optional[int]) -> Tuple[Optional[np.ndarray], int]:
        """mask_input [1,288,288] (or [Bp,1,288,288], sam1_task_predictor.py:327-333) -> [Bp,288,288].
        Without point/box prompts the batch is the mask's own (1 for the documented [1,H,W] form)."""
        if mask_input is None:
            return None, (1 if bp is None else bp)
        m = np.asarray(mask_input, dtype=np.float32)
        if m.ndim == 3:
            m = m[None]
        assert m.ndim == 4 and m.shape[1] == 1 and m.shape[-2:] == (LOW_RES, LOW_RES), \
            f"mask_input must be [1,{LOW_RES},{LOW_RES}] low-res logits, got {m.shape}"
        m = m[:, 0]
        if bp is None:
            bp = m.shape[0]
        if m.shape[0] == 1 and bp > 1:  # one mask shared by all prompt sets (broadcast in mask_decoder.py:196-197)
            m = np.repeat(m, bp, axis=0)
        assert m.shape[0] == bp, f"mask_input batch {m.shape[0]} != number of prompt sets {bp}"
        return np.ascontiguousarray(m), bp

    def _check_state(self, inference_state):
        if self.inst_interactive_predictor is None:
            raise RuntimeError("model was built with enable_inst_interactivity=False")
        bo = inference_state["backbone_out"]["sam2_backbone_out"]
        if bo is None or "_esam3_nhwc" not in bo:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        return bo["_esam3_nhwc"]

    def predict_inst(self, inference_state, point_coords=None, point_labels=None, box=None,
                     mask_input=None, multimask_output: bool = True, return_logits: bool = False,
                     normalize_coords: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Same contract as the reference (sam3_image.py:599-636): returns numpy
        (masks [C,H,W] float32 0/1 or logits, iou [C], low_res [C,288,288]); K boxes keep a
        leading K dimension."""
        sam2 = self._check_state(inference_state)
        h, w = inference_state["original_height"], inference_state["original_width"]
        coords, labels = self._prep_prompts(point_coords, point_labels, box, normalize_coords, (h, w))
        mask, bp = self._prep_mask_input(mask_input, None if coords is None else coords.shape[0])
        low, iou = self._decode(sam2, coords, labels, np.zeros((bp,), np.int32), multimask_output, mask)
        masks = self.engine.postprocess(low, (h, w), return_logits)
        self.engine.clamp_(low, -32.0, 32.0)
        masks_np = masks.squeeze(0).float().cpu().numpy()
        return masks_np, iou.squeeze(0).cpu().numpy(), low.squeeze(0).cpu().numpy()

    def _d2h_begin(self, t: torch.Tensor, slot: int = 0):
        """Start the device-to-host hand-back of one result tensor on the copy stream and return a handle for _d2h_end.

        uint8 (thresholded masks; the reference's contract returns them as float32, sam1_task_predictor.py:293-295): the bytes leave
        the device as they are (a quarter of the float32 size) into a pinned staging buffer and are widened on the host in _d2h_end.
        float32 (mask logits, low-res logits): copied straight into a PINNED result buffer whose numpy view is what the caller gets --
        no host-side copy at all (round 5 staged them and copied again).  Result buffers come from _pool_get: reused only when
        nothing outside this object references them."""
        if not t.is_cuda:  # an engine double on the host (tests): nothing to stage
            return (None, None, t.float().numpy(), None)
        t = t.contiguous()
        widen = t.dtype != torch.float32
        key = (tuple(t.shape), t.dtype)
        pin = None
        if widen:   # `slot`: hand-backs in flight at the same time never share a staging buffer
            pin = self._host_stage.get(key + (slot,))
            if pin is None:
                if len(self._host_stage) >= 6:
                    self._host_stage.clear()
                pin = self._host_stage[key + (slot,)] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        # The result array is the caller's (the reference returns fresh arrays), but 134 MB of never-touched pages cost ~30 ms of page
        # faults per call: a previous result buffer of this shape is handed out again ONLY if nothing outside this object references
        # it any more (a caller that kept its arrays keeps them untouched).
        ent = _pool_get(self._host_out.setdefault(key, []),
                        lambda: torch.empty(t.shape, dtype=torch.float32, pin_memory=not widen))
        cur = torch.cuda.current_stream(t.device)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=t.device)
        self._copy_stream.wait_stream(cur)
        with torch.cuda.stream(self._copy_stream):
            (pin if widen else ent[0]).copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return (ev, pin, ent[1], t)

    def _d2h_end(self, handle) -> np.ndarray:
        """Wait for one hand-back started by _d2h_begin; uint8 masks are widened to float32 here, on a few worker threads.  NOT torch's
        copy_: its OpenMP team (128 threads on the GPU host) keeps spinning after the copy, and the NEXT wait on the device then returned
        50-70 ms late in about every third step (tools/api_stall_probe.py, profiles/r04/api_stall_probe.txt)."""
        ev, pin, out, src = handle
        if ev is None:
            return out
        ev.synchronize()
        if pin is not None:
            _widen_into(out, pin.numpy())
        # `src` may be freed / reused by the caller's stream only after the copy has run
        torch.cuda.current_stream(src.device).wait_stream(self._copy_stream)
        return out

    def _masks_to_host(self, masks: torch.Tensor) -> np.ndarray:
        """Device masks of one size group -> the float32 numpy array the reference's contract returns, start to finish."""
        return self._d2h_end(self._d2h_begin(masks))

    def predict_inst_batch(self, inference_state, point_coords_batch=None, point_labels_batch=None,
                           box_batch=None, mask_input_batch=None, multimask_output: bool = True,
                           return_logits: bool = False, normalize_coords: bool = True):
        """sam3_image.py:638-684 / sam1_task_predictor.py:168-228, but all images that share a
        prompt layout are decoded in ONE engine call instead of a Python loop with a D2H sync per
        image.  Returns three lists (masks, ious, low_res), one entry per image."""
        sam2 = self._check_state(inference_state)
        hs, ws = inference_state["original_heights"], inference_state["original_widths"]
        n_img = sam2[2].shape[0]
        assert n_img == len(hs) == len(ws), \
            f"Batch size mismatch in predict_inst_batch. Got {n_img}, {len(hs)}, {len(ws)}"
        per = []
        for i in range(n_img):
            pc = point_coords_batch[i] if point_coords_batch is not None else None
            pl = point_labels_batch[i] if point_labels_batch is not None else None
            bx = box_batch[i] if box_batch is not None else None
            c, l = self._prep_prompts(pc, pl, bx, normalize_coords, (hs[i], ws[i]))
            mi = mask_input_batch[i] if mask_input_batch is not None else None
            m, bpi = self._prep_mask_input(mi, None if c is None else c.shape[0])
            per.append((c, l, m, bpi))
        # group images by prompt layout (Bp_i, Np, has mask)
        groups: Dict[Tuple[int, int, bool], List[int]] = {}
        for i, (c, _, m, bpi) in enumerate(per):
            groups.setdefault((bpi, 0 if c is None else c.shape[1], m is not None), []).append(i)
        masks_out: List[Optional[np.ndarray]] = [None] * n_img
        iou_out: List[Optional[np.ndarray]] = [None] * n_img
        low_out: List[Optional[np.ndarray]] = [None] * n_img
        for (bpi, npts, has_mask), idxs in groups.items():
            coords = np.concatenate([per[i][0] for i in idxs], axis=0) if npts else None
            labels = np.concatenate([per[i][1] for i in idxs], axis=0) if npts else None
            mask = np.concatenate([per[i][2] for i in idxs], axis=0) if has_mask else None
            pimg = np.repeat(np.asarray(idxs, dtype=np.int32), bpi)
            low, iou = self._decode(sam2, coords, labels, pimg, multimask_output, mask)
            # post-process per distinct original size
            by_size: Dict[Tuple[int, int], List[int]] = {}
            for j, i in enumerate(idxs):
                by_size.setdefault((hs[i], ws[i]), []).append(j)
            low_g = low.view(len(idxs), bpi, *low.shape[1:])
            iou_g = iou.view(len(idxs), bpi, -1)
            # Every hand-back is STARTED before any of them is waited for (round 5 widened the masks on the host before the clamp and
            # the low-res copy were even launched: 1.1 ms of idle device per step, profiles/r06/api_timeline_before.txt): the masks of
            # each size group, then the clamped low-res logits and the scores travel on the copy stream while the host widens.
            pending = []
            for (h, w), js in by_size.items():
                sel = low_g[js] if len(js) != len(idxs) else low_g
                pending.append((js, self._d2h_begin(self.engine.postprocess(sel.contiguous(), (h, w), return_logits), slot=len(pending))))
            self.engine.clamp_(low, -32.0, 32.0)
            big = low_g.is_cuda and low_g.numel() >= (1 << 20)   # 10 MB at 32 prompts: a pageable copy of that size costs 1-2 ms
            low_h = self._d2h_begin(low_g) if big else None
            iou_h = self._d2h_begin(iou_g) if big else None
            for js, hnd in pending:
                m = self._d2h_end(hnd)
                for k, j in enumerate(js):
                    masks_out[idxs[j]] = m[k].squeeze(0) if bpi == 1 else m[k]
            low_np = self._d2h_end(low_h) if big else low_g.cpu().numpy()
            iou_np = self._d2h_end(iou_h) if big else iou_g.cpu().numpy()
            for j, i in enumerate(idxs):
                low_out[i] = low_np[j].squeeze(0) if bpi == 1 else low_np[j]
                iou_out[i] = iou_np[j].squeeze(0) if bpi == 1 else iou_np[j]
        return masks_out, iou_out, low_out

    # ---- PCS text grounding (sam3_image.py:442-493) -------------------------------------------------
    def forward_grounding(self, backbone_out, find_input=None, find_target=None, geometric_prompt=None):
        """Sam3Image.forward_grounding for what Sam3Processor.set_text_prompt / add_geometric_prompt /
        add_point_prompt pass: the image features of set_image, the text features of forward_text (one text,
        broadcast to every image) and the geometric prompt (a ``geometry_prompt.Prompt``; empty = dummy)."""
        if self.text_encoder_type is None:
            raise NotImplementedError("model was built without a text encoder (pass text_encoder_type=, e.g. 'MobileCLIP-S0')")
        if not getattr(self, "_has_detector", False):
            raise RuntimeError("the loaded state dict has no grounding-detector weights (geometry_encoder.*, "
                               "transformer.*, segmentation_head.*, dot_prod_scoring.*)")
        if "language_features" not in backbone_out:
            raise ValueError("forward_text has not been run for this state")
        fpn = backbone_out.get("_esam3_nhwc_sam3")
        if fpn is None:
            raise RuntimeError("the sam3 neck features are missing (model built with dual_neck=False?)")
        b = fpn[2].shape[0]
        lf, lm = backbone_out["language_features"], backbone_out["language_mask"]
        if lf.shape[1] == 1 and b > 1:
            lf, lm = lf.expand(-1, b, -1), lm.expand(b, -1)
        geo = None
        if geometric_prompt is not None and geometric_prompt.n_prompts > 0:
            geo = geometric_prompt.batch_first()
            if geo["points"].shape[0] == 1 and b > 1:
                geo = {k: v.expand(b, *v.shape[1:]) for k, v in geo.items()}
        return self.engine.ground(fpn, lf, lm, geo=geo)

    def _get_dummy_prompt(self, num_prompts: int = 1):
        """The empty geometric Prompt (sam3_image.py:522-528): no boxes, no points."""
        from .geometry_prompt import Prompt
        return Prompt(box_embeddings=torch.zeros(0, num_prompts, 4, device=self.device),
                      box_mask=torch.zeros(num_prompts, 0, dtype=torch.bool, device=self.device))
