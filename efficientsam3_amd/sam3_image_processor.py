"""``Sam3Processor`` with the reference's interface
(sam3/sam3/model/sam3_image_processor.py:14-113).  Pre-processing (uint8 -> normalised
NCHW fp32) is a HIP kernel; the encoder is the HIP engine."""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List

import numpy as np
import torch

from .engine import NET_RES
from .sam3_image import host_threads

try:  # PIL is optional at import time
    import PIL.Image as _PILImage
except Exception:  # pragma: no cover
    _PILImage = None


class Sam3Processor:
    def __init__(self, model, resolution=NET_RES, device=None, confidence_threshold=0.5):
        if resolution != NET_RES:
            raise ValueError(f"the network resolution is fixed at {NET_RES}")
        self.model = model
        self.resolution = resolution
        self.device = model.device if device is None else torch.device(device)
        self.confidence_threshold = confidence_threshold
        self._stage = {}   # (B, H, W) -> pinned uint8 [B, H, W, 3] staging buffer of set_image_batch, allocated once
        self._pool = None  # worker threads that convert PIL images into the staging buffer
        # A/B switch: stage Pillow's 4-byte pixels with torch's copy_ instead of np.copyto from the pool.  Measured on the GPU host
        # (256 cores): both stage 32 images in 1.7-1.8 ms, but torch's 128 spinning OpenMP threads then starve the HIP runtime's
        # own threads -- the step went from 40 to 58 ms (profiles/r04/api_level_probe.txt).  Off.
        self.rgbx_torch_copy = False
        self.first_chunk_fraction = 0.25   # share of a batch in the first of the two pipelined encode chunks (set_image_batch)
        self._stage_busy = {}  # staging buffer key -> event recorded after the last H2D copy out of it
        self._h2d_stream = None  # side stream of the staged host-to-device copies (_stage_to_device)
        self._sender = None      # one helper thread that stages and sends the later chunks of a batch (_send_later)

    @staticmethod
    def _rgbx_view(im):
        """Zero-copy uint8 [H, W, 4] view of the storage of a PIL "RGB" image (Pillow keeps RGB as 4-byte pixels and exports them
        through the Arrow C data interface, Image.__arrow_c_array__, Pillow >= 11.2), or None where that is not available.
        Packing to 3 bytes per pixel (Image.tobytes, what np.asarray(image) runs) costs 3 ms per 1024 x 1024 image and is the
        longest host-side piece of a reference-shaped set_image_batch call; the device-side resize reads 4-byte pixels as well."""
        if im.mode != "RGB" or not hasattr(im, "__arrow_c_array__"):
            return None
        try:
            import pyarrow as pa
            flat = pa.array(im).values.to_numpy(zero_copy_only=True)
        except Exception:  # noqa: BLE001  (no pyarrow, an image held in several blocks, ...): the caller packs instead
            return None
        w, h = im.size
        return flat.reshape(h, w, 4) if flat.dtype == np.uint8 and flat.size == h * w * 4 else None

    def _stage_pil_batch(self, images, slot: int = 0, rgbx: bool = False) -> torch.Tensor:
        """Equal-sized PIL images -> ONE pinned uint8 [B, H, W, 3] host buffer (reused across calls; pinned allocations cost
        tens of milliseconds) filled by a few worker threads (PIL's raw encoder and numpy's copies release the GIL), ready
        for a single asynchronous host-to-device copy.  ``rgbx``: stage Pillow's own 4-byte pixels ([B, H, W, 4], see
        _rgbx_view) when every image of the batch exports them; the returned buffer's last dimension says which it was."""
        b, (w, h) = len(images), images[0].size
        views = [self._rgbx_view(im) for im in images] if rgbx else None
        ch = 4 if views is not None and all(v is not None for v in views) else 3
        key = (b, h, w, slot, ch)   # `slot`: half batches in flight at the same time get buffers of their own
        buf = self._stage.get(key)
        if buf is None:
            if len(self._stage) >= 6:
                self._stage.clear()
                self._stage_busy.clear()
            buf = torch.empty((b, h, w, ch), dtype=torch.uint8)
            buf = self._stage[key] = buf.pin_memory() if self.device.type == "cuda" else buf
        ev = self._stage_busy.get(key)
        if ev is not None:
            ev.synchronize()     # the host-to-device copy that last read this buffer must be done before it is refilled
        view = buf.numpy()

        if ch == 4 and self.rgbx_torch_copy:
            # Pillow's own 4-byte pixels: plain copies.  torch's copy_ splits every 4 MB image over the host cores itself;
            # np.copyto from a thread pool did NOT run in parallel (25 ms for 32 images on 8 cores against 3.7 ms this way,
            # measured on the host alone) and was the longest piece of an API-level step.
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")       # the Arrow-backed views are read-only: they are only read
                for i in range(b):
                    buf[i].copy_(torch.from_numpy(views[i]))
            return buf

        def fill(i):
            if ch == 4:
                np.copyto(view[i], views[i])
                return
            im = images[i] if images[i].mode == "RGB" else images[i].convert("RGB")
            np.copyto(view[i], np.frombuffer(im.tobytes(), dtype=np.uint8).reshape(h, w, 3))

        if b >= 4:
            if self._pool is None:
                self._pool = ThreadPoolExecutor(max_workers=host_threads(), thread_name_prefix="esam3-stage")
            list(self._pool.map(fill, range(b)))
        else:
            for i in range(b):
                fill(i)
        return buf

    # ---- helpers -------------------------------------------------------------------------------
    def _to_hwc_u8(self, image):
        """-> (uint8 HWC tensor on device, height, width) following set_image's type handling
        (sam3_image_processor.py:50-57): PIL -> size; ndarray/Tensor -> shape[-2:]."""
        if _PILImage is not None and isinstance(image, _PILImage.Image):
            width, height = image.size
            t = torch.from_numpy(np.array(image.convert("RGB")))  # a writable copy: PIL's buffer is read-only
        elif isinstance(image, np.ndarray):
            height, width = image.shape[-2:]  # (sic) the reference reads CHW-style dims here
            arr = image if image.ndim == 3 else image[:, :, None]
            t = torch.from_numpy(np.ascontiguousarray(arr))
        elif isinstance(image, torch.Tensor):
            height, width = image.shape[-2:]
            t = image.permute(1, 2, 0) if image.dim() == 3 else image  # CHW -> HWC
        else:
            raise ValueError("Image must be a PIL image or a tensor")
        if t.dtype != torch.uint8:
            if t.is_floating_point():  # v2.ToDtype(uint8, scale=True)
                t = (t * 255.999).clamp(0, 255).to(torch.uint8)
            else:
                t = t.to(torch.uint8)
        return t.contiguous(), int(height), int(width)

    def _stage_to_device(self, images, slot: int = 0, wait: bool = True):
        """staged PIL batch -> device (asynchronous copy); an event guards the pinned buffer until the copy has run.  Images that
        the device will resize anyway travel as Pillow's 4-byte pixels where possible (see _rgbx_view).  ``wait=False`` (the helper
        thread of set_image_batch): returns (device tensor, event) and leaves the ordering against the compute stream to the caller."""
        b, (w, h) = len(images), images[0].size
        rgbx = (h, w) != (self.resolution, self.resolution) and hasattr(self.model.engine, "preprocess_resize_u8_batch")
        host = self._stage_pil_batch(images, slot, rgbx=rgbx)
        if self.device.type != "cuda":
            dev_t = host.to(self.device, non_blocking=True)
            return dev_t if wait else (dev_t, None)
        # The copy goes out on a SIDE stream (round 5): issued on the compute stream it queued behind the previous chunk's encode, and
        # the device then sat idle for the 2 ms the second chunk's pixels took to arrive (tools/api_level_probe.py: 10.8 ms of device
        # tail per step against 9.6 ms for the same step with resident inputs).  The compute stream waits for the copy's event only.
        if self._h2d_stream is None:
            self._h2d_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._h2d_stream):
            dev_t = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._h2d_stream)
        self._stage_busy[(b, h, w, slot, host.shape[-1])] = ev     # the pinned buffer is refilled only after the copy has run
        if not wait:
            return dev_t, ev
        return self._arrived(dev_t, ev)

    def _arrived(self, dev_t: torch.Tensor, ev) -> torch.Tensor:
        """order the compute stream of THIS thread behind a staged copy"""
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            dev_t.record_stream(cur)                               # allocated on the side stream, consumed on the compute stream
        return dev_t

    def _send_later(self, images, slot: int):
        """stage + copy a chunk on the helper thread (round 6): the pixels of chunk k + 1 are converted and put on the wire while the
        calling thread is inside the engine call that launches chunk k's encode (a ctypes call: the GIL is free).  Round 5 did both on
        one thread: the second chunk's copy was issued only after the first chunk's ~150 launches, and the device idled for the whole
        transfer (profiles/r06/api_timeline_before.txt: 1.8 ms per step)."""
        if self._sender is None:
            self._sender = ThreadPoolExecutor(max_workers=1, thread_name_prefix="esam3-send")

        def work():
            if self.device.type == "cuda":
                with torch.cuda.device(self.device):               # the current device is a per-thread setting
                    return self._stage_to_device(images, slot=slot, wait=False)
            return self._stage_to_device(images, slot=slot, wait=False)

        return self._sender.submit(work)

    def _preprocess(self, hwc_u8_list: List[torch.Tensor]) -> torch.Tensor:
        """The reference's transform (uint8 -> Resize(1008) -> float/255 -> Normalize(.5,.5)) on
        the device; images already at the network resolution skip the resize exactly as
        torchvision's Resize does."""
        r = self.resolution
        for t in hwc_u8_list:
            if t.dim() != 3 or t.shape[-1] != 3:
                raise ValueError(f"expected an RGB image, got shape {tuple(t.shape)}")
        if all(tuple(t.shape[:2]) == (r, r) for t in hwc_u8_list):
            batch = torch.stack([t.to(self.device) for t in hwc_u8_list], dim=0)  # every image moved to the engine's device
            return self.model.engine.preprocess_u8(batch)
        out = torch.empty((len(hwc_u8_list), 3, r, r), dtype=torch.float32, device=self.device)
        eng = self.model.engine
        same_place = all(not t.is_cuda for t in hwc_u8_list) or all(t.device == self.device for t in hwc_u8_list)
        if same_place and len({tuple(t.shape) for t in hwc_u8_list}) == 1 and hasattr(eng, "preprocess_resize_u8_batch"):
            # one size, all on the host or all on the engine's device: ONE host-to-device copy of the stacked batch and
            # ONE resize launch (mixed-device lists take the per-image loop below, which moves every image itself)
            batch = torch.stack(hwc_u8_list, dim=0)
            if not batch.is_cuda:
                batch = batch.pin_memory().to(self.device, non_blocking=True)
            return eng.preprocess_resize_u8_batch(batch.contiguous(), out)
        for i, t in enumerate(hwc_u8_list):
            t = t.to(self.device).contiguous()
            if tuple(t.shape[:2]) == (r, r):
                out[i] = eng.preprocess_u8(t[None])[0]
            else:
                eng.preprocess_resize_u8(t, out[i])
        return out

    # ---- reference API ---------------------------------------------------------------------------
    @torch.inference_mode()
    def set_image(self, image, state=None):
        """Sets the image on which we want to do predictions."""
        if state is None:
            state = {}
        t, height, width = self._to_hwc_u8(image)
        x = self._preprocess([t])
        state["original_height"] = height
        state["original_width"] = width
        state["backbone_out"] = self.model.backbone.forward_image(x)
        return state

    @torch.inference_mode()
    def set_image_batch(self, images, state=None):
        """Sets the image batch on which we want to do predictions."""
        if state is None:
            state = {}
        if not isinstance(images, list):
            raise ValueError("Images must be a list of PIL images or tensors")
        assert len(images) > 0, "Images list must not be empty"
        assert _PILImage is not None and isinstance(images[0], _PILImage.Image), \
            "Images must be a list of PIL images"
        state["original_heights"] = [image.height for image in images]
        state["original_widths"] = [image.width for image in images]
        r = self.resolution
        eng = self.model.engine
        same = all(isinstance(im, _PILImage.Image) and im.size == images[0].size for im in images) and hasattr(eng, "preprocess_resize_u8_batch")
        if same and len(images) >= 8 and hasattr(self.model, "_features_dict"):
            # Equal sizes, a real batch: two chunks in a software pipeline -- while the device encodes the first
            # the host converts the PIL images of the second into ITS pinned staging buffer (the conversion is the longest
            # host-side piece of an API-level step).  Every chunk writes its slice of the full feature buffers.
            b = len(images)
            dt, dev = eng.torch_dtype, self.device
            want3, want2 = self.model.dual_neck, self.model.inst_interactive_predictor is not None
            full = {}
            if want3:
                full["sam3_fpn"] = [torch.empty((b, h, h, c), dtype=dt, device=dev) for h, c in ((288, 256), (144, 256), (72, 256))]
            if want2:
                full["sam2_fpn"] = [torch.empty((b, h, h, c), dtype=dt, device=dev) for h, c in ((288, 32), (144, 64), (72, 256))]
            # uneven chunks: the device idles until the first chunk is staged and copied (1.3 + 1.25 ms for 16 images), so the first
            # one is a quarter of the batch; the rest travels under its encode
            first = min(b - 1, max(4, int(b * self.first_chunk_fraction)))
            bounds = ((0, first), (first, b))
            if self._pool is None:                                 # both threads fill their staging buffers through it
                self._pool = ThreadPoolExecutor(max_workers=host_threads(), thread_name_prefix="esam3-stage")
            if self._h2d_stream is None and dev.type == "cuda":
                self._h2d_stream = torch.cuda.Stream(device=dev)
            nxt = self._send_later(images[first:b], slot=1)        # chunk 1 is staged and sent while chunk 0 is staged, sent and launched here
            batch = self._stage_to_device(images[0:first], slot=0)
            for ci, (a, e) in enumerate(bounds):
                if ci:
                    batch = self._arrived(*nxt.result())
                if tuple(batch.shape[1:3]) == (r, r):
                    x = eng.preprocess_u8(batch)
                else:
                    x = eng.preprocess_resize_u8_batch(batch, torch.empty((e - a, 3, r, r), dtype=torch.float32, device=dev))
                eng.encode(x, want_sam3=want3, want_sam2=want2, out={k: [t[a:e] for t in v] for k, v in full.items()})
            state["backbone_out"] = self.model._features_dict(full, b)
            return state
        if same:
            # equal sizes: pinned staging buffer -> one H2D copy -> one preprocessing launch
            batch = self._stage_to_device(images)
            if tuple(batch.shape[1:3]) == (r, r):
                x = eng.preprocess_u8(batch)
            else:
                x = eng.preprocess_resize_u8_batch(batch, torch.empty((len(images), 3, r, r), dtype=torch.float32, device=self.device))
        else:
            x = self._preprocess([self._to_hwc_u8(im)[0] for im in images])
        state["backbone_out"] = self.model.backbone.forward_image(x)
        return state

    @torch.inference_mode()
    def set_image_tensor_batch(self, images_nchw_f32: torch.Tensor, original_hw=None, state=None):
        """Extension (no reference counterpart): batch that is already normalised NCHW fp32 on
        the device -- the "tensor-in" timing variant of SURVEY.md §8(d)."""
        if state is None:
            state = {}
        b = images_nchw_f32.shape[0]
        hw = original_hw if original_hw is not None else [(self.resolution, self.resolution)] * b
        state["original_heights"] = [h for h, _ in hw]
        state["original_widths"] = [w for _, w in hw]
        state["backbone_out"] = self.model.backbone.forward_image(images_nchw_f32)
        return state

    @torch.inference_mode()
    def set_text_prompt(self, prompt: str, state: Dict):
        """Sets the text prompt and runs the grounding detector (sam3_image_processor.py:115-131)."""
        if "backbone_out" not in state:
            raise ValueError("You must call set_image before set_text_prompt")
        text_outputs = self.model.backbone.forward_text([prompt], device=self.device)
        state["backbone_out"].update(text_outputs)  # erases the previous text prompt if any
        if "geometric_prompt" not in state:
            state["geometric_prompt"] = self.model._get_dummy_prompt()
        return self._forward_grounding(state)

    def _ensure_text(self, state):
        """Without a text prompt the reference encodes the word "visual" so that the detector relies on the
        geometric prompt alone (sam3_image_processor.py:140-146,166-172)."""
        if "language_features" not in state["backbone_out"]:
            state["backbone_out"].update(self.model.backbone.forward_text(["visual"], device=self.device))
        if "geometric_prompt" not in state:
            state["geometric_prompt"] = self.model._get_dummy_prompt()

    @torch.inference_mode()
    def add_geometric_prompt(self, box, label, state):
        """Adds a box prompt ([cx, cy, w, h] normalised to [0, 1]; label True = positive) and reruns the
        detector (sam3_image_processor.py:130-158)."""
        if "backbone_out" not in state:
            raise ValueError("You must call set_image before add_geometric_prompt")
        self._ensure_text(state)
        boxes = torch.tensor(box, device=self.device, dtype=torch.float32).view(1, 1, 4)
        labels = torch.tensor([label], device=self.device, dtype=torch.bool).view(1, 1)
        state["geometric_prompt"].append_boxes(boxes, labels)
        return self._forward_grounding(state)

    @torch.inference_mode()
    def add_point_prompt(self, point, label, state):
        """Adds a point prompt ([x, y] in pixels of the original image; label 1 = foreground, 0 = background)
        and reruns the detector (sam3_image_processor.py:160-190)."""
        if "backbone_out" not in state:
            raise ValueError("You must call set_image before add_point_prompt")
        self._ensure_text(state)
        x_norm = point[0] / state["original_width"]
        y_norm = point[1] / state["original_height"]
        points = torch.tensor([[x_norm, y_norm]], device=self.device, dtype=torch.float32).view(1, 1, 2)
        labels = torch.tensor([label], device=self.device, dtype=torch.bool).view(1, 1)
        state["geometric_prompt"].append_points(points, labels)
        return self._forward_grounding(state)

    @torch.inference_mode()
    def _forward_grounding(self, state: Dict):
        """sam3_image_processor.py:219-259: scores = sigmoid(logits) * sigmoid(presence), keep > threshold,
        boxes to XYXY pixels, masks bilinearly upsampled to the original size and passed through a sigmoid."""
        out = self.model.forward_grounding(backbone_out=state["backbone_out"], find_input=None, find_target=None,
                                           geometric_prompt=state["geometric_prompt"])
        probs = (out["pred_logits"].sigmoid() * out["presence_logit_dec"].sigmoid().unsqueeze(1)).squeeze(-1)
        keep = probs > self.confidence_threshold
        scores = probs[keep]
        cxcywh = out["pred_boxes"][keep]
        cx, cy, w, h = cxcywh.unbind(-1)
        boxes = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)
        img_h, img_w = state["original_height"], state["original_width"]
        boxes = boxes * torch.tensor([img_w, img_h, img_w, img_h], dtype=torch.float32, device=boxes.device)[None]
        low = out["pred_masks"][keep].contiguous()  # [K, 288, 288]
        if low.shape[0] > 0:
            logits = self.model.engine.postprocess(low[:, None], (img_h, img_w), return_logits=True, max_hole_area=0.0)
            logits = torch.sigmoid_(logits)
        else:
            logits = torch.empty((0, 1, img_h, img_w), dtype=torch.float32, device=low.device)
        state["masks_logits"] = logits
        state["masks"] = logits > 0.5
        state["boxes"] = boxes
        state["scores"] = scores
        return state

    def reset_all_prompts(self, state):
        """Removes all the prompts and results (sam3_image_processor.py:190-206)."""
        if "backbone_out" in state:
            for k in ("language_features", "language_mask", "language_embeds"):
                state["backbone_out"].pop(k, None)
        for k in ("geometric_prompt", "boxes", "masks", "masks_logits", "scores"):
            state.pop(k, None)

    @torch.inference_mode()
    def set_confidence_threshold(self, threshold: float, state=None):
        """sam3_image_processor.py:208-217: filtering again means running the heads again."""
        self.confidence_threshold = threshold
        if state is not None and "boxes" in state:
            return self._forward_grounding(state)
        return state
