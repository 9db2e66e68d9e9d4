"""Multi-GPU execution of the image path: one process per GPU, images sharded, masks gathered.

The path is embarrassingly parallel by image (no cross-image op in eval; SURVEY.md §8(e), the
reference's own multi-GPU eval shards images the same way,
sam3/scripts/eval/gold/eval_efficientsam3_all_subsets.py:302-307).  Every rank holds a full
weight replica and processes a contiguous shard of the batch; the only exchange step is one
gather of the per-image outputs to a root rank (RCCL over xGMI on GPUs - backend "nccl" - or
gloo on CPU for the tests).  No collective touches the model's data path.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_ranks() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as exported by torch.distributed.run."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: Optional[str] = None, device: Optional[torch.device] = None,
                       force: bool = False) -> Tuple[int, int, int]:
    """Join the job described by the environment (no-op for a single process unless ``force``: a one-rank group, which
    is how the collectives of this module are exercised on RCCL with device tensors on a single-GPU box)."""
    rank, local_rank, world = env_ranks()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this platform
        if backend is None:
            backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kw = {"device_id": device} if backend == "nccl" and device is not None else {}
        if world == 1 and ("MASTER_PORT" not in os.environ or "RANK" not in os.environ):
            # a forced one-rank group needs no rendezvous over TCP: an in-process store (picking a free port by bind-then-close
            # would leave a window in which another process can take it).  Also taken when MASTER_PORT is exported but RANK is
            # not (a plain `python` run inside a job's environment): env:// would raise "RANK expected, but not set"
            dist.init_process_group(backend, store=dist.HashStore(), rank=0, world_size=1, **kw)
        else:
            dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [start, stop) of `n_items` for `rank`; the first n % world ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(int(n_items), world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]


def _collective_needed(group, force: bool) -> bool:
    """world > 1, or a one-rank group that the caller wants driven through the collective anyway (``force``)."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or force


def gather_to_root(local: torch.Tensor, n_items: Optional[int] = None, dst: int = 0,
                   group=None, force_collective: bool = False) -> Optional[torch.Tensor]:
    """Gather per-image outputs [n_local, ...] of every rank on `dst`, in image order.

    With `n_items` given the shards follow shard_bounds(n_items, r, world) (ragged shards are
    padded to the largest one for the collective and trimmed on the root); without it every rank
    must hold the same number of rows.  Returns the [n_items, ...] tensor on `dst`, None elsewhere.
    """
    if not _collective_needed(group, force_collective):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_items, world) if n_items is not None else [local.shape[0]] * world
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: {local.shape[0]} rows, expected shard of {sizes[rank]}")
    cap = max(sizes)
    send = local
    if local.shape[0] != cap:
        send = local.new_zeros((cap,) + tuple(local.shape[1:]))
        send[: local.shape[0]] = local
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


class MaskGatherer:
    """The path's only exchange step, made cheap for a steady stream of equal-sized steps (bench.py, batch eval):

      * the root's receive buffers and two send buffers are allocated ONCE (the first call) and reused;
      * the collective is issued on a SIDE stream: the compute stream only pays a device-to-device copy of the step's
        masks into a send buffer, the gather itself overlaps with the next step's kernels;
      * two send buffers alternate, so step k+1 can fill its buffer while step k's gather is still in flight; a buffer is
        reused only after the gather that read it has completed.

    `submit(masks)` enqueues the gather of this step; `result()` waits for the most recent gather and returns the
    [world * n_local, ...] tensor on the root (None elsewhere).  Shards must have equal size (pad ragged shards with
    `gather_to_root`).  On CPU tensors (gloo, the tests) the same calls run synchronously.

    Ownership: `submit` always COPIES the caller's tensor into a private slot (also with one rank), so the caller may
    overwrite its buffer as soon as `submit` returns (stream-ordered).  `result()` returns the most recent step only, as
    a tensor that stays valid until the SECOND following `submit` (the two slots alternate); a step whose result was
    never read is dropped when its slot is reused -- `dropped_unread` counts them.

    With one rank nothing is exchanged unless ``force_collective`` is set: then the one-rank process group (backend
    "nccl" = RCCL on a GPU) is driven through exactly the multi-rank code -- private buffers, side stream, `dist.gather`,
    event-ordered slot reuse -- which is how this path is tested on device tensors on a single-GPU box."""

    def __init__(self, dst: int = 0, group=None, force_collective: bool = False):
        self.dst, self.group = dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = _collective_needed(group, force_collective)
        self._send = [None, None]
        self._recv = [None, None]      # root only: one list of world buffers per send slot
        self._work = [None, None]
        self._read = [True, True]      # the slot's gathered result has been handed out (or the slot is empty)
        self._k = 0
        self._stream = None
        self.allocations = 0           # how many times buffers were (re)allocated: 1 in steady state
        self.dropped_unread = 0
        self.side_stream_gathers = 0   # collectives issued on the side stream (device tensors)

    def _alloc(self, like: torch.Tensor):
        self.allocations += 1
        for i in range(2):
            self._send[i] = torch.empty_like(like)
            self._recv[i] = ([torch.empty_like(like) for _ in range(self.world)]
                             if self.collective and self.rank == self.dst else None)
        self._stream = torch.cuda.Stream(device=like.device) if (like.is_cuda and self.collective) else None

    def submit(self, local: torch.Tensor) -> None:
        local = local.contiguous()
        if self._send[0] is None or self._send[0].shape != local.shape or self._send[0].dtype != local.dtype \
                or self._send[0].device != local.device:
            self.flush()
            self._alloc(local)
        i = self._k & 1
        if self._work[i] is not None:       # the gather that last read this slot must be done before it is overwritten
            self._wait(i)
        if not self._read[i]:
            self.dropped_unread += 1
        self._send[i].copy_(local, non_blocking=True)
        self._read[i] = False
        if self.collective:
            if local.is_cuda:
                side = self._stream
                side.wait_stream(torch.cuda.current_stream(local.device))
                with torch.cuda.stream(side):
                    self._work[i] = dist.gather(self._send[i], self._recv[i], dst=self.dst, group=self.group, async_op=True)
                self.side_stream_gathers += 1
            else:
                self._work[i] = dist.gather(self._send[i], self._recv[i], dst=self.dst, group=self.group, async_op=True)
        self._k += 1

    def _wait(self, i: int) -> None:
        w = self._work[i]
        if w is not None:
            w.wait()                        # NCCL: makes the CURRENT stream wait for the collective; gloo: blocks
            self._work[i] = None

    def flush(self) -> None:
        for i in range(2):
            self._wait(i)

    def result(self) -> Optional[torch.Tensor]:
        """the gathered tensor of the most recent submit() (root) / None (other ranks)"""
        if self._k == 0:
            return None
        i = (self._k - 1) & 1
        self._wait(i)
        self._read[i] = True
        if not self.collective:
            return self._send[i]
        if self.rank != self.dst:
            return None
        return self._recv[i][0] if self.world == 1 else torch.cat(self._recv[i], dim=0)


def run_sharded(process: Callable[[Sequence], torch.Tensor], items: Sequence, dst: int = 0,
                group=None) -> Optional[torch.Tensor]:
    """process(shard_of_items) -> [n_local, ...] on every rank; the gathered result on `dst`."""
    if not dist.is_initialized():
        return process(items)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    a, b = shard_bounds(len(items), rank, world)
    return gather_to_root(process(items[a:b]), n_items=len(items), dst=dst, group=group)


# ---- detector outputs of a chunk of video frames: the reference's only GPU-to-GPU tensor collective ----------
def all_gather_tensor(x: torch.Tensor, async_op: bool = False, group=None, force_collective: bool = False):
    """``Sam3ImageOnVideoMultiGPU._gather_tensor`` (sam3/sam3/model/sam3_image.py:869-883): every rank receives every
    rank's ``x`` -> (list of world_size tensors, work handle or None).  The input is made contiguous first (the
    reference notes that NCCL/RCCL all_gather needs it)."""
    if not _collective_needed(group, force_collective):
        return [x], None
    x = x.contiguous()
    outs = [torch.empty_like(x) for _ in range(dist.get_world_size(group))]
    handle = dist.all_gather(outs, x, async_op=async_op, group=group)
    return outs, handle


def local_frame_index(frame_idx_begin: int, frame_idx_end: int, rank: int) -> int:
    """Round-robin frame of a chunk that this rank runs the detector on (sam3_image.py:808-809); ranks past the end
    of the chunk repeat its last frame."""
    return min(frame_idx_begin + rank, frame_idx_end - 1)


def gather_detector_chunk(out_local: dict, frame_idx_begin: int, num_frames: int, sam2_fpn: Optional[Sequence[torch.Tensor]] = None,
                          vision_pos_enc=None, async_op: bool = True, group=None, force_collective: bool = False) -> dict:
    """``_build_multigpu_buffer_next_chunk`` after the detector ran (sam3_image.py:836-867): all-gather the
    detector outputs of this rank's frame (``pred_logits``, ``pred_boxes``, ``pred_boxes_xyxy``, ``pred_masks``) and,
    when given, the three SAM2 FPN levels cast to bf16; returns {frame_idx: {key: (tensor_of_that_frame, handle)}}
    for the frames ``frame_idx_begin + rank`` that exist.  Call ``handle.wait()`` before reading (``async_op``)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    keys = ("pred_logits", "pred_boxes", "pred_boxes_xyxy", "pred_masks")
    gathered = {k: all_gather_tensor(out_local[k], async_op, group, force_collective) for k in keys if k in out_local}
    fpn = None
    if sam2_fpn is not None:
        assert len(sam2_fpn) == 3, "the SAM2 backbone always has 3 levels"
        fpn = [all_gather_tensor(x.to(torch.bfloat16), async_op, group, force_collective) for x in sam2_fpn]
    buf = {}
    for r in range(world):
        f = frame_idx_begin + r
        if f >= num_frames:
            continue
        fb = {k: (v[r], h) for k, (v, h) in gathered.items()}
        if fpn is not None:
            for i, (v, h) in enumerate(fpn):
                fb[f"tracker_backbone_fpn_{i}"] = (v[r], h)
            fb["tracker_backbone_pos_enc"] = (vision_pos_enc, None)
        buf[f] = fb
    return buf


# ---- stage-1 training: the gradient exchange of data-parallel distillation ------------------------
class GradientAllReducer:
    """Bucketed averaging all-reduce of a list of gradient tensors across the data-parallel ranks -- the exchange
    ``torch.nn.parallel.DistributedDataParallel`` performs for the student trunk in stage 1
    (stage1/train_image_encoder_stage1.py:67-72).  Gradients are packed into flat buckets of about ``bucket_bytes``, in
    the order of ``params`` (xGMI rings are per-link bound: a few large collectives beat many small ones; the default
    64 MB is one bucket for an EV-M trunk and a handful for ViT-H).

    Two ways to feed it:

      * ``push(i, grad)`` as the backward pass produces gradient ``i`` (any order): the gradient is copied into its
        bucket at once, and the bucket's asynchronous ``all_reduce`` is issued the moment its LAST member has arrived --
        buckets whose layers finish early are on the wire while the rest of the backward pass still runs.  This is the
        overlap; list ``params`` in the order gradients become available (last layer first) so that buckets fill in turn.
      * ``start(grads)`` when all gradients already exist: pushes them in order (no overlap with compute to be had).

    ``finish(grads)`` waits, divides by the world size and scatters the averages back into ``grads`` in place.  The flat
    buckets are allocated once and reused every step.  Backend "nccl" is RCCL on ROCm; gloo serves the CPU tests; with
    one rank nothing is exchanged unless ``force_collective`` (a one-rank group driven through the same code)."""

    def __init__(self, params: Sequence[torch.Tensor], bucket_bytes: int = 64 << 20, group=None, force_collective: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = _collective_needed(group, force_collective)
        self.buckets: List[dict] = []
        self._where: List[Tuple[int, int]] = [(-1, 0)] * len(params)   # parameter -> (bucket, element offset)
        cur, cur_bytes = [], 0
        for i, p in enumerate(params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != params[cur[0]].dtype or p.device != params[cur[0]].device):
                self._close(params, cur)
                cur, cur_bytes = [], 0
            cur.append(i)
            cur_bytes += nbytes
        if cur:
            self._close(params, cur)
        self._pending: List[Tuple[int, object]] = []
        self._missing = [len(b["idx"]) for b in self.buckets]
        self._got = [False] * len(params)
        self.issue_order: List[int] = []   # bucket indices in the order their all-reduce was issued (last step)

    def _close(self, params, idx):
        n = sum(params[i].numel() for i in idx)
        bi = len(self.buckets)
        off = 0
        for i in idx:
            self._where[i] = (bi, off)
            off += params[i].numel()
        self.buckets.append({"idx": list(idx), "flat": torch.empty(n, dtype=params[idx[0]].dtype, device=params[idx[0]].device)})

    @property
    def n_buckets(self) -> int:
        return len(self.buckets)

    def push(self, i: int, grad: torch.Tensor) -> None:
        """Gradient of parameter ``i`` is ready: copy it into its bucket; issue the bucket when it is complete."""
        if self._got[i]:
            raise RuntimeError(f"gradient {i} pushed twice in one step (finish() the previous step first)")
        if not self._pending and not any(self._got):
            self.issue_order = []
        bi, off = self._where[i]
        b = self.buckets[bi]
        b["flat"][off:off + grad.numel()].copy_(grad.reshape(-1))
        self._got[i] = True
        self._missing[bi] -= 1
        if self._missing[bi] == 0:
            work = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.collective else None
            self._pending.append((bi, work))
            self.issue_order.append(bi)

    def start(self, grads: Sequence[torch.Tensor]) -> None:
        """All gradients exist already: push them in order (every bucket is issued as soon as it is packed)."""
        assert not self._pending and not any(self._got), "finish() the previous step first"
        for i, g in enumerate(grads):
            self.push(i, g)

    def finish(self, grads: Sequence[torch.Tensor]) -> None:
        """Wait for the buckets, average, and write the result back into ``grads`` (in place)."""
        if any(m != 0 for m in self._missing):
            raise RuntimeError("finish() before every gradient was pushed")
        for bi, work in self._pending:
            if work is not None:
                work.wait()
            b = self.buckets[bi]
            if self.world > 1:
                b["flat"].div_(self.world)
            off = 0
            for i in b["idx"]:
                g = grads[i]
                g.copy_(b["flat"][off:off + g.numel()].view_as(g))
                off += g.numel()
        self._pending = []
        self._missing = [len(b["idx"]) for b in self.buckets]
        self._got = [False] * len(self._got)

    def __call__(self, grads: Sequence[torch.Tensor]) -> None:
        self.start(grads)
        self.finish(grads)


# ---- the multi-GPU entry of the video path: detector outputs computed a chunk at a time, one frame per rank -----------------
class VideoGroundingMultiGPU:
    """``Sam3ImageOnVideoMultiGPU.forward_video_grounding_multigpu`` (sam3/sam3/model/sam3_image.py:701-790) around a
    detector callable: the frames of a video are processed in chunks of ``world_size`` frames, every rank runs the detector
    on ONE frame of the chunk (round robin, ``local_frame_index``), the outputs (and the bf16 SAM2 features) are
    all-gathered asynchronously (``gather_detector_chunk``) into ``multigpu_buffer``; a call for frame ``t``

      1. builds the chunk that contains ``t`` if it is not buffered yet (only the first chunk: later ones are built ahead),
      2. reads frame ``t`` out of the buffer, waiting for its all-gather handles,
      3. drops the previous chunk from the buffer,
      4. builds the NEXT chunk in tracking direction, so that its all-gather overlaps whatever the caller does with
         frame ``t`` (the tracker, which is out of scope here).

    ``detect(frame_idx) -> (out_local, sam2_fpn, vision_pos_enc)``: the detector on one frame -- a dict with ``pred_logits``,
    ``pred_boxes``, ``pred_boxes_xyxy``, ``pred_masks`` (``model.forward_grounding`` / ``engine.ground`` outputs), the three
    SAM2 FPN levels or None (``gather_backbone_out``), and the position encodings (identical on all frames, not gathered).
    Mask NMS on the detections (``run_nms``, perflib/nms.py) belongs to the video tracker's perf helpers and is not part of
    this path."""

    def __init__(self, detect: Callable[[int], tuple], async_all_gather: bool = True, group=None, force_collective: bool = False):
        self.detect = detect
        self.async_all_gather = async_all_gather
        self.group = group
        self.force_collective = force_collective
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.chunks_built: List[Tuple[int, int]] = []     # (begin, end) of every chunk this rank took part in, in order

    def _build_chunk(self, begin: int, end: int, num_frames: int, multigpu_buffer: dict) -> None:
        frame = local_frame_index(begin, end, self.rank)
        out_local, sam2_fpn, pos_enc = self.detect(frame)
        multigpu_buffer.update(gather_detector_chunk(out_local, begin, num_frames, sam2_fpn=sam2_fpn, vision_pos_enc=pos_enc,
                                                     async_op=self.async_all_gather, group=self.group,
                                                     force_collective=self.force_collective))
        self.chunks_built.append((begin, end))

    def forward(self, frame_idx: int, num_frames: int, multigpu_buffer: dict, track_in_reverse: bool = False,
                return_sam2_backbone_feats: bool = False) -> dict:
        w = self.world_size
        cur_b = frame_idx - frame_idx % w
        cur_e = min(cur_b + w, num_frames)
        if frame_idx not in multigpu_buffer:
            self._build_chunk(cur_b, cur_e, num_frames, multigpu_buffer)
        out = {}
        for k, (v, handle) in multigpu_buffer[frame_idx].items():
            # (the reference tests the prefix "sam2_backbone_" here although its buffer stores "tracker_backbone_*", so its
            # flag never filters anything; the keys that exist are filtered here)
            if k.startswith("tracker_backbone_") and not return_sam2_backbone_feats:
                continue
            if handle is not None:
                handle.wait()
            out[k] = v
        # drop the previous chunk
        if not track_in_reverse and cur_b - w >= 0:
            prev = range(cur_b - w, cur_b)
        elif track_in_reverse and cur_e < num_frames:
            prev = range(cur_e, min(cur_e + w, num_frames))
        else:
            prev = range(0)
        for f in prev:
            multigpu_buffer.pop(f, None)
        # build the next chunk ahead of time
        if not track_in_reverse and cur_e < num_frames:
            nb, ne = cur_e, min(cur_e + w, num_frames)
        elif track_in_reverse and cur_b - w >= 0:
            nb, ne = cur_b - w, cur_b
        else:
            nb = ne = None
        if nb is not None and nb not in multigpu_buffer:
            self._build_chunk(nb, ne, num_frames, multigpu_buffer)
        return out
