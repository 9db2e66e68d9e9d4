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


def init_process_group(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int, int]:
    """Join the job described by the environment (no-op for a single process)."""
    rank, local_rank, world = env_ranks()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this platform
        if backend is None:
            backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kw = {"device_id": device} if backend == "nccl" and device is not None else {}
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


def gather_to_root(local: torch.Tensor, n_items: Optional[int] = None, dst: int = 0,
                   group=None) -> Optional[torch.Tensor]:
    """Gather per-image outputs [n_local, ...] of every rank on `dst`, in image order.

    With `n_items` given the shards follow shard_bounds(n_items, r, world) (ragged shards are
    padded to the largest one for the collective and trimmed on the root); without it every rank
    must hold the same number of rows.  Returns the [n_items, ...] tensor on `dst`, None elsewhere.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
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


def run_sharded(process: Callable[[Sequence], torch.Tensor], items: Sequence, dst: int = 0,
                group=None) -> Optional[torch.Tensor]:
    """process(shard_of_items) -> [n_local, ...] on every rank; the gathered result on `dst`."""
    if not dist.is_initialized():
        return process(items)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    a, b = shard_bounds(len(items), rank, world)
    return gather_to_root(process(items[a:b]), n_items=len(items), dst=dst, group=group)
