"""world_size-2 CPU (gloo) coverage of the multi-GPU path: image shards per
rank, one gather of the per-image outputs on rank 0, no other collective.  bench.py --gpus N
uses the same helpers with backend "nccl" (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from efficientsam3_amd import dist as esdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_masks(ids):
    """Deterministic stand-in for encode+decode of images `ids`: uint8 [n,1,8,8]."""
    out = torch.zeros((len(ids), 1, 8, 8), dtype=torch.uint8)
    for j, i in enumerate(ids):
        g = torch.Generator().manual_seed(1000 + int(i))
        out[j] = (torch.rand((1, 8, 8), generator=g) > 0.5).to(torch.uint8)
    return out


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, lr, w = esdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    try:
        ids = list(range(n_items))
        got = esdist.run_sharded(_fake_masks, ids, dst=0)
        # equal-shard fast path (what bench.py does each step)
        a, b = esdist.shard_bounds(world * 3, rank, world)
        eq = esdist.gather_to_root(_fake_masks(range(a, b)), n_items=world * 3, dst=0)
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            q.put((got.numpy(), eq.numpy(), float(t.item())))
        else:
            assert got is None and eq is None
            q.put(None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 5, 1])
def test_sharded_gather_matches_single_process(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got, eq, tmax = [r for r in results if r is not None][0]
    np.testing.assert_array_equal(got, _fake_masks(range(n_items)).numpy())
    np.testing.assert_array_equal(eq, _fake_masks(range(world * 3)).numpy())
    assert tmax == float(world)


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [esdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == esdist.shard_sizes(n, world)
    with pytest.raises(ValueError):
        esdist.shard_bounds(4, 2, 2)


def test_single_process_is_passthrough():
    x = _fake_masks(range(3))
    assert esdist.gather_to_root(x) is x
    assert torch.equal(esdist.run_sharded(_fake_masks, list(range(3))), x)
