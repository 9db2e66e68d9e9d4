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


def _detector_out(frame):
    g = torch.Generator().manual_seed(77 + frame)
    return {"pred_logits": torch.randn((1, 6, 1), generator=g), "pred_boxes": torch.rand((1, 6, 4), generator=g),
            "pred_boxes_xyxy": torch.rand((1, 6, 4), generator=g), "pred_masks": torch.randn((1, 6, 8, 8), generator=g),
            "extra_key_not_gathered": torch.zeros(1)}


def _fpn(frame):
    g = torch.Generator().manual_seed(900 + frame)
    return [torch.randn((1, c, s, s), generator=g) for c, s in ((4, 8), (8, 4), (16, 2))]


def _chunk_worker(rank, world, port, begin, num_frames, q, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    esdist.init_process_group(backend)
    dev = torch.device("cuda", rank) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(dev)
    try:
        end = min(begin + world, num_frames)
        mine = esdist.local_frame_index(begin, end, rank)
        # non-contiguous input on purpose: the helper must make it contiguous
        out_local = {k: (v.transpose(-1, -2).contiguous().transpose(-1, -2) if v.dim() == 4 else v).to(dev)
                     for k, v in _detector_out(mine).items()}
        buf = esdist.gather_detector_chunk(out_local, begin, num_frames, sam2_fpn=[x.to(dev) for x in _fpn(mine)], vision_pos_enc="pos",
                                           async_op=True)
        res = {}
        for f, fb in buf.items():
            for k, (t, h) in fb.items():
                if h is not None:
                    h.wait()
            res[f] = {k: (t.float().cpu().numpy() if torch.is_tensor(t) else t) for k, (t, h) in fb.items()}
        q.put((rank, mine, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("begin,num_frames", [(0, 7), (6, 7)])
def test_detector_chunk_all_gather_rccl(begin, num_frames):
    """The same chunk exchange with device tensors over RCCL (backend "nccl") between two GPUs of one node; skipped on a
    single-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    test_detector_chunk_all_gather(begin, num_frames, backend="nccl")


@pytest.mark.parametrize("begin,num_frames", [(0, 7), (6, 7)])
def test_detector_chunk_all_gather(begin, num_frames, backend="gloo"):
    """Sam3ImageOnVideoMultiGPU's chunk exchange (sam3_image.py:792-883) on gloo: every rank ends with the detector
    outputs and bf16 SAM2 features of every frame of the chunk; frames past the end of the video are dropped."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, begin, num_frames, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = [f for f in range(begin, begin + world) if f < num_frames]
    for rank, mine, res in results:
        assert mine == min(begin + rank, min(begin + world, num_frames) - 1)
        assert sorted(res) == frames
        for f in frames:
            want = _detector_out(f)
            assert set(res[f]) == {"pred_logits", "pred_boxes", "pred_boxes_xyxy", "pred_masks", "tracker_backbone_fpn_0",
                                   "tracker_backbone_fpn_1", "tracker_backbone_fpn_2", "tracker_backbone_pos_enc"}
            for k in ("pred_logits", "pred_boxes", "pred_boxes_xyxy", "pred_masks"):
                np.testing.assert_array_equal(res[f][k], want[k].numpy())
            for i, x in enumerate(_fpn(f)):
                np.testing.assert_array_equal(res[f][f"tracker_backbone_fpn_{i}"], x.to(torch.bfloat16).float().numpy())
            assert res[f]["tracker_backbone_pos_enc"] == "pos"


# ---- MaskGatherer: preallocated buffers, gather overlapped with the next step ------------------------------------
def _gatherer_worker(rank, world, port, backend, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(dev)
    esdist.init_process_group(backend, dev)
    try:
        g = esdist.MaskGatherer(dst=0)
        outs = []
        for step in range(4):  # four steps through the two alternating send slots
            local = _fake_masks(range(100 * step + 3 * rank, 100 * step + 3 * rank + 3)).to(dev)
            g.submit(local)
            if step in (1, 3):
                r = g.result()
                outs.append(None if r is None else r.cpu().numpy())
        g.flush()
        assert g.allocations == 1, g.allocations
        assert dist.get_world_size() == world and dist.get_backend() == backend
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def _run_gatherer(backend):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[1] == [None, None]
    for got, step in zip(results[0], (1, 3)):
        want = torch.cat([_fake_masks(range(100 * step + 3 * r, 100 * step + 3 * r + 3)) for r in range(world)]).numpy()
        np.testing.assert_array_equal(got, want)


def test_mask_gatherer_gloo():
    _run_gatherer("gloo")


@pytest.mark.gpu
def test_mask_gatherer_rccl():
    """The same exchange over RCCL (backend "nccl") between two GPUs of one node; skipped on a single-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_gatherer("nccl")


def test_bench_respawns_itself_for_multi_gpu(monkeypatch):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE) re-executes itself through torch.distributed.run
    with N ranks on 127.0.0.1 instead of refusing to run."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_exec(file, argv, env):
        seen.update(file=file, argv=argv, env=env)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execvpe", fake_exec)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert a[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in a and "127.0.0.1" in a
    assert a[-4:] == ["--gpus", "4", "--steps", "2"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


# ---- GradientAllReducer: the data-parallel gradient exchange of stage 1 ---------------------------------------------
def _grads(rank, shapes, dtype=torch.float32):
    g = torch.Generator().manual_seed(4000 + rank)
    return [torch.randn(s, generator=g).to(dtype) for s in shapes]


_GSHAPES = [(64, 3, 3, 3), (64,), (128, 64, 1, 1), (1000,), (7, 5), (300000,), (1,)]


def _allreduce_worker(rank, world, port, backend, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    esdist.init_process_group(backend)
    dev = torch.device("cuda", rank) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(dev)
    try:
        grads = [g.to(dev) for g in _grads(rank, _GSHAPES)]
        red = esdist.GradientAllReducer(grads, bucket_bytes=256 << 10)   # small buckets: several of them, one tensor alone > bucket
        flats = [b["flat"].data_ptr() for b in red.buckets]
        for step in range(2):                                            # the flat buckets are reused
            cur = [g.clone() * (step + 1) for g in grads]
            red(cur)
            assert [b["flat"].data_ptr() for b in red.buckets] == flats
        q.put((rank, red.n_buckets, [c.cpu().numpy() for c in cur]))
    finally:
        dist.destroy_process_group()


def _run_allreduce(backend):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [2 * (a + b) / 2 for a, b in zip(_grads(0, _GSHAPES), _grads(1, _GSHAPES))]   # step 2: grads * 2, averaged
    for rank, nb, got in results:
        assert nb >= 3                                                   # (64*27 + 64 + 8192) | 1000 + 35 ... | 300000 alone | ...
        for g, w in zip(got, want):
            np.testing.assert_allclose(g, w.numpy(), rtol=1e-6, atol=1e-7)


def test_gradient_allreducer_gloo():
    """Bucketed averaging all-reduce of a gradient list (stage-1 data parallelism, train_image_encoder_stage1.py:67-72)."""
    _run_allreduce("gloo")


@pytest.mark.gpu
def test_gradient_allreducer_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_allreduce("nccl")


def test_gradient_allreducer_single_process_is_identity():
    grads = _grads(0, _GSHAPES)
    keep = [g.clone() for g in grads]
    esdist.GradientAllReducer(grads)(grads)
    for a, b in zip(grads, keep):
        assert torch.equal(a, b)


# ---- one-rank process group driven through the collectives (force_collective) -------------------------------------------
# A gpurun box has ONE GPU, so the two-rank RCCL tests above are skipped there.  The same code paths -- private buffers, the
# side stream, dist.gather / all_reduce / all_gather, event-ordered slot reuse -- run on DEVICE tensors over backend "nccl"
# (RCCL) with a one-rank group; on CPU the identical test body runs over gloo.
def _one_rank_worker(backend, q):
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(dev)
    esdist.init_process_group(backend, dev, force=True)
    report = {}
    try:
        assert dist.is_initialized() and dist.get_world_size() == 1 and dist.get_backend() == backend
        # MaskGatherer: six steps of "compute" (a matmul chain that keeps the compute stream busy on a GPU) + submit;
        # the caller's buffer is overwritten right after submit (ownership rule), results are read every second step
        g = esdist.MaskGatherer(dst=0, force_collective=True)
        assert g.collective
        work = torch.randn((512, 512), device=dev)
        local = torch.empty((3, 1, 8, 8), dtype=torch.uint8, device=dev)
        outs = []
        for step in range(6):
            for _ in range(4):
                work = torch.tanh(work @ work.t() * 1e-3)
            local.copy_(_fake_masks(range(100 * step, 100 * step + 3)), non_blocking=False)
            g.submit(local)
            local.zero_()                                   # allowed: submit copied it (stream-ordered)
            if step % 2 == 1:
                outs.append(g.result().cpu().numpy().copy())   # the slot is reused two submits later: keep a copy
        g.flush()
        report["gatherer"] = dict(outs=outs, allocations=g.allocations, dropped=g.dropped_unread,
                                  side=g.side_stream_gathers, cuda=dev.type == "cuda")
        # GradientAllReducer through all_reduce, gradients pushed LAST layer first
        grads = [x.to(dev) for x in _grads(0, _GSHAPES)]
        keep = [x.clone() for x in grads]
        red = esdist.GradientAllReducer(grads, bucket_bytes=256 << 10, force_collective=True)
        assert red.collective and red.n_buckets >= 3
        for i in reversed(range(len(grads))):
            red.push(i, grads[i])
        red.finish(grads)
        report["allreduce"] = dict(same=all(torch.equal(a, b) for a, b in zip(grads, keep)), order=list(red.issue_order),
                                   n=red.n_buckets)
        # detector chunk all-gather
        out_local = {k: v.to(dev) for k, v in _detector_out(5).items()}
        buf = esdist.gather_detector_chunk(out_local, 5, 9, sam2_fpn=[x.to(dev) for x in _fpn(5)], vision_pos_enc="pos",
                                           async_op=True, force_collective=True)
        for fb in buf.values():
            for k, (t, h) in fb.items():
                if h is not None:
                    h.wait()
        report["chunk"] = {f: {k: (t.float().cpu().numpy() if torch.is_tensor(t) else t) for k, (t, h) in fb.items()}
                           for f, fb in buf.items()}
        # the chunked video entry on top of it: three frames, one rank -> three chunks of one frame
        v = esdist.VideoGroundingMultiGPU(lambda f: ({k: t.to(dev) for k, t in _detector_out(f).items()}, [x.to(dev) for x in _fpn(f)], "pos"),
                                          force_collective=True)
        vbuf, vok = {}, True
        for t_ in range(3):
            o = v.forward(t_, 3, vbuf, return_sam2_backbone_feats=True)
            vok = vok and torch.equal(o["pred_masks"].cpu(), _detector_out(t_)["pred_masks"]) and o["tracker_backbone_fpn_2"].dtype == torch.bfloat16
        report["video"] = bool(vok and v.chunks_built == [(0, 1), (1, 2), (2, 3)])
        # gather_to_root
        x = _fake_masks(range(5)).to(dev)
        r = esdist.gather_to_root(x, n_items=5, force_collective=True)
        report["gather_to_root"] = bool(torch.equal(r.cpu(), _fake_masks(range(5))) and r.data_ptr() != x.data_ptr())
        q.put(report)
    finally:
        dist.destroy_process_group()


def _run_one_rank(backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(backend, q))
    p.start()
    rep = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    gt = rep["gatherer"]
    for got, step in zip(gt["outs"], (1, 3, 5)):
        np.testing.assert_array_equal(got, _fake_masks(range(100 * step, 100 * step + 3)).numpy())
    assert gt["allocations"] == 1
    assert gt["dropped"] == 2          # steps 0 and 2 were overwritten unread (4 is still in its slot)
    assert gt["side"] == (6 if gt["cuda"] else 0)
    assert rep["allreduce"]["same"] and rep["allreduce"]["n"] >= 3
    assert rep["allreduce"]["order"] == sorted(rep["allreduce"]["order"], reverse=True)   # last bucket went out first
    assert sorted(rep["chunk"]) == [5]
    want = _detector_out(5)
    for k in ("pred_logits", "pred_boxes", "pred_boxes_xyxy", "pred_masks"):
        np.testing.assert_array_equal(rep["chunk"][5][k], want[k].numpy())
    for i, x in enumerate(_fpn(5)):
        np.testing.assert_array_equal(rep["chunk"][5][f"tracker_backbone_fpn_{i}"], x.to(torch.bfloat16).float().numpy())
    assert rep["gather_to_root"] and rep["video"]


def test_one_rank_group_runs_the_collectives_gloo():
    _run_one_rank("gloo")


@pytest.mark.gpu
def test_one_rank_group_runs_the_collectives_rccl():
    """MaskGatherer (side stream), GradientAllReducer.push, gather_detector_chunk and gather_to_root on DEVICE tensors over
    RCCL (backend "nccl") with a one-rank process group: runs on a single-GPU box."""
    _run_one_rank("nccl")


def test_gradient_allreducer_push_any_order_and_guards():
    grads = _grads(0, _GSHAPES)
    keep = [g.clone() for g in grads]
    red = esdist.GradientAllReducer(grads, bucket_bytes=256 << 10)
    order = [3, 0, 6, 1, 5, 2, 4]
    for i in order[:-1]:
        red.push(i, grads[i])
    with pytest.raises(RuntimeError):
        red.finish(grads)              # one gradient is missing
    with pytest.raises(RuntimeError):
        red.push(3, grads[3])          # pushed twice
    red.push(order[-1], grads[order[-1]])
    red.finish(grads)
    assert all(torch.equal(a, b) for a, b in zip(grads, keep))
    red(grads)                         # the state was reset: a second step works


def test_mask_gatherer_single_rank_copies_and_counts_drops():
    g = esdist.MaskGatherer()
    a = _fake_masks(range(3))
    g.submit(a)
    r = g.result()
    assert r.data_ptr() != a.data_ptr() and torch.equal(r, a)
    a.zero_()
    assert torch.equal(r, _fake_masks(range(3)))       # the result does not alias the caller's buffer
    g.submit(_fake_masks(range(3, 6)))
    g.submit(_fake_masks(range(6, 9)))                  # slot of step 0 reused (it was read): no drop
    g.submit(_fake_masks(range(9, 12)))                 # slot of step 1 reused unread: one drop
    assert g.dropped_unread == 1 and g.allocations == 1
    assert torch.equal(g.result(), _fake_masks(range(9, 12)))


# ---- VideoGroundingMultiGPU: the chunked multi-GPU entry of the video path ------------------------------------------------
def _video_worker(rank, world, port, num_frames, reverse, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    esdist.init_process_group("gloo")
    try:
        calls = []

        def detect(frame):
            calls.append(frame)
            return _detector_out(frame), _fpn(frame), "pos"

        v = esdist.VideoGroundingMultiGPU(detect)
        buf, seen, sizes = {}, {}, []
        order = range(num_frames - 1, -1, -1) if reverse else range(num_frames)
        for t in order:
            out = v.forward(t, num_frames, buf, track_in_reverse=reverse, return_sam2_backbone_feats=(t % 2 == 0))
            seen[t] = {k: (x.float().numpy() if torch.is_tensor(x) else x) for k, x in out.items()}
            sizes.append(len(buf))
        q.put((rank, calls, seen, sizes, v.chunks_built))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_frames,reverse", [(5, False), (4, False), (5, True)])
def test_video_grounding_multigpu_chunks(num_frames, reverse):
    """forward_video_grounding_multigpu's bookkeeping (sam3_image.py:701-790) with two ranks on gloo: every frame's detector
    outputs equal the single-process detector's, each rank ran the detector once per chunk on its round-robin frame, the
    next chunk is built one call ahead and the previous one is dropped (at most two chunks buffered)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_video_worker, args=(r, world, port, num_frames, reverse, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_chunks = (num_frames + world - 1) // world
    for rank, calls, seen, sizes, built in results:
        assert sorted(seen) == list(range(num_frames))
        for t, out in seen.items():
            want = _detector_out(t)
            for k in ("pred_logits", "pred_boxes", "pred_boxes_xyxy", "pred_masks"):
                np.testing.assert_array_equal(out[k], want[k].numpy())
            has_fpn = "tracker_backbone_fpn_0" in out
            assert has_fpn == (t % 2 == 0)
            if has_fpn:
                for i, x in enumerate(_fpn(t)):
                    np.testing.assert_array_equal(out[f"tracker_backbone_fpn_{i}"], x.to(torch.bfloat16).float().numpy())
        assert len(calls) == len(built) == n_chunks                      # one detector run per chunk and rank
        for (b, e), f in zip(built, calls):
            assert f == min(b + rank, e - 1) and e == min(b + world, num_frames)
        assert max(sizes) <= 2 * world                                   # current + next chunk, the previous one is dropped
        first = built[0]
        assert first == ((num_frames - 1) // world * world, num_frames) if reverse else first == (0, min(world, num_frames))


def test_video_grounding_single_process():
    v = esdist.VideoGroundingMultiGPU(lambda f: (_detector_out(f), None, None))
    buf = {}
    for t in range(3):
        out = v.forward(t, 3, buf)
        assert set(out) == {"pred_logits", "pred_boxes", "pred_boxes_xyxy", "pred_masks"}
        assert torch.equal(out["pred_masks"], _detector_out(t)["pred_masks"])
    assert v.chunks_built == [(0, 1), (1, 2), (2, 3)] and len(buf) <= 2


# ---- the same bookkeeping pinned against the REAL reference function (oracle/gen_golden_video_grounding.py) ------------------
def _describe(x):
    if torch.is_tensor(x):
        return {"shape": list(x.shape), "dtype": str(x.dtype).replace("torch.", ""), "sum": float(x.double().sum()),
                "abs_sum": float(x.double().abs().sum())}
    return {"value": x}


def _video_trace_worker(rank, world, port, schedules, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    esdist.init_process_group("gloo")
    try:
        traces = {}
        for num_frames, reverse in schedules:
            calls = []

            def detect(frame):
                calls.append(frame)
                return _detector_out(frame), _fpn(frame), "pos"

            v = esdist.VideoGroundingMultiGPU(detect)
            buf, per_call = {}, []
            order = range(num_frames - 1, -1, -1) if reverse else range(num_frames)
            for t in order:
                n0 = len(calls)
                out = v.forward(t, num_frames, buf, track_in_reverse=reverse, return_sam2_backbone_feats=True)
                per_call.append({"frame": t, "detector_ran_on": calls[n0:], "buffered_frames": sorted(buf),
                                 "out": {k: _describe(x) for k, x in sorted(out.items())}})
            traces[f"{num_frames}_{'reverse' if reverse else 'forward'}"] = per_call
        q.put((rank, traces))
    finally:
        dist.destroy_process_group()


def test_video_grounding_matches_reference_trace():
    """`VideoGroundingMultiGPU.forward` against the trace of the reference's own `forward_video_grounding_multigpu`
    (sam3/sam3/model/sam3_image.py:701-883) run on two gloo ranks with the same stub detector
    (tests/golden/video_grounding/trace.json, written by oracle/gen_golden_video_grounding.py): per rank and per call the frames
    the detector ran on (chunk order, round-robin assignment, next chunk built one call ahead), the frames left in the buffer
    (previous chunk dropped), the returned keys and every tensor's shape / dtype / checksum (the all-gathered detector outputs
    and the bf16 SAM2 features)."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "video_grounding", "trace.json")) as f:
        ref = json.load(f)
    world = ref["world_size"]
    schedules = [(5, False), (4, False), (5, True)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_video_trace_worker, args=(r, world, port, schedules, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        for name, ref_calls in ref["ranks"][str(r)].items():
            mine = got[r][name]
            assert len(mine) == len(ref_calls), (r, name)
            for a, b in zip(mine, ref_calls):
                assert a["frame"] == b["frame"] and a["detector_ran_on"] == b["detector_ran_on"], (r, name, a["frame"])
                assert a["buffered_frames"] == b["buffered_frames"], (r, name, a["frame"], a["buffered_frames"], b["buffered_frames"])
                assert sorted(a["out"]) == sorted(b["out"]), (r, name, a["frame"], sorted(a["out"]), sorted(b["out"]))
                for k, d in b["out"].items():
                    m = a["out"][k]
                    if "value" in d:
                        assert m == d, (k, m, d)
                    else:
                        assert m["shape"] == d["shape"] and m["dtype"] == d["dtype"], (k, m, d)
                        assert abs(m["sum"] - d["sum"]) <= 1e-9 * max(1.0, d["abs_sum"]) and abs(m["abs_sum"] - d["abs_sum"]) <= 1e-9 * max(1.0, d["abs_sum"]), (k, m, d)


def _forced_group_without_rank_worker(q):
    # a plain `python` run inside a job's environment: MASTER_PORT exported, RANK / WORLD_SIZE not
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        os.environ.pop(k, None)
    os.environ["MASTER_PORT"] = "29999"
    try:
        esdist.init_process_group("gloo", torch.device("cpu"), force=True)
        q.put(bool(dist.is_initialized() and dist.get_world_size() == 1))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_forced_one_rank_group_with_master_port_but_no_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_group_without_rank_worker, args=(q,))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
    assert p.exitcode == 0


# ---- SyncBatchNorm (train_image_encoder_stage1.py:62-63) -------------------------------------------------------------------------------------
def _bn_case():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 5, 6, 16, generator=g) * 2.0 + torch.randn(16, generator=g)     # NHWC, 8 samples
    dy = torch.randn(8, 5, 6, 16, generator=g)
    gamma, beta = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.2
    return x, dy, gamma, beta


_BN_SPLIT = (3, 5)      # samples per rank: unequal, so the combination has to weight by the row counts


def _bn_standins(tb):
    """torch stand-ins of the four kernel wrappers behind the SyncBatchNorm path (same contracts)"""
    def stats(x, eps=1e-5):
        x2 = x.reshape(-1, x.shape[-1]).double()
        mean, var = x2.mean(0), x2.var(0, unbiased=False)
        return mean.float(), (1.0 / torch.sqrt(var + eps)).float(), var.float()

    def sums(x, dy, mean, rstd):
        xh = (x - mean) * rstd
        c = x.shape[-1]
        return (dy * xh).reshape(-1, c).sum(0), dy.reshape(-1, c).sum(0)

    tb.bn_stats = stats
    tb.bn_apply = lambda x, gamma, beta, mean, rstd: (x - mean) * rstd * gamma + beta
    tb.bn_backward_sums = sums
    tb.bn_backward_apply = lambda x, dy, gamma, mean, rstd, m_dyx, m_dy: gamma * rstd * (dy - m_dy - (x - mean) * rstd * m_dyx)


def _sync_bn_worker(rank, world, port, backend, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    esdist.init_process_group(backend)
    from efficientsam3_amd import train_blocks as tb
    dev = torch.device("cuda", rank) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(dev)
    else:
        _bn_standins(tb)
    try:
        x, dy, gamma, beta = _bn_case()
        lo = sum(_BN_SPLIT[:rank])
        xs, dys = x[lo:lo + _BN_SPLIT[rank]].contiguous().to(dev), dy[lo:lo + _BN_SPLIT[rank]].contiguous().to(dev)
        rm, rv = torch.zeros(16, device=dev), torch.ones(16, device=dev)
        tb.SYNC_BN = True
        y, mean, rstd = tb.bn_train_forward(xs, gamma.to(dev), beta.to(dev), rm, rv, 0.1, 1e-5)
        dx, dgamma, dbeta = tb.bn_train_backward(xs, dys, gamma.to(dev), mean, rstd)
        q.put((rank, [t.cpu().numpy() for t in (y, dx, dgamma, dbeta, rm, rv)]))
    finally:
        tb.SYNC_BN = None
        dist.destroy_process_group()


def _run_sync_bn(backend):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_bn_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the reference: ONE BatchNorm2d over the whole batch (what SyncBatchNorm is defined to equal), torch autograd
    x, dy, gamma, beta = _bn_case()
    xr, gr, br = x.permute(0, 3, 1, 2).contiguous().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(16), torch.ones(16)
    yr = torch.nn.functional.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    yr.backward(dy.permute(0, 3, 1, 2).contiguous())
    y_full, dx_full = yr.detach().permute(0, 2, 3, 1).numpy(), xr.grad.permute(0, 2, 3, 1).numpy()
    lo = 0
    dgamma_sum, dbeta_sum = 0.0, 0.0
    for rank in range(world):
        y, dx, dgamma, dbeta, rm_r, rv_r = results[rank]
        n = _BN_SPLIT[rank]
        np.testing.assert_allclose(y, y_full[lo:lo + n], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dx, dx_full[lo:lo + n], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(rm_r, rm.numpy(), rtol=1e-5, atol=1e-6)          # every rank holds the all-rank running statistics
        np.testing.assert_allclose(rv_r, rv.numpy(), rtol=1e-5, atol=1e-6)
        dgamma_sum, dbeta_sum = dgamma_sum + dgamma, dbeta_sum + dbeta
        lo += n
    np.testing.assert_allclose(dgamma_sum, gr.grad.numpy(), rtol=2e-4, atol=2e-4)    # the ranks' own sums add up to the whole batch's gradient
    np.testing.assert_allclose(dbeta_sum, br.grad.numpy(), rtol=2e-4, atol=2e-4)


def test_sync_batchnorm_two_ranks_equal_one_big_batch_gloo():
    """``Stage1Trainer(sync_bn=True)``'s BatchNorm on two ranks holding 3 and 5 samples = nn.BatchNorm2d over the 8 (forward, running statistics,
    input gradient; the weight / bias gradients are each rank's own sums, as torch.nn.SyncBatchNorm leaves them to the gradient all-reduce)"""
    _run_sync_bn("gloo")


@pytest.mark.gpu
def test_sync_batchnorm_two_ranks_equal_one_big_batch_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_sync_bn("nccl")


def _sync_bn_one_rank_worker(q):
    """one GPU, a one-rank RCCL group: the SyncBatchNorm path (split kernels + all_gather / all_reduce on device tensors) against the plain path
    (the fused two-call kernels) on the same tensors, fp32 and bf16"""
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    esdist.init_process_group("nccl", dev, force=True)
    from efficientsam3_amd import train_blocks as tb
    rep = {}
    try:
        for dtype in (torch.float32, torch.bfloat16):
            g = torch.Generator().manual_seed(11)
            x = (torch.randn(4, 33, 31, 48, generator=g) * 2.0 + 3.0).to(dtype).to(dev)
            dy = torch.randn(4, 33, 31, 48, generator=g).to(dtype).to(dev)
            gamma, beta = (torch.rand(48, generator=g) + 0.5).to(dev), torch.randn(48, generator=g).to(dev)
            outs = []
            for sync in (None, True):
                tb.SYNC_BN = sync
                rm, rv = torch.zeros(48, device=dev), torch.ones(48, device=dev)
                y, mean, rstd = tb.bn_train_forward(x, gamma, beta, rm, rv, 0.1, 1e-5)
                dx, dgamma, dbeta = tb.bn_train_backward(x, dy, gamma, mean, rstd)
                outs.append([t.float().cpu() for t in (y, mean, rstd, rm, rv, dx, dgamma, dbeta)])
            rep[str(dtype)] = [float((a - b).abs().max()) / max(1e-6, float(a.abs().max())) for a, b in zip(*outs)]
        q.put(rep)
    finally:
        tb.SYNC_BN = None
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sync_batchnorm_path_equals_the_plain_path_on_one_rank_rccl():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_sync_bn_one_rank_worker, args=(q,))
    p.start()
    rep = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    names = ("y", "mean", "rstd", "running_mean", "running_var", "dx", "dgamma", "dbeta")
    for dtype, errs in rep.items():
        print(f"[sync-bn one rank {dtype}] relative max differences to the plain path: " + ", ".join(f"{n} {e:.1e}" for n, e in zip(names, errs)))
        for n, e in zip(names, errs):
            assert e <= (1e-5 if n not in ("y", "dx") or "float32" in dtype else 8e-3), (dtype, n, e)


# ---- the stage-1 trainer on two ranks (DDP gradient averaging + SyncBatchNorm) = one rank with the whole batch -----------------------------------
class _Setter:
    """what the kernel stand-in installers need of pytest's monkeypatch, for a spawned worker"""

    @staticmethod
    def setattr(obj, name, value):
        setattr(obj, name, value)


def _install_trainer_standins():
    """every kernel wrapper of the trainer replaced by its torch stand-in (tests/test_*_host.py), the BatchNorm dispatcher of train_blocks left
    REAL (so that its SyncBatchNorm path runs) over stand-ins of the local kernels"""
    import types
    from efficientsam3_amd import train_blocks as tb
    from tests.test_stage1_trainer_host import install_host_trainer
    from tests.test_train_blocks_host import install_cpu_kernels
    from tests.test_train_repvit_host import install_repvit_kernels
    from tests.test_train_tinyvit_host import install_tinyvit_kernels
    real_fwd, real_bwd = tb.bn_train_forward, tb.bn_train_backward
    for install in (install_cpu_kernels, install_repvit_kernels, install_tinyvit_kernels, install_host_trainer):
        install(_Setter)
    tb._s1 = types.SimpleNamespace(bn_train_forward=tb.bn_train_forward, bn_train_backward=tb.bn_train_backward)
    tb.bn_train_forward, tb.bn_train_backward = real_fwd, real_bwd
    _bn_standins(tb)


def _trainer_case():
    from efficientsam3_amd import schema
    pre = "backbone.vision_backbone.trunk.model."
    sd = {k[len(pre):]: v.clone() for k, v in schema.synthetic_state_dict("repvit", "m0.9", seed=1).items() if k.startswith(pre)}
    g = torch.Generator().manual_seed(21)
    imgs = torch.randn(4, 3, 128, 96, generator=g)
    teacher = torch.randn(4, 8, 8, 1024, generator=g) * 0.5
    return sd, imgs, teacher


def _trainer_grads(sd, imgs, teacher, sync_bn):
    from efficientsam3_amd import stage1_train
    tr = stage1_train.Stage1Trainer(sd, "repvit_m0_9", embed_size=8, dtype="f32", device="cpu", lr=1e-3, cosine_weight=0.5, sync_bn=sync_bn)
    out = tr.step(imgs, teacher, [(128, 96)] * imgs.shape[0], update_grad=False)
    tr._allreduce()                                             # the bucketed averaging all-reduce (a no-op without a process group)
    state = tr.state_dict()
    return float(out["loss"]), {k: v.numpy() for k, v in tr.gradients().items()}, {k: v.numpy() for k, v in state.items() if k.endswith("running_var")}


def _trainer_worker(rank, world, port, sync_bn, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))   # two ranks on one host: no oversubscription of the cores
    esdist.init_process_group("gloo")
    try:
        _install_trainer_standins()
        sd, imgs, teacher = _trainer_case()
        lo = 2 * rank
        q.put((rank,) + _trainer_grads(sd, imgs[lo:lo + 2].contiguous(), teacher[lo:lo + 2].contiguous(), sync_bn))
    finally:
        from efficientsam3_amd import train_blocks as tb
        tb.SYNC_BN = None
        dist.destroy_process_group()


def _run_trainer(sync_bn):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, sync_bn, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r[0]: r[1:] for r in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def _one_rank_reference(q):
    _install_trainer_standins()
    sd, imgs, teacher = _trainer_case()
    q.put(_trainer_grads(sd, imgs, teacher, False))


def test_stage1_trainer_two_ranks_with_sync_bn_equal_one_rank_with_the_whole_batch():
    """``Stage1Trainer(sync_bn=True)`` on two gloo ranks with 2 samples each -- DistributedDataParallel's gradient averaging
    (``GradientAllReducer``) + torch.nn.SyncBatchNorm's statistics -- gives every rank the gradients of ONE rank training on the 4 samples: the
    identity the reference's ``--use-sync-bn`` multi-GPU run relies on (train_image_encoder_stage1.py:62-72).  Without SyncBatchNorm the two
    differ (the negative control).  Kernel stand-ins on the CPU; the host logic, the collectives and the BatchNorm dispatcher are the real ones."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_reference, args=(q,))
    p.start()
    loss1, grads1, rv1 = q.get(timeout=600)
    p.join(timeout=60)
    assert p.exitcode == 0
    synced = _run_trainer(True)
    gmax = max(float(np.abs(v).max()) for v in grads1.values())
    assert abs(0.5 * (synced[0][0] + synced[1][0]) - loss1) <= 1e-5 * abs(loss1)           # the mean of the ranks' losses is the batch's loss
    for rank in (0, 1):
        _, grads, rv = synced[rank]
        worst = max(float(np.abs(grads[k] - grads1[k]).max()) / (float(np.abs(grads1[k]).max()) + 1e-3 * gmax) for k in grads1)
        assert worst <= 2e-3, (rank, worst)
        for k in rv1:                                                                       # all-rank running statistics on every rank
            np.testing.assert_allclose(rv[k], rv1[k], rtol=1e-4, atol=1e-6)
    plain = _run_trainer(False)
    worst_plain = max(float(np.abs(plain[0][1][k] - grads1[k]).max()) / (float(np.abs(grads1[k]).max()) + 1e-3 * gmax) for k in grads1)
    assert worst_plain > 2e-2, worst_plain                                                  # per-rank statistics are a different function


# ---- the in-backward push path of the gradient all-reduce ---------------------------------------------------------------
def _trainer_updates(sd, imgs, teacher, sync_bn, accumulation_steps, n_updates, bucket_bytes):
    """`n_updates` updating iterations (each `accumulation_steps` micro-steps on the same samples); returns the gradients the LAST update
    consumed, the parameters after it, and what the reducer did on that last update"""
    from efficientsam3_amd import stage1_train
    tr = stage1_train.Stage1Trainer(sd, "repvit_m0_9", embed_size=8, dtype="f32", device="cpu", lr=1e-3, cosine_weight=0.5, sync_bn=sync_bn,
                                    accumulation_steps=accumulation_steps, bucket_bytes=bucket_bytes)
    sizes = [(128, 96)] * imgs.shape[0]
    pushed, snap = [], {}
    orig_step = tr.updater.step

    def step_and_snapshot(*a, **k):     # the arena is zeroed by the update: keep what the update consumed (after the all-reduce)
        snap.update({n: tr.updater.grad(n).detach().cpu().clone().numpy() for n in tr.names})
        return orig_step(*a, **k)

    tr.updater.step = step_and_snapshot
    for _ in range(n_updates):
        for m in range(accumulation_steps):
            tr.step(imgs, teacher, sizes, update_grad=(m == accumulation_steps - 1))
        pushed.append(bool(tr._pushed))
    red = tr.reducer
    info = None if red is None else dict(n_buckets=red.n_buckets, issue_order=list(red.issue_order), pushed=pushed,
                                         first_buckets=[[tr._arrival[i] for i in b_["idx"]] for b_ in red.buckets[:2]])
    return (dict(snap), {k: v.numpy() for k, v in tr.state_dict().items()}, info)


def _updates_worker(rank, world, port, acc, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))   # two ranks on one host: no oversubscription of the cores
    esdist.init_process_group("gloo")
    try:
        _install_trainer_standins()
        sd, imgs, teacher = _trainer_case()
        lo = 2 * rank
        q.put((rank,) + _trainer_updates(sd, imgs[lo:lo + 2].contiguous(), teacher[lo:lo + 2].contiguous(), True, acc, 2, 4 << 20))
    finally:
        from efficientsam3_amd import train_blocks as tb
        tb.SYNC_BN = None
        dist.destroy_process_group()


def _updates_one_rank(acc, q):
    _install_trainer_standins()
    sd, imgs, teacher = _trainer_case()
    q.put(_trainer_updates(sd, imgs, teacher, False, acc, 2, 4 << 20))


@pytest.mark.parametrize("acc", [1, 2])
def test_stage1_trainer_two_ranks_push_gradients_during_backward(acc):
    """Two UPDATING iterations on two gloo ranks (SyncBatchNorm on): the first learns the arrival order of the gradients and reduces after
    its backward pass, the second pushes every gradient into the bucketed all-reduce from INSIDE the backward pass (`_pushed`), head
    bucket first -- with the trainer's DDP-sized buckets the head's 3x3 weight (37.7 MB) closes bucket 0 on its own and goes on the wire
    while the trunk is still in its backward pass.  Gradients and parameters of the second update equal ONE rank training on all four
    samples; with ACCUMULATION_STEPS = 2 only the updating micro-step pushes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_updates_one_rank, args=(acc, q))
    p.start()
    grads1, params1, info1 = q.get(timeout=900)
    p.join(timeout=60)
    assert p.exitcode == 0 and info1 is None
    world = 2
    port = _free_port()
    procs = [ctx.Process(target=_updates_worker, args=(r, world, port, acc, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = {r[0]: r[1:] for r in (q.get(timeout=900) for _ in range(world))}
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    gmax = max(float(np.abs(v).max()) for v in grads1.values())
    for rank in (0, 1):
        grads, params, info = results[rank]
        assert info["pushed"] == [False, True], info["pushed"]                       # order learnt in update 1, pushed in update 2
        assert info["n_buckets"] >= 3 and info["issue_order"] == list(range(info["n_buckets"])), info
        big = [b_ for b_ in info["first_buckets"] if "head.3.weight" in b_]
        assert big and all(n.startswith("head.") for n in big[0]) and len(big[0]) <= 3, info["first_buckets"]   # the head's big weight does not wait for the trunk
        worst = max(float(np.abs(grads[k] - grads1[k]).max()) / (float(np.abs(grads1[k]).max()) + 1e-3 * gmax) for k in grads1)
        assert worst <= 5e-3, (rank, worst)
        moved = [k for k in params1 if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
        far = sum(int((np.abs(params[k] - params1[k]) > 2.5e-3).sum()) for k in moved)   # 2 updates x lr 1e-3: no element may be further
        assert far == 0, far
