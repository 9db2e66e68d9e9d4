"""ORACLE tooling (test infrastructure only): the reference's chunked multi-GPU detector entry as a fixture.

Runs the REAL `Sam3ImageOnVideoMultiGPU.forward_video_grounding_multigpu` / `_build_multigpu_buffer_next_chunk` /
`_gather_tensor` (sam3/sam3/model/sam3_image.py:701-883) on TWO gloo ranks (CPU) around a stub detector
(`forward_grounding` replaced by a deterministic function of the frame index: the detector itself is pinned elsewhere,
tests/golden/pcs_ev_m), and records per rank and per call

  * on which frame the rank ran the detector (the chunk order and the round-robin frame assignment),
  * the keys the call returned and a float64 checksum + shape + dtype of every tensor,
  * how many frames the buffer held after the call (the previous chunk is dropped, the next one built ahead),

for three schedules (5 and 4 frames forward, 5 frames in reverse tracking order).
`tests/test_dist_gloo.py::test_video_grounding_matches_reference_trace` drives `efficientsam3_amd.dist.VideoGroundingMultiGPU`
with the same stub on two gloo ranks and requires the same trace.

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_video_grounding.py

Output: tests/golden/video_grounding/trace.json
"""
from __future__ import annotations

import json
import multiprocessing as mp
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCHEDULES = [(5, False), (4, False), (5, True)]
WORLD = 2


def detector_out(frame: int) -> dict:
    """the stub detector's outputs on frame `frame` (the same function lives in tests/test_dist_gloo.py)"""
    g = torch.Generator().manual_seed(77 + frame)
    return {"pred_logits": torch.randn((1, 6, 1), generator=g), "pred_boxes": torch.rand((1, 6, 4), generator=g),
            "pred_boxes_xyxy": torch.rand((1, 6, 4), generator=g), "pred_masks": torch.randn((1, 6, 8, 8), generator=g),
            "extra_key_not_gathered": torch.zeros(1)}


def fpn(frame: int):
    g = torch.Generator().manual_seed(900 + frame)
    return [torch.randn((1, c, s, s), generator=g) for c, s in ((4, 8), (8, 4), (16, 2))]


def describe(x):
    if torch.is_tensor(x):
        return {"shape": list(x.shape), "dtype": str(x.dtype).replace("torch.", ""), "sum": float(x.double().sum()),
                "abs_sum": float(x.double().abs().sum())}
    return {"value": x}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from sam3.model.sam3_image import Sam3ImageOnVideoMultiGPU  # the REAL reference class
    try:
        traces = {}
        for num_frames, reverse in SCHEDULES:
            m = object.__new__(Sam3ImageOnVideoMultiGPU)   # no model is built: only the bookkeeping under test runs
            calls = []

            def forward_grounding(backbone_out=None, find_input=None, find_target=None, geometric_prompt=None):
                calls.append(int(find_input))
                out = detector_out(int(find_input))
                out["prev_encoder_out"] = {"backbone_out": {"sam2_backbone_out": {"backbone_fpn": fpn(int(find_input)),
                                                                                   "vision_pos_enc": "pos"}}}
                return out

            for k, v in dict(rank=rank, world_size=WORLD, async_all_gather=True, gather_backbone_out=True,
                             forward_grounding=forward_grounding).items():
                object.__setattr__(m, k, v)
            buf, per_call = {}, []
            order = range(num_frames - 1, -1, -1) if reverse else range(num_frames)
            for t in order:
                n0 = len(calls)
                out, _ = m.forward_video_grounding_multigpu(
                    backbone_out=None, find_inputs=list(range(num_frames)), geometric_prompt=None, frame_idx=t,
                    num_frames=num_frames, multigpu_buffer=buf, track_in_reverse=reverse, return_sam2_backbone_feats=True)
                per_call.append({"frame": t, "detector_ran_on": calls[n0:], "buffered_frames": sorted(buf),
                                 "out": {k: describe(v) for k, v in sorted(out.items())}})
            traces[f"{num_frames}_{'reverse' if reverse else 'forward'}"] = per_call
        q.put((rank, traces))
    finally:
        dist.destroy_process_group()


def main():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out = {"world_size": WORLD, "reference": "sam3/sam3/model/sam3_image.py:701-883 (forward_video_grounding_multigpu, "
           "_build_multigpu_buffer_next_chunk, _gather_tensor) run on gloo with a stub forward_grounding",
           "torch": torch.__version__, "ranks": {str(r): res[r] for r in sorted(res)}}
    d = os.path.join(ROOT, "tests", "golden", "video_grounding")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "trace.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for r in sorted(res):
        for name, calls in res[r].items():
            print(f"rank {r} {name}: detector frames {[c['detector_ran_on'] for c in calls]}  buffer {[c['buffered_frames'] for c in calls]}")


if __name__ == "__main__":
    main()
