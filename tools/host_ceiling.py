"""The host side of an API-level step under N concurrent ranks, with the device step left out ("Next" 6; SURVEY.md
8(e) names host IO, not xGMI, as the risk for >= 6 x scaling at 8 GPUs -- and no 8-GPU node is available to measure it on).

Every rank (one process, CPUs = its 1 / N share of the host, as bench.py binds a rank to the CPUs next to its GPU) repeats what
`Sam3Processor.set_image_batch` + `Sam3Image.predict_inst_batch` do on the HOST for a batch of 32 PIL images of 1024 x 1024:

    stage      two half batches of Pillow pixels -> pinned staging buffers        (Sam3Processor._stage_pil_batch, the product code)
    prompts    32 x coordinate transforms                                         (Sam3Image._prep_prompts, the product code)
    widen      32 x 1 x 1024 x 1024 uint8 masks in a pinned buffer -> float32       (sam3_image._widen_into, the product code)

The device work between them (H2D, encode, decode, D2H: 10.3 - 10.6 ms per step on one MI355X) is NOT run: the question is whether the
host part of a rank still fits beside it when 8 ranks share the host's memory system.  Prints per-rank host milliseconds per step at
N = 1 and N = `--ranks`, and the host-side ceiling in images/s per rank.

    python tools/host_ceiling.py [--ranks 8] [--steps 30]"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, steps, q, go):
    cpus = sorted(os.sched_getaffinity(0))
    share = cpus[rank * len(cpus) // world:(rank + 1) * len(cpus) // world]
    os.sched_setaffinity(0, share)
    import torch
    from PIL import Image
    from efficientsam3_amd import sam3_image as SI
    from efficientsam3_amd.sam3_image_processor import Sam3Processor

    class _Eng:   # just enough of an engine for the processor's constructor and the prompt transforms
        torch_dtype = torch.bfloat16

        def preprocess_resize_u8_batch(self, *a):
            raise RuntimeError("device step is not part of this measurement")

    class _Model:
        engine = _Eng()
        device = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
        dual_neck, inst_interactive_predictor = True, object()
        image_size = 1008

    model = _Model()
    proc = Sam3Processor.__new__(Sam3Processor)
    Sam3Processor.__init__(proc, model) if False else None   # the constructor wants a full model: set the fields the staging path reads
    proc.model, proc.device, proc.resolution = model, model.device, 1008
    proc._stage, proc._stage_busy, proc._pool, proc.rgbx_torch_copy = {}, {}, None, False
    rng = np.random.default_rng(rank)
    base = rng.integers(0, 256, (4, 1024, 1024, 3), dtype=np.uint8)
    pil = [Image.fromarray(base[i % 4]) for i in range(32)]
    pin = torch.empty((32, 1, 1024, 1024), dtype=torch.uint8)
    pin = pin.pin_memory() if torch.cuda.is_available() else pin
    pin.random_(0, 2)
    outs = [np.empty((32, 1, 1024, 1024), np.float32) for _ in range(2)]
    for o in outs:
        o.fill(0.0)        # touch the pages once, as the product's result pool has after its first step
    pts = rng.uniform(100, 900, (32, 1, 2)).astype(np.float32)
    labels = np.ones((32, 1), np.int32)
    boxes = np.concatenate([pts[:, 0] - 50, pts[:, 0] + 50], axis=1).astype(np.float32)
    prep = SI.Sam3Image._prep_prompts

    def step(k):
        t0 = time.perf_counter()
        proc._stage_pil_batch(pil[:16], slot=0, rgbx=True)
        proc._stage_pil_batch(pil[16:], slot=1, rgbx=True)
        t1 = time.perf_counter()
        for i in range(32):
            prep(pts[i], labels[i], boxes[i], True, (1024, 1024))
        t2 = time.perf_counter()
        SI._widen_into(outs[k & 1], pin.numpy())
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    for k in range(5):
        step(k)
    q.put(("ready", rank))
    go.wait()
    acc = np.zeros(3)
    t0 = time.perf_counter()
    for k in range(steps):
        acc += step(k)
    wall = time.perf_counter() - t0
    q.put(("done", rank, wall / steps * 1e3, (acc / steps * 1e3).tolist(), len(share)))


def run(world, steps):
    ctx = mp.get_context("spawn")
    q, go = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=worker, args=(r, world, steps, q, go)) for r in range(world)]
    for p in ps:
        p.start()
    ready = 0
    while ready < world:
        if q.get(timeout=600)[0] == "ready":
            ready += 1
    go.set()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
    return sorted(res, key=lambda r: r[1])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    print(f"host: {len(os.sched_getaffinity(0))} cpus; batch 32 x 1024 x 1024 PIL images per rank and step; device step (not run) = 10.3 - 10.6 ms")
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            print(f"  {f}: {open(f).read().strip()}")     # a CPU quota below ranks x 16 worker threads throttles the N-rank run
        except OSError:
            pass
    try:
        import subprocess
        print("  " + subprocess.run(["lscpu"], capture_output=True, text=True).stdout.replace("\n", " | ")[:600])
    except Exception:  # noqa: BLE001
        pass
    for world in (1, a.ranks):
        res = run(world, a.steps)
        worst = max(r[2] for r in res)
        print(f"{world} rank(s) x {res[0][4]} cpus: host ms per step, slowest rank {worst:.2f}  (stage / prompts / widen of rank 0: "
              f"{res[0][3][0]:.2f} / {res[0][3][1]:.2f} / {res[0][3][2]:.2f}); per rank: " + " ".join(f"{r[2]:.2f}" for r in res))
        print(f"   host-side ceiling per rank = {32e3 / worst:.0f} images/s; {world} ranks = {world * 32e3 / worst:.0f} images/s "
              f"(device-bound rate per rank ~ 3050 images/s)")
