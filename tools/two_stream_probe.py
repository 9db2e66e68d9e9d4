"""Development probe (round 6): do TWO batches in flight on two HIP streams beat one, when the power-bound GEMM launches leave CUs free?

A power-bound launch loses little when its persistent grid is capped (profiles/r06/gemm_grid_cap.txt: the dominant launch takes +4.5 % on
224 CUs, +11 % on 192, +33 % on 128), so the CUs it leaves can run another batch's latency-bound kernels.  Two engine replicas, one Python
thread and one stream each, run the bench's step (encode + decode + post-processing, fixed buffers); ESAM3_P_GRID (dev library) caps the
GEMM grid.  Prints images/s for one stream and for two.

    ESAM3_LIB=build_dev/libesam3_dev.so ESAM3_P_GRID=192 python tools/two_stream_probe.py
"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientsam3_amd import _lib  # noqa: E402

if os.environ.get("ESAM3_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["ESAM3_LIB"])
from efficientsam3_amd import build_efficientsam3_image_model, schema, synth  # noqa: E402

B, STEPS = 32, 12
dev = torch.device("cuda", 0)
sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0)


def make():
    m = build_efficientsam3_image_model(device=dev, enable_inst_interactivity=True, backbone_type="efficientvit", model_name="b1", dtype="bf16",
                                        state_dict=sd)
    base = [synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=1)), synth.normalise_to_chw_f32(synth.noise_image_u8(seed=2))]
    x = torch.from_numpy(np.stack([base[i % 2] for i in range(B)])).to(dev)
    pts, labels, boxes = synth.prompts(B, seed=2)
    coords, labs = m._prep_prompts(pts, labels, boxes, True, (1008, 1008))
    c_d, l_d = torch.from_numpy(coords).to(dev), torch.from_numpy(labs).to(dev)
    pi_d = torch.arange(B, dtype=torch.int32, device=dev)
    bufs = {"enc": None, "dec": None, "post": None}
    eng = m.engine

    def step():
        bufs["enc"] = out = eng.encode(x, want_sam3=True, want_sam2=True, out=bufs["enc"])
        bufs["dec"] = low, iou = eng.decode(out["sam2_fpn"], pi_d, c_d, l_d, multimask_output=False, out=bufs["dec"])
        bufs["post"] = eng.postprocess(low, (1008, 1008), return_logits=False, out=bufs["post"])
        return bufs["post"]

    return m, step


def run(n_streams):
    reps = [make() for _ in range(n_streams)]
    gate = threading.Barrier(n_streams + 1)
    sums = [None] * n_streams

    def worker(i):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            step = reps[i][1]
            for _ in range(3):
                step()
            torch.cuda.current_stream().synchronize()
            gate.wait()
            for _ in range(STEPS):
                out = step()
            torch.cuda.current_stream().synchronize()
            gate.wait()
            sums[i] = float(out.float().mean())

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(n_streams)]
    for t in ths:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    dt = time.perf_counter() - t0
    for t in ths:
        t.join()
    return n_streams * STEPS * B / dt, sums


for n in (1, 2, 1, 2):
    ips, sums = run(n)
    print(f"P_GRID={os.environ.get('ESAM3_P_GRID', '-')} streams={n}: {ips:.0f} img/s  (mask means {sums})", flush=True)
