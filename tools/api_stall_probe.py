"""Development aid: where the occasional +60-80 ms of an API-level step (Sam3Processor.set_image_batch + predict_inst_batch, 32 PIL
images) comes from: per iteration the host time of each call and the DEVICE time between stream events around them."""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientsam3_amd import Sam3Processor, build_efficientsam3_image_model, schema, synth  # noqa: E402

sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0)
model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=True, backbone_type="efficientvit", model_name="b1",
                                        dtype="bf16", state_dict=sd)
B = 32
rng_img = np.random.default_rng(0).integers(0, 256, (4, 1024, 1024, 3), dtype=np.uint8)
pil = [Image.fromarray(rng_img[i % 4]) for i in range(B)]
proc = Sam3Processor(model)
pts, labels, boxes = synth.prompts(B, seed=2)
sx = 1024.0 / 1008.0
pcs, bxs, lbl = [pts[i] * sx for i in range(B)], [boxes[i] * sx for i in range(B)], [labels[i] for i in range(B)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
# time the pieces of predict_inst_batch (host clocks; a piece that waits for the device shows the wait)
acc = {}


def timed(obj, name):
    fn = getattr(obj, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + 1e3 * (time.perf_counter() - t)
        return r
    setattr(obj, name, wrapper)


for nm in ("_masks_to_host", "_decode", "_prep_prompts"):
    timed(model, nm)
for nm in ("postprocess", "clamp_", "decode", "encode"):
    timed(model.engine, nm)
import gc
gc_t = [0.0]
gc.callbacks.append(lambda phase, info: gc_t.__setitem__(0, time.perf_counter()) if phase == "start" else acc.__setitem__("gc", acc.get("gc", 0.0) + 1e3 * (time.perf_counter() - gc_t[0])))
mode = sys.argv[1] if len(sys.argv) > 1 else "api"
u8 = torch.from_numpy(np.stack([np.asarray(im.convert("RGBA")) for im in pil])).pin_memory() if mode == "h2d" else None
for rep in range(14):
    t0 = time.perf_counter()
    ev[0].record()
    if mode == "h2d":      # only the pinned host -> device copies of the two halves (the same bytes set_image_batch moves)
        a = u8[:16].to("cuda", non_blocking=True); b = u8[16:].to("cuda", non_blocking=True)
        ev[1].record(); ev[2].record()
        t1 = t2 = time.perf_counter()
    else:
        st = proc.set_image_batch(pil)
        t1 = time.perf_counter()
        ev[1].record()
        if mode == "api":
            out = model.predict_inst_batch(st, point_coords_batch=pcs, point_labels_batch=lbl, box_batch=bxs, multimask_output=False)
        t2 = time.perf_counter()
        ev[2].record()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    ms_ = torch.cuda.memory_stats()
    acc["dev_allocs"] = float(ms_.get("num_device_alloc", 0))
    acc["dev_frees"] = float(ms_.get("num_device_free", 0))
    pieces = " ".join(f"{k} {v:.1f}" for k, v in acc.items() if v >= 0.05)
    acc.clear()
    print(f"{mode} {rep:2d}: [{pieces}] host set_image {1e3*(t1-t0):6.1f} predict {1e3*(t2-t1):6.1f} sync {1e3*(t3-t2):6.1f} | device: set_image {ev[0].elapsed_time(ev[1]):6.1f} "
          f"predict {ev[1].elapsed_time(ev[2]):6.1f} | total {1e3*(t3-t0):6.1f} ms")
