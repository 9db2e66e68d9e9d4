"""Where an API-level step (Sam3Processor.set_image_batch + model.predict_inst_batch on 32 PIL images) spends its time.
Development aid; run on an MI355X:  python tools/api_level_probe.py"""
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
pcs = [pts[i] * sx for i in range(B)]
bxs = [boxes[i] * sx for i in range(B)]
lbl = [labels[i] for i in range(B)]


def step():
    st = proc.set_image_batch(pil)
    return model.predict_inst_batch(st, point_coords_batch=pcs, point_labels_batch=lbl, box_batch=bxs, multimask_output=False)


for _ in range(3):
    out = step()
torch.cuda.synchronize()
# (a) results dropped every step (a loop that consumes and forgets), (b) results kept (every step allocates fresh arrays)
for keep in (False, True):
    kept = []
    t0 = time.perf_counter()
    for _ in range(5):
        out = step()
        if keep:
            kept.append(out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"api step, results {'kept' if keep else 'dropped'}: {dt * 1e3:.1f} ms = {B / dt:.0f} img/s")
    del kept
for tc in (True, False):
    proc.rgbx_torch_copy = tc
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"api step, rgbx_torch_copy={tc}: {dt * 1e3:.1f} ms = {B / dt:.0f} img/s")
proc.rgbx_torch_copy = False
for _ in range(2):
    step()
torch.cuda.synchronize()
for rep in range(3):   # the two calls of a step, each fully synchronised
    t0 = time.perf_counter(); st = proc.set_image_batch(pil); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    out = model.predict_inst_batch(st, point_coords_batch=pcs, point_labels_batch=lbl, box_batch=bxs, multimask_output=False); t3 = time.perf_counter()
    print(f"step {rep}: set_image_batch host {1e3*(t1-t0):.1f} + device tail {1e3*(t2-t1):.1f} | predict_inst_batch {1e3*(t3-t2):.1f}  (ms)")
    del out
for mode in ("keep previous", "drop before next"):
    ts = []
    out = None
    for rep in range(8):
        t0 = time.perf_counter()
        if mode == "drop before next":
            out = None
        out = step()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"8 api steps ({mode}):", " ".join(f"{t:.1f}" for t in ts), "ms")
print("host cpus:", os.cpu_count(), "torch threads:", torch.get_num_threads())
# pieces
for rep in range(2):
    t0 = time.perf_counter(); proc._stage_pil_batch(pil, rgbx=True); t1 = time.perf_counter()
    st = proc.set_image_batch(pil); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    sam2 = model._check_state(st)
    per = [model._prep_prompts(pcs[i], lbl[i], bxs[i], True, (1024, 1024)) for i in range(B)]; t4 = time.perf_counter()
    coords = np.concatenate([c for c, _ in per]); labs = np.concatenate([l for _, l in per])
    low, iou = model._decode(sam2, coords, labs, np.arange(B, dtype=np.int32), False, None); torch.cuda.synchronize(); t5 = time.perf_counter()
    m = model.engine.postprocess(low.view(B, 1, 288, 288), (1024, 1024), False); torch.cuda.synchronize(); t6 = time.perf_counter()
    h = model._masks_to_host(m); t7 = time.perf_counter()
    l_np, i_np = low.cpu().numpy(), iou.cpu().numpy(); t8 = time.perf_counter()
    print(f"rep {rep}: stage_pil {1e3*(t1-t0):.1f} | set_image_batch {1e3*(t2-t1):.1f} (+{1e3*(t3-t2):.1f} GPU) | prep {1e3*(t4-t3):.1f} | decode {1e3*(t5-t4):.1f} | "
          f"postprocess {1e3*(t6-t5):.1f} | masks_to_host {1e3*(t7-t6):.1f} | low/iou D2H {1e3*(t8-t7):.1f}  (ms)")
    del h
# staging variants on this box's host: Pillow's 4-byte pixels by torch copies / by np.copyto from the pool; packed 3-byte pixels
for name, kw in (("rgbx torch copy_", dict(rgbx=True, torch_copy=True)), ("rgbx np.copyto pool", dict(rgbx=True, torch_copy=False)),
                 ("rgb tobytes pool", dict(rgbx=False, torch_copy=True))):
    proc.rgbx_torch_copy = kw["torch_copy"]
    proc._stage_pil_batch(pil, rgbx=kw["rgbx"])
    t0 = time.perf_counter()
    for _ in range(5):
        proc._stage_pil_batch(pil, rgbx=kw["rgbx"])
    print(f"stage 32 PIL images, {name}: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
