"""A/B of the API-level step's host pipeline in ONE process (box-to-box spread is larger than the effects): first encode chunk
(half / quarter of the batch) x device-to-host mask copy in 1 / 4 pieces.  (Round 5 also tried the second chunk's staged H2D in pieces of 8 images, each sent as soon as it was staged: no difference, 15.8 vs 15.8 ms -- profiles/r05/api_ab_h2d_pieces.txt.)   python tools/api_ab.py"""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientsam3_amd import Sam3Processor, build_efficientsam3_image_model, schema, synth  # noqa: E402
from efficientsam3_amd import sam3_image as SI  # noqa: E402

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


def step():
    st = proc.set_image_batch(pil)
    return model.predict_inst_batch(st, point_coords_batch=pcs, point_labels_batch=lbl, box_batch=bxs, multimask_output=False)


ref = None
for rnd in range(2):
    for frac in (0.5, 0.25):
        for chunks, piece in ((1, 0), (4, 0)):
            proc.first_chunk_fraction, SI.D2H_CHUNKS = frac, chunks
            for _ in range(3):
                out = step()
            torch.cuda.synchronize()
            if ref is None:
                ref = [m.copy() for m in out[0]]
            assert all(np.array_equal(a, b) for a, b in zip(ref, out[0])), "masks changed with the pipeline setting"
            t0 = time.perf_counter()
            for _ in range(8):
                out = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 8
            print(f"round {rnd}: first chunk {frac:5.3f}, D2H pieces {chunks}, H2D piece {piece}: {dt * 1e3:6.2f} ms = {B / dt:5.0f} images/s", flush=True)
