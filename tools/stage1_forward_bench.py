"""BASELINE config 5, forward half, as a measurement: ViT-H teacher trunk + EV-M student trunk on the same batch and
the distillation loss between the two embeddings (efficientsam3_amd.stage1.paired_forward).  Not bench.py's metric;
run it by hand on an MI355X:

    python tools/stage1_forward_bench.py [--batch 8] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientsam3_amd import build_efficientsam3_image_model, build_sam3_image_model, schema, stage1, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    teacher = build_sam3_image_model(device="cuda", enable_inst_interactivity=False, dtype="bf16",
                                     state_dict=schema.synthetic_state_dict("sam3", "vit_h", seed=0, enable_inst_interactivity=False))
    student = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit",
                                              model_name="b1", dtype="bf16",
                                              state_dict=schema.synthetic_state_dict("efficientvit", "b1", seed=0,
                                                                                     enable_inst_interactivity=False))
    base = [synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s)) for s in (1, 3)]
    x = torch.from_numpy(np.stack([base[i % 2] for i in range(args.batch)])).cuda()
    sizes = [(1008, 1008)] * args.batch
    for _ in range(args.warmup):
        out = stage1.paired_forward(teacher, student, x, sizes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = stage1.paired_forward(teacher, student, x, sizes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "images/sec stage-1 paired forward (ViT-H teacher trunk + EV-M student trunk + loss, bf16)",
                      "value": round(args.batch / dt, 2), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3),
                      "batch": args.batch, "steps": args.steps, "mse": float(out["mse"]), "cosine": float(out["cosine"])}))


if __name__ == "__main__":
    main()
