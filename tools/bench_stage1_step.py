"""Stage-1 training step on one MI355X: a student (backbone + head; default EfficientViT-B1, --model b0 | b1 | b2 | repvit_m0_9 | repvit_m1_1 |
repvit_m2_3 | tiny_vit_5m | tiny_vit_11m | tiny_vit_21m) at 1008^2, synthetic images / teacher embeddings,
forward + masked MSE / cosine loss + backward + clip + AdamW (efficientsam3_amd.stage1_train.Stage1Trainer; the step the reference runs in
stage1/train_image_encoder_stage1.py:165-226).  Prints ONE JSON line; the roofline leg prices the whole step's algorithmic FLOPs
(forward graph of SURVEY.md 8(d): backbone 20.3 + head 19.9 GFLOP / image; a training step is forward + input gradients + weight
gradients = 3 x) against the dense bf16 MFMA peak.

For the other students the forward FLOPs are COUNTED from the layers after a forward pass (2 x output elements x weight elements per output
channel of every convolution of trunk and head; the SqueezeExcite MLPs, LiteMLA's attention products and, for TinyViT, the Linear layers
and attention products are left out: its figure is a lower bound).

    python tools/bench_stage1_step.py [--batch 8] [--steps 5] [--warmup 2] [--dtype bf16|f32] [--model b1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientsam3_amd import schema  # noqa: E402
from efficientsam3_amd.stage1_train import Stage1Trainer  # noqa: E402

PREFIX = "backbone.vision_backbone.trunk.model."
FWD_GFLOP_PER_IMAGE = 20.3 + 19.9
PEAK_BF16_TFLOPS = 2500.0


def counted_forward_flops(obj, seen=None) -> float:
    """2 x conv_out elements x (weight elements / output channels) over every layer object below ``obj`` that kept its weight ``w`` and its
    ``conv_out`` from the last forward (ConvLayerTrain, StemConvTrain, Conv3x3S2Train)"""
    seen = set() if seen is None else seen
    if id(obj) in seen:
        return 0.0
    seen.add(id(obj))
    total = 0.0
    w, out = getattr(obj, "w", None), getattr(obj, "conv_out", None)
    if torch.is_tensor(w) and torch.is_tensor(out):
        total += 2.0 * out.numel() * (w.numel() / w.shape[0])
    if isinstance(obj, (list, tuple)):
        children = list(obj)
    elif isinstance(obj, dict):
        children = list(obj.values())
    elif type(obj).__module__.startswith("efficientsam3_amd"):
        children = list(vars(obj).values())
    else:
        children = []
    for c in children:
        if isinstance(c, (list, tuple, dict)) or type(c).__module__.startswith("efficientsam3_amd"):
            total += counted_forward_flops(c, seen)
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="b1", choices=["b0", "b1", "b2", "repvit_m0_9", "repvit_m1_1", "repvit_m2_3", "tiny_vit_5m", "tiny_vit_11m", "tiny_vit_21m"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    a = ap.parse_args()
    if a.model.startswith("repvit_"):
        family, name = "repvit", a.model[len("repvit_"):].replace("_", ".")
    elif a.model.startswith("tiny_vit_"):
        family, name = "tinyvit", a.model[len("tiny_vit_"):]
    else:
        family, name = "efficientvit", a.model
    sd = schema.synthetic_state_dict(family, name, seed=0)
    sd = {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}
    tr = Stage1Trainer(sd, a.model, embed_size=72, dtype=a.dtype, lr=1e-4, weight_decay=0.05, clip_grad=5.0, cosine_weight=0.5)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn((a.batch, 3, 1008, 1008), generator=g).cuda()
    teacher = (torch.randn((a.batch, 72, 72, 1024), generator=g) * 0.5).to("cuda", torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    sizes = [(1008, 1008)] * a.batch
    for _ in range(a.warmup):
        tr.step(imgs, teacher, sizes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.step(imgs, teacher, sizes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    counted = counted_forward_flops(tr.trunk) + counted_forward_flops(tr.head.l0)
    hw3 = tr.head.hw[0] * tr.head.hw[1]
    counted += 2.0 * a.batch * hw3 * tr.head.w3.numel()          # the head's 3x3 conv
    fwd = FWD_GFLOP_PER_IMAGE * 1e9 * a.batch if a.model == "b1" else counted
    flops = 3.0 * fwd
    print(json.dumps({"metric": f"images/sec stage-1 distillation training step @1008^2 ({a.model} student)", "value": round(a.batch / dt, 2),
                      "unit": "images/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt * 1e3, 2), "dtype": a.dtype,
                      "data": "synthetic", "config": {"workload": "forward + loss + backward + clip + AdamW, device-resident fp32 master weights",
                                                       "batch": a.batch, "model": a.model, "loss": float(out["loss"]), "grad_norm": float(out["grad_norm"]),
                                                       "forward_gflop_per_image": round(fwd / a.batch / 1e9, 2),
                                                       "counted_conv_gflop_per_image": round(counted / a.batch / 1e9, 2)},
                      "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(flops / dt / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": None,
                                   "note": "whole step: 3 x the forward graph's algorithmic FLOPs / wall time of a step"}}))


if __name__ == "__main__":
    main()
