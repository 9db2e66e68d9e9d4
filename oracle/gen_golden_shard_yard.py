"""ORACLE tooling (test infrastructure only): the reference's own bf16 behaviour on the INPUTS OF THE FULL-SHARD TESTS.

tests/test_students_gpu.py::test_tinyvit_full_shard_32_is_image_independent and tests/test_e2e_gpu.py::test_full_batch_32_is_image_independent
compare four distinct (image, point + box prompt) pairs with the fp32 oracle: two smooth synthetic images and two uniform-noise images
(`efficientsam3_amd/synth.py`, seeds 1-4; prompts `synth.prompts(4, seed=2)`).  The per-model yardstick (bf16ref_manifest / bf16ref_draws) was
measured on smooth images; a noise image's mask is speckle and moves more under any change of precision.  This script measures, for exactly
those four inputs, how far the REAL reference under `torch.autocast("cpu", bfloat16)` is from its own fp32 run (low-res logits, IoU head,
thresholded-mask IoU), so that the shard tests can hold every image to 1.5 x ITS OWN distance instead of a flat floor.

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_shard_yard.py [--backbone tinyvit --model 11m]

Output: tests/golden[/<backbone>_<model>]/shard_yard.json
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from efficientsam3_amd import schema, synth  # noqa: E402
from oracle import gen_golden as G  # noqa: E402
from oracle.gen_golden_bf16ref import mask_iou  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="tinyvit")
    ap.add_argument("--model", default="11m")
    args = ap.parse_args()
    default = (args.backbone, args.model) == ("efficientvit", "b1")
    gold = G.GOLD if default else os.path.join(G.GOLD, f"{args.backbone}_{args.model}")
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    from sam3 import build_efficientsam3_image_model  # the REAL reference
    from sam3.model.sam3_image_processor import Sam3Processor
    model = build_efficientsam3_image_model(device="cpu", checkpoint_path=None, load_from_HF=False, enable_inst_interactivity=True,
                                            backbone_type=args.backbone, model_name=args.model, text_encoder_type="MobileCLIP-S0",
                                            text_encoder_context_length=16)
    sd = schema.synthetic_state_dict(args.backbone, args.model, seed=0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected[:5]
    model.eval()
    proc = Sam3Processor(model, device="cpu")
    images = [synth.smooth_image_u8(seed=1), synth.noise_image_u8(seed=2), synth.smooth_image_u8(seed=3), synth.noise_image_u8(seed=4)]
    pts, labels, boxes = synth.prompts(4, seed=2)

    def run(i, amp):
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if amp else torch.autocast("cpu", enabled=False)
        chw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(images[i], -1, 0)))
        with torch.inference_mode(), ctx:
            state = proc.set_image(chw)
            return model.predict_inst(state, point_coords=pts[i], point_labels=labels[i], box=boxes[i], multimask_output=False)

    out = {"model": f"{args.backbone}-{args.model}", "inputs": "synth.smooth_image_u8(1), noise_image_u8(2), smooth_image_u8(3), noise_image_u8(4); "
           "synth.prompts(4, seed=2): point + box, multimask_output=False", "torch": torch.__version__, "images": []}
    t0 = time.time()
    for i in range(4):
        (m32, i32, l32), (m16, i16, l16) = run(i, False), run(i, True)
        r = {"low_res": G.maxerr(l32, l16), "iou": G.maxerr(i32, i16), "mask_iou": mask_iou(m32, m16), "fg_fraction_fp32": float((m32 > 0).mean())}
        out["images"].append(r)
        print(f"image {i}: {r}  [{time.time() - t0:.0f}s]", flush=True)
    with open(os.path.join(gold, "shard_yard.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(gold, "shard_yard.json"))


if __name__ == "__main__":
    main()
