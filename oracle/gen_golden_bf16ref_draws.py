"""ORACLE tooling (test infrastructure only): the bf16 yardstick as a DISTRIBUTION.

`oracle/gen_golden_bf16ref.py` measures, per prompt case, how far the REAL reference under `torch.autocast("cpu", bfloat16)`
is from its own fp32 run -- on ONE image.  A maximum over a 288 x 288 logit map, a 4-number IoU-head maximum and the IoU of
a thresholded mask are noisy single draws: the reference's own values scatter by 3 - 4 x between the prompt cases of one
model ("What's weak").  This script repeats the measurement on `--draws` further seeded images (the
smooth synthetic images of `efficientsam3_amd/synth.py`, seeds 101, 102, ...) with the SAME prompts, weights and code path,
and records every draw.  `tests/util.py: bf16_case_limits` then takes, per case and quantity, the WORST of the reference's own
draws (the fixture image included) as the yardstick the engine's bf16 mode is held to (x 1.5).  Nothing about the engine
enters this file: it is the reference against itself.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_bf16ref_draws.py \
        [--backbone efficientvit|repvit|tinyvit|sam3 --model b1|m1.1|11m|vit_h] [--draws 5]

Output: tests/golden[/<backbone>_<model>]/bf16ref_draws.json
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
    ap.add_argument("--backbone", default="efficientvit")
    ap.add_argument("--model", default="b1")
    ap.add_argument("--draws", type=int, default=5)
    ap.add_argument("--first-seed", type=int, default=101)
    args = ap.parse_args()
    default = (args.backbone, args.model) == ("efficientvit", "b1")
    gold = G.GOLD if default else os.path.join(G.GOLD, f"{args.backbone}_{args.model}")
    with open(os.path.join(gold, "bf16ref_manifest.json")) as f:
        single = json.load(f)
    # the prompt cases this model has a single-draw yardstick for, on the shared 1008 x 1008 image (resize cases keep theirs)
    cases = [c for c in G.CASES if c["name"] in single["cases"]]
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    from sam3 import build_efficientsam3_image_model, build_sam3_image_model  # the REAL reference
    from sam3.model.sam3_image_processor import Sam3Processor

    if args.backbone == "sam3":
        model = build_sam3_image_model(device="cpu", checkpoint_path=None, load_from_HF=False,
                                       enable_inst_interactivity=True, enable_text_encoder=False)
    else:
        model = build_efficientsam3_image_model(
            device="cpu", checkpoint_path=None, load_from_HF=False, enable_inst_interactivity=True,
            backbone_type=args.backbone, model_name=args.model, text_encoder_type="MobileCLIP-S0",
            text_encoder_context_length=16)
    sd = schema.synthetic_state_dict(args.backbone, args.model, seed=0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected[:5]
    model.eval()
    proc = Sam3Processor(model, device="cpu")

    def run(chw_u8, amp: bool):
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if amp else torch.autocast("cpu", enabled=False)
        outs = []
        with torch.inference_mode(), ctx:
            state = proc.set_image(chw_u8)
            for c in cases:
                state["original_height"], state["original_width"] = tuple(c["hw"])
                outs.append(model.predict_inst(state, **G.np_kw(c["kw"])))
        return outs

    out = {"model": f"{args.backbone}-{args.model}", "weights_sha256": single["weights_sha256"],
           "image_seeds": [], "torch": torch.__version__, "cases": {c["name"]: {"low_res": [], "iou": [], "mask_iou": []} for c in cases}}
    t0 = time.time()
    for d in range(args.draws):
        seed = args.first_seed + d
        img = synth.smooth_image_u8(seed=seed)
        chw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0)))
        o32 = run(chw, False)
        o16 = run(chw, True)
        out["image_seeds"].append(seed)
        for c, (m32, i32, l32), (m16, i16, l16) in zip(cases, o32, o16):
            r = out["cases"][c["name"]]
            r["low_res"].append(G.maxerr(l32, l16))
            r["iou"].append(G.maxerr(i32, i16))
            r["mask_iou"].append(mask_iou(m32, m16))
        print(f"draw {d} (image seed {seed}) done after {time.time() - t0:.0f}s", flush=True)
        with open(os.path.join(gold, "bf16ref_draws.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
    for n, r in out["cases"].items():
        s = single["cases"][n]
        print(f"  {n:32s} low_res {s['low_res']:.3f} | {max(r['low_res']):.3f}   iou {s['iou']:.2e} | {max(r['iou']):.2e}   "
              f"mask_iou {s['mask_iou']:.4f} | {min(r['mask_iou']):.4f}   (fixture image | worst of {args.draws} draws)")


if __name__ == "__main__":
    main()
