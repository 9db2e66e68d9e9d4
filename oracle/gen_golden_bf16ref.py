"""ORACLE tooling (test infrastructure only): the REFERENCE's own bf16 behaviour as a yardstick.

On a GPU host the reference runs under bf16 autocast without being asked to
(`sam3/sam3/model/sam3_tracking_predictor.py:49-60` enters a process-global autocast; the examples wrap
`set_image` / `predict_inst` in `torch.autocast(..., dtype=torch.bfloat16)`,
`sam3/efficientsam3_examples/efficientsam3_for_sam1_task_example.py:42-46`).  This script runs the REAL reference twice
on the same seeded weights and inputs as `oracle/gen_golden.py` -- once in fp32, once under
`torch.autocast("cpu", dtype=torch.bfloat16)` -- and records

  * the bf16 run's outputs per prompt case (low-res logits, IoU scores, bit-packed final masks), and
  * its distance to the fp32 run: max-abs error of the low-res logits and of the IoU head, IoU of the thresholded final
    masks, and per stage tensor the max-abs error (absolute and relative to the fp32 tensor's max).

`tests/` hold the engine's bf16 mode to this yardstick: engine-bf16-vs-reference-fp32 error <= FACTOR x
reference-bf16-vs-reference-fp32 error (+ a small absolute floor), per case and per stage.

Runs only in the build container (needs /root/reference); the fp32 fixtures of gen_golden.py are not touched:

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_bf16ref.py \
        [--backbone efficientvit|repvit|tinyvit|sam3 --model b1|m1.1|11m|vit_h]

Output: tests/golden[/<backbone>_<model>]/bf16ref.npz + bf16ref_manifest.json
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


def mask_iou(a, b) -> float:
    a, b = np.asarray(a) > 0, np.asarray(b) > 0
    return float(np.logical_and(a, b).sum() / max(np.logical_or(a, b).sum(), 1))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="efficientvit")
    ap.add_argument("--model", default="b1")
    args = ap.parse_args()
    default = (args.backbone, args.model) == ("efficientvit", "b1")
    gold = G.GOLD if default else os.path.join(G.GOLD, f"{args.backbone}_{args.model}")
    cases = G.CASES if default else [c for c in G.CASES if c["name"] in G.OTHER_STUDENT_CASES]
    resize_cases = G.RESIZE_CASES if default else []
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    os.makedirs(gold, exist_ok=True)
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
    import hashlib
    digest = hashlib.sha256()
    for k, v in sd.items():
        digest.update(k.encode())
        digest.update(np.ascontiguousarray(v.numpy()).tobytes())

    trunk = model.backbone.vision_backbone.trunk
    bb = trunk if args.backbone == "sam3" else trunk.model.backbone

    def run(chw_u8, hw, kws, amp: bool):
        """set_image + every prompt case on one image; stage tensors of the same pass"""
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if amp else torch.autocast("cpu", enabled=False)
        with torch.inference_mode(), ctx:
            x = proc.transform(chw_u8)[None] if tuple(chw_u8.shape[-2:]) != (1008, 1008) else \
                torch.from_numpy(synth.normalise_to_chw_f32(np.moveaxis(chw_u8.numpy(), 0, -1)))[None]
            stages = G.reference_stage_taps(args.backbone, bb, x)
            stages["trunk"] = trunk(x)[0]
            state = proc.set_image(chw_u8)
            bo = state["backbone_out"]
            for i in range(3):
                stages[f"sam3_fpn{i}"] = bo["backbone_fpn"][i]
                stages[f"sam2_fpn{i}"] = bo["sam2_backbone_out"]["backbone_fpn"][i]
            outs = []
            for (h, w), kw in zip(hw, kws):
                state["original_height"], state["original_width"] = h, w
                outs.append(model.predict_inst(state, **kw))
        return {k: v.float() for k, v in stages.items()}, outs

    manifest = {"weights_sha256": digest.hexdigest(), "model": f"{args.backbone}-{args.model}",
                "autocast": "torch.autocast('cpu', dtype=torch.bfloat16)", "torch": torch.__version__,
                "cases": {}, "stages": {}}
    arrays = {}
    t0 = time.time()

    img_u8 = synth.smooth_image_u8(seed=1)
    chw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(img_u8, -1, 0)))
    hw = [tuple(c["hw"]) for c in cases]
    kws = [G.np_kw(c["kw"]) for c in cases]
    st32, out32 = run(chw, hw, kws, amp=False)
    st16, out16 = run(chw, hw, kws, amp=True)
    for k in st32:
        e = G.maxerr(st32[k], st16[k])
        peak = float(st32[k].abs().max())
        manifest["stages"][f"img0/{k}"] = {"maxabs": e, "rel_to_peak": e / max(peak, 1e-30), "peak": peak}
        print(f"  stage {k:12s} bf16ref-vs-fp32 maxabs {e:.3e}  ({e / max(peak, 1e-30):.2%} of peak {peak:.2f})")

    if default:  # the second stage-fixture image of gen_golden.py (noise): stage yardstick only
        chw1 = torch.from_numpy(np.ascontiguousarray(np.moveaxis(synth.noise_image_u8(seed=3), -1, 0)))
        s32, _ = run(chw1, [], [], amp=False)
        s16, _ = run(chw1, [], [], amp=True)
        for k in s32:
            e = G.maxerr(s32[k], s16[k])
            peak = float(s32[k].abs().max())
            manifest["stages"][f"img1/{k}"] = {"maxabs": e, "rel_to_peak": e / max(peak, 1e-30), "peak": peak}

    def record(name, r32, r16, return_logits):
        m32, i32, l32 = r32
        m16, i16, l16 = r16
        errs = {"low_res": G.maxerr(l32, l16), "iou": G.maxerr(i32, i16), "mask_iou": mask_iou(m32, m16),
                "low_res_range": [float(l32.min()), float(l32.max())]}
        manifest["cases"][name] = errs
        arrays[f"{name}/low_res"] = np.asarray(l16, dtype=np.float32)
        arrays[f"{name}/iou"] = np.asarray(i16, dtype=np.float32)
        arrays[f"{name}/mask_bits"] = np.packbits((np.asarray(m16) > 0).reshape(-1))
        arrays[f"{name}/mask_shape"] = np.asarray(np.asarray(m16).shape, dtype=np.int64)
        print(f"  case {name:30s} bf16ref-vs-fp32: {errs}")

    for c, r32, r16 in zip(cases, out32, out16):
        record(c["name"], r32, r16, c["kw"].get("return_logits"))
    for c in resize_cases:
        img = G.resized_smooth_image(c["size"], c["seed"])
        chw_r = torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0)))
        kw = [G.np_kw(c["kw"])]
        _, o32 = run(chw_r, [tuple(c["size"])], kw, amp=False)
        _, o16 = run(chw_r, [tuple(c["size"])], kw, amp=True)
        record(c["name"], o32[0], o16[0], False)

    np.savez_compressed(os.path.join(gold, "bf16ref.npz"), **arrays)
    with open(os.path.join(gold, "bf16ref_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(f"wrote {gold}/bf16ref.npz in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
