"""ORACLE tooling (test infrastructure only): pin ``oracle/ref_pcs.py`` against the REAL reference's
text-grounding path and write fixtures under ``tests/golden/pcs_ev_m/``.  Runs only where
``/root/reference`` exists:

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_pcs.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from efficientsam3_amd import schema, synth  # noqa: E402
from oracle import ref_pcs  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "pcs_ev_m")
CTX = 16
PROMPTS = ["dog", "traffic light"]
SAMPLE = 8192
THRESH = 0.05  # the random-weight presence score is ~0.09: a low threshold keeps some detections in the fixtures


def sample(t):
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // SAMPLE)
    return flat[::step][:SAMPLE].float().numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    os.makedirs(GOLD, exist_ok=True)
    from sam3 import build_efficientsam3_image_model  # the REAL reference
    from sam3.model.sam3_image_processor import Sam3Processor
    model = build_efficientsam3_image_model(
        device="cpu", checkpoint_path=None, load_from_HF=False, enable_inst_interactivity=False,
        backbone_type="efficientvit", model_name="b1", text_encoder_type="MobileCLIP-S0", text_encoder_context_length=CTX)
    sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0, enable_inst_interactivity=False)
    sd.update(schema.synthetic_text_state_dict("MobileCLIP-S0", CTX, seed=0))
    sd.update(schema.synthetic_pcs_state_dict(seed=0))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected[:5]
    assert not missing, missing[:8]
    model.eval()
    digest = hashlib.sha256()
    for k, v in schema.synthetic_pcs_state_dict(seed=0).items():
        digest.update(k.encode())
        digest.update(np.ascontiguousarray(v.numpy()).tobytes())

    proc = Sam3Processor(model, device="cpu", confidence_threshold=THRESH)
    img = synth.smooth_image_u8(seed=1)
    chw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0)))
    captured = {}
    orig = model.forward_grounding

    def wrapped(*a, **k):
        out = orig(*a, **k)
        captured["out"] = out
        return out

    model.forward_grounding = wrapped
    manifest = {"weights_sha256_pcs": digest.hexdigest(), "prompts": PROMPTS, "context_length": CTX, "confidence_threshold": THRESH,
                "cases": {}}
    arrays = {}
    with torch.inference_mode():
        state = proc.set_image(chw)
        bo = state["backbone_out"]
        for pi, text in enumerate(PROMPTS):
            state = proc.set_text_prompt(text, state)
            out_r = captured["out"]
            taps = {}
            out_o = ref_pcs.forward_grounding(sd, bo["backbone_fpn"], bo["vision_pos_enc"][-1], bo["language_features"],
                                              bo["language_mask"], taps)
            post_o = ref_pcs.postprocess_grounding(out_o, (1008, 1008), THRESH)
            errs = {k: float((out_r[k] - out_o[k]).abs().max()) for k in
                    ("pred_logits", "pred_boxes", "presence_logit_dec", "pred_masks", "semantic_seg")}
            errs["encoder_hidden_states"] = float((out_r["encoder_hidden_states"].transpose(0, 1) - taps["memory"]).abs().max())
            errs["n_kept_ref"], errs["n_kept_oracle"] = int(state["scores"].numel()), int(post_o["scores"].numel())
            if errs["n_kept_ref"] == errs["n_kept_oracle"] and errs["n_kept_ref"] > 0:
                errs["scores"] = float((state["scores"] - post_o["scores"]).abs().max())
                errs["boxes"] = float((state["boxes"] - post_o["boxes"]).abs().max())
                errs["masks_mismatch_frac"] = float((state["masks"] != post_o["masks"]).float().mean())
            print(text, errs, "| logits range", float(out_r["pred_logits"].min()), float(out_r["pred_logits"].max()),
                  "presence", float(out_r["presence_logit_dec"]), "masks range", float(out_r["pred_masks"].min()),
                  float(out_r["pred_masks"].max()))
            manifest["cases"][text] = {"oracle_vs_reference_maxabs": errs}
            arrays[f"{pi}_language_features"] = bo["language_features"].numpy()
            arrays[f"{pi}_language_mask"] = bo["language_mask"].numpy()
            arrays[f"{pi}_pred_logits"] = out_r["pred_logits"].numpy()
            arrays[f"{pi}_pred_boxes"] = out_r["pred_boxes"].numpy()
            arrays[f"{pi}_presence_logit_dec"] = out_r["presence_logit_dec"].numpy()
            arrays[f"{pi}_pred_masks_sample"] = sample(out_r["pred_masks"])
            arrays[f"{pi}_semantic_seg_sample"] = sample(out_r["semantic_seg"])
            arrays[f"{pi}_encoder_sample"] = sample(out_r["encoder_hidden_states"].transpose(0, 1))
            arrays[f"{pi}_scores"] = state["scores"].numpy()
            arrays[f"{pi}_boxes"] = state["boxes"].numpy()
            arrays[f"{pi}_mask_bits"] = np.packbits(state["masks"].numpy().reshape(-1))
        # ---- geometric prompts (add_geometric_prompt / add_point_prompt, sam3_image_processor.py:130-190) ----
        def geo_of(state):
            gp = state["geometric_prompt"]
            b = 1
            pts = gp.point_embeddings if gp.point_embeddings is not None else torch.zeros(0, b, 2)
            bxs = gp.box_embeddings if gp.box_embeddings is not None else torch.zeros(0, b, 4)
            pl = gp.point_labels if gp.point_labels is not None else torch.zeros(0, b, dtype=torch.long)
            bl = gp.box_labels if gp.box_labels is not None else torch.zeros(0, b, dtype=torch.long)
            pm = gp.point_mask if gp.point_mask is not None else torch.zeros(b, pts.shape[0], dtype=torch.bool)
            bm = gp.box_mask if gp.box_mask is not None else torch.zeros(b, bxs.shape[0], dtype=torch.bool)
            return {"points": pts.transpose(0, 1).float(), "point_labels": pl.transpose(0, 1).long(), "point_mask": pm.bool(),
                    "boxes": bxs.transpose(0, 1).float(), "box_labels": bl.transpose(0, 1).long(), "box_mask": bm.bool()}

        def record(name, state):
            out_r = captured["out"]
            geo = geo_of(state)
            out_o = ref_pcs.forward_grounding(sd, bo["backbone_fpn"], bo["vision_pos_enc"][-1], bo["language_features"],
                                              bo["language_mask"], None, geo)
            errs = {k: float((out_r[k] - out_o[k]).abs().max()) for k in
                    ("pred_logits", "pred_boxes", "presence_logit_dec", "pred_masks", "semantic_seg")}
            print(name, errs, {k: tuple(v.shape) for k, v in geo.items()})
            manifest["cases"][name] = {"oracle_vs_reference_maxabs": errs}
            for k, v in geo.items():
                arrays[f"{name}_in_{k}"] = v.numpy()
            arrays[f"{name}_language_features"] = bo["language_features"].numpy()
            arrays[f"{name}_language_mask"] = bo["language_mask"].numpy()
            arrays[f"{name}_pred_logits"] = out_r["pred_logits"].numpy()
            arrays[f"{name}_pred_boxes"] = out_r["pred_boxes"].numpy()
            arrays[f"{name}_presence_logit_dec"] = out_r["presence_logit_dec"].numpy()
            arrays[f"{name}_pred_masks_sample"] = sample(out_r["pred_masks"])
            arrays[f"{name}_scores"] = state["scores"].numpy()
            arrays[f"{name}_boxes"] = state["boxes"].numpy()
            arrays[f"{name}_mask_bits"] = np.packbits(state["masks"].numpy().reshape(-1))

        proc.reset_all_prompts(state)
        state = proc.set_text_prompt("dog", state)
        state = proc.add_geometric_prompt([0.45, 0.5, 0.3, 0.4], True, state)
        record("geo_text_box", state)
        proc.reset_all_prompts(state)
        state = proc.add_point_prompt([300.0, 420.0], 1, state)          # no text: the reference encodes "visual"
        state = proc.add_geometric_prompt([0.6, 0.4, 0.2, 0.25], False, state)
        state = proc.add_point_prompt([700.5, 200.0], 0, state)
        state = proc.add_geometric_prompt([0.25, 0.7, 0.45, 0.5], True, state)
        record("geo_visual_mixed", state)
    manifest["geometric_cases"] = ["geo_text_box", "geo_visual_mixed"]
    np.savez_compressed(os.path.join(GOLD, "pcs_cases.npz"), **arrays)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
