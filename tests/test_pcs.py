"""PCS text-grounding detector (SURVEY.md §8a "PCS detector", config 4).

CPU: the oracle restatement (oracle/ref_pcs.py) reproduces the REAL reference's outputs (fixtures written by
oracle/gen_golden_pcs.py) from the image oracle's features and the reference's text features.
GPU: the HIP engine (esam3_ground through the C ABI / Sam3Processor.set_text_prompt) vs the same fixtures."""
import json
import os

import numpy as np
import pytest
import torch

from efficientsam3_amd import schema, synth

SAMPLE = 8192


def _sample(t):
    flat = t.detach().float().cpu().reshape(-1)
    step = max(1, flat.numel() // SAMPLE)
    return flat[::step][:SAMPLE].numpy()


@pytest.fixture(scope="module")
def pcs_gold(golden_dir):
    d = os.path.join(golden_dir, "pcs_ev_m")
    with open(os.path.join(d, "manifest.json")) as f:
        man = json.load(f)
    return man, np.load(os.path.join(d, "pcs_cases.npz"))


@pytest.fixture(scope="module")
def pcs_sd():
    sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0, enable_inst_interactivity=False)
    sd.update(schema.synthetic_text_state_dict("MobileCLIP-S0", 16, seed=0))
    sd.update(schema.synthetic_pcs_state_dict(seed=0))
    return sd


def test_pcs_oracle_pinned_and_reproduces_golden(pcs_gold, pcs_sd):
    from oracle import ref_model, ref_pcs
    man, g = pcs_gold
    for case in man["cases"].values():
        e = case["oracle_vs_reference_maxabs"]
        assert e["pred_logits"] <= 1e-5 and e["pred_boxes"] <= 1e-5 and e["pred_masks"] <= 1e-3
        assert e["n_kept_ref"] == e["n_kept_oracle"]
    x = torch.from_numpy(synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=1)))[None]
    with torch.inference_mode():
        st = ref_model.set_image(pcs_sd, x, (1008, 1008), "b1")
        bo = st["backbone_out"]
        for pi in range(len(man["prompts"])):
            out = ref_pcs.forward_grounding(pcs_sd, bo["backbone_fpn"], bo["vision_pos_enc"][-1],
                                            torch.from_numpy(g[f"{pi}_language_features"]),
                                            torch.from_numpy(g[f"{pi}_language_mask"]))
            assert float(np.abs(out["pred_logits"].numpy() - g[f"{pi}_pred_logits"]).max()) <= 1e-4
            assert float(np.abs(out["pred_boxes"].numpy() - g[f"{pi}_pred_boxes"]).max()) <= 1e-4
            assert float(np.abs(out["presence_logit_dec"].numpy() - g[f"{pi}_presence_logit_dec"]).max()) <= 1e-4
            assert float(np.abs(_sample(out["pred_masks"]) - g[f"{pi}_pred_masks_sample"]).max()) <= 1e-3
            post = ref_pcs.postprocess_grounding(out, (1008, 1008), man["confidence_threshold"])
            assert post["scores"].numel() == g[f"{pi}_scores"].size
            assert float(np.abs(post["scores"].numpy() - g[f"{pi}_scores"]).max()) <= 1e-5
            assert float(np.abs(post["boxes"].numpy() - g[f"{pi}_boxes"]).max()) <= 1e-2  # pixels
            bits = np.unpackbits(g[f"{pi}_mask_bits"])[: post["masks"].numel()].reshape(post["masks"].shape).astype(bool)
            assert float((post["masks"].numpy() != bits).mean()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_pcs_engine_vs_golden(pcs_gold, pcs_sd, mode):
    """The HIP engine's grounding path (esam3_ground) on the engine's own EV-M features and the REFERENCE's text
    features vs the reference's outputs.  f32: logits / boxes within 1e-4, mask logits within 2e-3 on a -21..63
    range, identical detections after thresholding.  bf16: inside the bf16 envelope of the same quantities."""
    from efficientsam3_amd import Sam3Processor, build_efficientsam3_image_model
    man, g = pcs_gold
    model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit",
                                            model_name="b1", dtype=mode, state_dict=pcs_sd, text_encoder_type="MobileCLIP-S0",
                                            text_encoder_context_length=16)
    proc = Sam3Processor(model, confidence_threshold=man["confidence_threshold"])
    img = synth.smooth_image_u8(seed=1)
    state = proc.set_image(torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0))))
    lim = dict(f32=dict(logits=1e-4, boxes=1e-4, presence=1e-4, masks=2e-3), bf16=dict(logits=0.08, boxes=0.06, presence=0.08, masks=2.5))[mode]
    for pi in range(len(man["prompts"])):
        state["backbone_out"]["language_features"] = torch.from_numpy(g[f"{pi}_language_features"]).to("cuda")
        state["backbone_out"]["language_mask"] = torch.from_numpy(g[f"{pi}_language_mask"]).to("cuda")
        out = model.forward_grounding(state["backbone_out"], geometric_prompt=model._get_dummy_prompt())
        e = dict(logits=float(np.abs(out["pred_logits"].cpu().numpy() - g[f"{pi}_pred_logits"]).max()),
                 boxes=float(np.abs(out["pred_boxes"].cpu().numpy() - g[f"{pi}_pred_boxes"]).max()),
                 presence=float(np.abs(out["presence_logit_dec"].cpu().numpy() - g[f"{pi}_presence_logit_dec"]).max()),
                 masks=float(np.abs(_sample(out["pred_masks"]) - g[f"{pi}_pred_masks_sample"]).max()))
        print(f"[pcs {mode}] prompt {pi}: {e}")
        for k, v in e.items():
            assert v <= lim[k], (k, v, lim[k])
        state["geometric_prompt"] = model._get_dummy_prompt()
        st = proc._forward_grounding(state)
        n_ref = g[f"{pi}_scores"].size
        if mode == "f32":
            assert st["scores"].numel() == n_ref
            assert float(np.abs(st["scores"].cpu().numpy() - g[f"{pi}_scores"]).max()) <= 1e-4
            assert float(np.abs(st["boxes"].cpu().numpy() - g[f"{pi}_boxes"]).max()) <= 1.0  # pixels
            bits = np.unpackbits(g[f"{pi}_mask_bits"])[: st["masks"].numel()].reshape(tuple(st["masks"].shape)).astype(bool)
            assert float((st["masks"].cpu().numpy() != bits).mean()) <= 1e-4
        else:
            assert abs(st["scores"].numel() - n_ref) <= max(4, n_ref // 10)
        assert st["masks"].dtype == torch.bool and st["masks"].shape[1:] == (1, 1008, 1008)
        assert st["masks_logits"].shape == st["masks"].shape and st["boxes"].shape == (st["scores"].numel(), 4)
