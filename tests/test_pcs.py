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
