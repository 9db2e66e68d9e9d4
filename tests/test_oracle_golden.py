"""CPU: the oracle (oracle/ref_model.py) reproduces the committed golden fixtures, which are
the REAL reference's outputs (oracle/gen_golden.py).  Also pins the seeded weight generator:
if torch's CPU RNG stream ever differed on another machine the checksum test fails first."""
import hashlib
import os

import numpy as np
import pytest
import torch

from efficientsam3_amd import synth
from oracle import ref_model
from tests import util as U

SAMPLE = 4096


def _sample(t):
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // SAMPLE)
    return flat[::step][:SAMPLE].float().numpy()


def _sd_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.numpy()).tobytes())
    return h.hexdigest()


def test_weight_generator_is_pinned(state_dict, manifest):
    assert _sd_digest(state_dict) == manifest["weights_sha256"]


def test_oracle_was_pinned_against_reference(manifest):
    errs = manifest["oracle_vs_reference_maxabs"]
    stage = [v for k, v in errs.items() if k.startswith("img")]
    assert stage and max(stage) <= 1e-6
    for k, v in errs.items():
        if k.startswith("case/"):
            assert v["low_res"] <= 1e-4 and v["iou"] <= 1e-5 and v["mask_iou"] >= 0.999, (k, v)


@pytest.fixture(scope="module")
def oracle_state(state_dict):
    img = synth.smooth_image_u8(seed=1)
    x = torch.from_numpy(synth.normalise_to_chw_f32(img))[None]
    taps = {}
    with torch.inference_mode():
        st = ref_model.set_image(state_dict, x, (1008, 1008), "b1", taps)
    return st, taps


def test_oracle_stages_vs_golden(oracle_state, golden_dir):
    st, taps = oracle_state
    gold = np.load(os.path.join(golden_dir, "stages_img0.npz"))
    bo = st["backbone_out"]
    got = {f"stage{i}": taps[f"stage{i}"] for i in range(5)}
    got["trunk"] = taps["trunk"]
    for i in range(3):
        got[f"sam3_fpn{i}"] = bo["backbone_fpn"][i]
        got[f"sam2_fpn{i}"] = bo["sam2_backbone_out"]["backbone_fpn"][i]
        got[f"pos{i}"] = bo["vision_pos_enc"][i]
    for k, v in got.items():
        assert float(np.abs(_sample(v) - gold[k]).max()) <= 1e-5, k


def test_oracle_cases_vs_golden(oracle_state, state_dict, golden_dir, manifest):
    st, _ = oracle_state
    for name, case in manifest["cases"].items():
        g = np.load(os.path.join(golden_dir, f"case_{name}.npz"))
        kw = U.case_kwargs(case)
        own = U.case_image_chw_u8(case)
        with torch.inference_mode():
            if own is None:
                st["original_height"], st["original_width"] = case["hw"]
                cst = st
            else:  # image that goes through the processor's antialiased resize
                x = ref_model.processor_transform(own)
                assert float(np.abs(_sample(x) - g["input_sample"]).max()) == 0.0, name
                cst = ref_model.set_image(state_dict, x[None], tuple(case["hw"]), "b1")
            masks, iou, low = ref_model.predict_inst(state_dict, cst, **kw)
        assert list(masks.shape) == list(g["mask_shape"])
        assert float(np.abs(low - g["low_res"]).max()) <= 1e-4, name
        assert float(np.abs(iou - g["iou"]).max()) <= 1e-5, name
        ref_bits = np.unpackbits(g["mask_bits"])[: masks.size].reshape(masks.shape).astype(bool)
        inter = np.logical_and(masks > 0, ref_bits).sum()
        union = max(np.logical_or(masks > 0, ref_bits).sum(), 1)
        assert inter / union >= 0.999, name
