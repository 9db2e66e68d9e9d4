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


ALL_STUDENTS = [("efficientvit", "b0"), ("efficientvit", "b2"), ("repvit", "m0.9"), ("repvit", "m1.1"), ("repvit", "m2.3"),
                ("tinyvit", "5m"), ("tinyvit", "11m"), ("tinyvit", "21m"), ("sam3", "vit_h")]


@pytest.mark.parametrize("bt,mn", ALL_STUDENTS, ids=lambda v: str(v))
def test_every_model_size_is_pinned_against_the_reference(golden_dir, bt, mn):
    """Every student size (S / M / L of the three families) and the ViT-H teacher has fixtures written by the REAL
    reference; the manifests record that the oracle reproduced them (stage tensors exactly or within fp32 rounding
    for ViT-H's real-valued RoPE, decode within 1e-4) and the checksum of the seeded weights they were made with."""
    import json
    from efficientsam3_amd import schema
    with open(os.path.join(golden_dir, f"{bt}_{mn}", "manifest.json")) as f:
        man = json.load(f)
    errs = man["oracle_vs_reference_maxabs"]
    stage = [v for k, v in errs.items() if k.startswith("img")]
    assert stage and max(stage) <= (1e-4 if bt == "sam3" else 1e-6)
    cases = {k: v for k, v in errs.items() if k.startswith("case/")}
    assert cases
    for k, v in cases.items():
        assert v["low_res"] <= 2e-4 and v["iou"] <= 1e-5 and v["mask_iou"] >= 0.999, (bt, mn, k, v)
    assert _sd_digest(schema.synthetic_state_dict(bt, mn, seed=0)) == man["weights_sha256"], (bt, mn)


@pytest.mark.parametrize("bt,mn", [("efficientvit", "b0"), ("repvit", "m0.9"), ("tinyvit", "5m")], ids=lambda v: str(v))
def test_small_sizes_oracle_reproduces_reference_trunk(golden_dir, bt, mn):
    """The S sizes re-run here (the M sizes are re-run by the tests above and on the GPU box): oracle trunk embedding
    and low-res logits of one prompt vs the reference's fixtures."""
    from efficientsam3_amd import schema
    gdir = os.path.join(golden_dir, f"{bt}_{mn}")
    sd = schema.synthetic_state_dict(bt, mn, seed=0)
    x = torch.from_numpy(synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=1)))[None]
    taps = {}
    with torch.inference_mode():
        st = ref_model.set_image(sd, x, (1008, 1008), mn, taps)
        g = np.load(os.path.join(gdir, "stages_img0.npz"))
        assert float(np.abs(_sample(taps["trunk"]) - g["trunk"]).max()) <= 1e-5
        case = np.load(os.path.join(gdir, "case_point_box_single.npz"))
        masks, iou, low = ref_model.predict_inst(sd, st, point_coords=np.array([[450.0, 500.0]], np.float32),
                                                 point_labels=np.array([1]), box=np.array([180.0, 240.0, 700.0, 820.0], np.float32),
                                                 multimask_output=False)
    key = "low_res" if "low_res" in case.files else [k for k in case.files if "low" in k][0]
    assert float(np.abs(low - case[key]).max()) <= 2e-4
