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
from tests.util import pcs_bf16_yard

SAMPLE = 8192
BF16_FACTOR = 1.5  # engine-bf16 error allowed as a multiple of the reference's own bf16 error (tests/util.py)
# Since the detector's 200-query decoder stream is kept in fp32 (as the reference's autocast keeps it), the fixture cases
# (same image and prompts as the yardstick run) are inside the 1.5 x rule on every output.  Round 5: NO flat factor on inputs the
# yardstick was not measured on remains -- the geometric-prompt cases have their own reference-bf16 runs (bf16ref_geo.json,
# oracle/gen_golden_pcs_bf16ref.py --geo), the config-4 batch-8 test runs on the yardstick's own (image, prompt) pairs, and both are
# held to the DISTRIBUTION rule of tests/util.py (median <= 1.25 x the reference's median, max <= 1.5 x its max, same images).
PRESENCE_BF16_ULP = 2.0 ** -6  # the presence logit is ONE number per image (about -2): one bf16 ulp at that magnitude


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
    for text in man["prompts"]:
        e = man["cases"][text]["oracle_vs_reference_maxabs"]
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
    yard = pcs_bf16_yard("pcs_ev_m")  # the reference's own bf16-autocast-vs-fp32 distance per output, worst over 1 + 6 seeded images
    for pi in range(len(man["prompts"])):
        y = yard[man["prompts"][pi]]
        lim = dict(logits=1e-4, boxes=1e-4, presence=1e-4, masks=2e-3) if mode == "f32" else \
            dict(logits=BF16_FACTOR * y["pred_logits"], boxes=BF16_FACTOR * y["pred_boxes"],
                 presence=BF16_FACTOR * y["presence_logit_dec"] + PRESENCE_BF16_ULP, masks=BF16_FACTOR * y["pred_masks"])
        state["backbone_out"]["language_features"] = torch.from_numpy(g[f"{pi}_language_features"]).to("cuda")
        state["backbone_out"]["language_mask"] = torch.from_numpy(g[f"{pi}_language_mask"]).to("cuda")
        if pi == 0 and mode == "bf16":   # the phases of esam3_ground carry the reference's record_function names (sam3_image.py:449-479)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
                out = model.forward_grounding(state["backbone_out"], geometric_prompt=model._get_dummy_prompt())
            names = [e_.name for e_ in sorted(prof.events(), key=lambda e_: e_.time_range.start)]
            scopes = ["SAM3Image._encode_prompt", "SAM3Image._run_encoder", "SAM3Image._run_decoder", "SAM3Image._run_segmentation_heads"]
            assert [n for n in names if n in scopes] == scopes, [n for n in names if n.startswith("SAM3Image")]
        else:
            out = model.forward_grounding(state["backbone_out"], geometric_prompt=model._get_dummy_prompt())
        e = dict(logits=float(np.abs(out["pred_logits"].cpu().numpy() - g[f"{pi}_pred_logits"]).max()),
                 boxes=float(np.abs(out["pred_boxes"].cpu().numpy() - g[f"{pi}_pred_boxes"]).max()),
                 presence=float(np.abs(out["presence_logit_dec"].cpu().numpy() - g[f"{pi}_presence_logit_dec"]).max()),
                 masks=float(np.abs(_sample(out["pred_masks"]) - g[f"{pi}_pred_masks_sample"]).max()))
        print(f"[pcs {mode}] prompt {pi}: err {e} allowed {lim}")
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


def _geo_case(g, name):
    keys = ("points", "point_labels", "point_mask", "boxes", "box_labels", "box_mask")
    return {k: torch.from_numpy(g[f"{name}_in_{k}"]) for k in keys}


def test_pcs_oracle_geometric_prompts_reproduce_golden(pcs_gold, pcs_sd):
    """Box / point geometric prompts (add_geometric_prompt / add_point_prompt): the oracle's geometry encoder
    (roi_align, grid_sample, sine encodings, padded concatenation) vs the REAL reference's outputs."""
    from oracle import ref_model, ref_pcs
    man, g = pcs_gold
    x = torch.from_numpy(synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=1)))[None]
    with torch.inference_mode():
        bo = ref_model.set_image(pcs_sd, x, (1008, 1008), "b1")["backbone_out"]
        for name in man["geometric_cases"]:
            e = man["cases"][name]["oracle_vs_reference_maxabs"]
            assert e["pred_logits"] <= 1e-5 and e["pred_boxes"] <= 1e-5 and e["pred_masks"] <= 1e-3
            out = ref_pcs.forward_grounding(pcs_sd, bo["backbone_fpn"], bo["vision_pos_enc"][-1],
                                            torch.from_numpy(g[f"{name}_language_features"]),
                                            torch.from_numpy(g[f"{name}_language_mask"]), None, _geo_case(g, name))
            assert float(np.abs(out["pred_logits"].numpy() - g[f"{name}_pred_logits"]).max()) <= 1e-4
            assert float(np.abs(out["pred_boxes"].numpy() - g[f"{name}_pred_boxes"]).max()) <= 1e-4
            assert float(np.abs(_sample(out["pred_masks"]) - g[f"{name}_pred_masks_sample"]).max()) <= 1e-3
    # the fixtures do depend on the geometric prompt
    assert float(np.abs(g["geo_text_box_pred_logits"] - g["0_pred_logits"]).max()) > 0.1


def test_geometry_prompt_container_matches_reference_semantics():
    """geometry_prompt.Prompt: appended boxes / points stay right-padded per image (geometry_encoders.py:22-79,331-375)."""
    from efficientsam3_amd.geometry_prompt import Prompt
    p = Prompt(box_embeddings=torch.zeros(0, 2, 4), box_mask=torch.zeros(2, 0, dtype=torch.bool))
    assert p.n_prompts == 0
    b1 = torch.tensor([[[0.1, 0.2, 0.3, 0.4], [0.5, 0.5, 0.2, 0.2]]])           # [1, B=2, 4]
    p.append_boxes(b1, torch.tensor([[True, False]]), mask=torch.tensor([[False], [True]]))  # image 1: padding
    b2 = torch.tensor([[[0.6, 0.6, 0.1, 0.1], [0.7, 0.7, 0.3, 0.3]]])
    p.append_boxes(b2, torch.tensor([[False, True]]))
    bf = p.batch_first()
    assert bf["boxes"].shape == (2, 2, 4) and bf["box_mask"].tolist() == [[0, 0], [0, 1]]
    assert torch.allclose(bf["boxes"][0], torch.stack([b1[0, 0], b2[0, 0]]))
    assert torch.allclose(bf["boxes"][1, 0], b2[0, 1])                         # compacted to the front
    assert bf["box_labels"][0].tolist() == [1, 0] and bf["box_labels"][1, 0].item() == 1
    p.append_points(torch.tensor([[[0.5, 0.25], [0.1, 0.9]]]), torch.tensor([[1, 0]]))
    assert p.n_prompts == 3 and p.batch_first()["points"].shape == (2, 1, 2)
    # mask prompts (geometry_encoders.py:697-745, _encode_masks) are refused by name, never dropped
    with pytest.raises(NotImplementedError, match="_encode_masks"):
        Prompt(box_embeddings=torch.zeros(0, 1, 4), mask_embeddings=torch.zeros(1, 1, 1, 8, 8))
    with pytest.raises(NotImplementedError, match="_encode_masks"):
        p.append_masks(torch.zeros(1, 2, 1, 8, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_pcs_engine_geometric_prompts_vs_golden(pcs_gold, pcs_sd, mode):
    """esam3_ground with box / point prompts (geo_tokens kernel: grid_sample, roi_align 7x7, sine encodings,
    label embeddings; masked self-attention among the geometry tokens) vs the reference's outputs, and the
    processor methods add_geometric_prompt / add_point_prompt replaying the reference's call sequence."""
    from efficientsam3_amd import Sam3Processor, build_efficientsam3_image_model
    man, g = pcs_gold
    model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit",
                                            model_name="b1", dtype=mode, state_dict=pcs_sd, text_encoder_type="MobileCLIP-S0",
                                            text_encoder_context_length=16)
    proc = Sam3Processor(model, confidence_threshold=man["confidence_threshold"])
    img = synth.smooth_image_u8(seed=1)
    state = proc.set_image(torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0))))
    # bf16, fixture image: 1.5 x the worst of the reference's own bf16 runs of THESE cases (bf16ref_geo.json: seven images) -- the
    # "max" half of the distribution rule on one sample; test_pcs_bf16_distribution_vs_reference_draws holds both halves
    with open(os.path.join(os.path.dirname(__file__), "golden", "pcs_ev_m", "bf16ref_geo.json")) as f:
        geo_yard = json.load(f)["cases"]
    gmax = lambda name, k: max(d[k] for d in geo_yard[name])  # noqa: E731
    lims = {name: dict(logits=BF16_FACTOR * gmax(name, "pred_logits"), boxes=BF16_FACTOR * gmax(name, "pred_boxes"),
                       presence=BF16_FACTOR * gmax(name, "presence_logit_dec") + PRESENCE_BF16_ULP,
                       masks=BF16_FACTOR * gmax(name, "pred_masks")) for name in man["geometric_cases"]}
    if mode == "f32":
        lims = {name: dict(logits=1e-4, boxes=1e-4, presence=1e-4, masks=2e-3) for name in man["geometric_cases"]}
    fpn = state["backbone_out"]["_esam3_nhwc_sam3"]
    for name in man["geometric_cases"]:
        lf = torch.from_numpy(g[f"{name}_language_features"]).to("cuda")
        lm = torch.from_numpy(g[f"{name}_language_mask"]).to("cuda")
        out = model.engine.ground(fpn, lf, lm, geo=_geo_case(g, name))
        e = dict(logits=float(np.abs(out["pred_logits"].cpu().numpy() - g[f"{name}_pred_logits"]).max()),
                 boxes=float(np.abs(out["pred_boxes"].cpu().numpy() - g[f"{name}_pred_boxes"]).max()),
                 presence=float(np.abs(out["presence_logit_dec"].cpu().numpy() - g[f"{name}_presence_logit_dec"]).max()),
                 masks=float(np.abs(_sample(out["pred_masks"]) - g[f"{name}_pred_masks_sample"]).max()))
        lim = lims[name]
        print(f"[pcs geo {mode}] {name}: err {e} allowed {lim}")
        for k, v in e.items():
            assert v <= lim[k], (name, k, v, lim[k])
    # a padded batch: image 0 = the mixed case, image 1 = the same prompt with one extra (masked) entry of junk
    geo = _geo_case(g, "geo_visual_mixed")
    pad = {k: torch.cat([v, v], 0) for k, v in geo.items()}
    for k, w in (("points", 2), ("boxes", 4)):
        pad[k] = torch.cat([pad[k], torch.full((2, 1, w), 0.77)], 1)
    for k in ("point_labels", "box_labels"):
        pad[k] = torch.cat([pad[k], torch.ones((2, 1), dtype=pad[k].dtype)], 1)
    for k in ("point_mask", "box_mask"):
        pad[k] = torch.cat([pad[k], torch.ones((2, 1), dtype=torch.bool)], 1)
    fpn2 = [torch.cat([t, t], 0).contiguous() for t in fpn]
    name = "geo_visual_mixed"
    lim = lims[name]
    lf = torch.from_numpy(g[f"{name}_language_features"]).to("cuda").expand(-1, 2, -1)
    lm = torch.from_numpy(g[f"{name}_language_mask"]).to("cuda").expand(2, -1)
    out2 = model.engine.ground(fpn2, lf, lm, geo=pad)
    for bi in range(2):
        assert float(np.abs(out2["pred_logits"][bi].cpu().numpy() - g[f"{name}_pred_logits"][0]).max()) <= lim["logits"]
        assert float(np.abs(out2["pred_boxes"][bi].cpu().numpy() - g[f"{name}_pred_boxes"][0]).max()) <= lim["boxes"]
    # the processor replays the reference's call sequence; the "visual" text features come from the fixture (the
    # BPE merge table the tokenizer needs is a reference asset that does not travel to the GPU box)
    proc.reset_all_prompts(state)
    state["backbone_out"]["language_features"] = torch.from_numpy(g[f"{name}_language_features"]).to("cuda")
    state["backbone_out"]["language_mask"] = torch.from_numpy(g[f"{name}_language_mask"]).to("cuda")
    state = proc.add_point_prompt([300.0, 420.0], 1, state)
    state = proc.add_geometric_prompt([0.6, 0.4, 0.2, 0.25], False, state)
    state = proc.add_point_prompt([700.5, 200.0], 0, state)
    state = proc.add_geometric_prompt([0.25, 0.7, 0.45, 0.5], True, state)
    n_ref = g[f"{name}_scores"].size
    if mode == "f32":
        assert state["scores"].numel() == n_ref
        assert float(np.abs(state["scores"].cpu().numpy() - g[f"{name}_scores"]).max()) <= 1e-4
        assert float(np.abs(state["boxes"].cpu().numpy() - g[f"{name}_boxes"]).max()) <= 1.0
    else:
        assert abs(state["scores"].numel() - n_ref) <= max(4, n_ref // 10)


# token ids of the yardstick's prompts (oracle/gen_golden_pcs_bf16ref.py: PROMPTS) as the reference's tokenizer produces them at context
# length 16 (tokenizer_ve.py:128-253; the BPE merge table is a reference asset that does not travel to the GPU box; tests/test_text_encoder.py
# pins the tokenizer itself): <start_of_text> ... <end_of_text>, zero padding
YARD_TOKENS = {"dog": [49406, 1929, 49407], "traffic light": [49406, 3399, 1395, 49407]}
PCS_OUT = ("pred_logits", "pred_boxes", "presence_logit_dec", "pred_masks")
PCS_DIST_IMAGES = 5   # of the 7 the reference's draws cover (fixture image + seeds 2..7)


def _pcs_verdicts(tag, eng, ref, failures):
    """eng / ref: {case: {key: [distance per image]}} on the same images -> the distribution rule (tests/util.py: distribution_report)"""
    from tests.util import distribution_report
    extra = {c: {"presence_logit_dec": PRESENCE_BF16_ULP} for c in eng}
    failures.extend(distribution_report(tag, eng, ref, extra))


@pytest.mark.gpu
def test_pcs_bf16_distribution_vs_reference_draws(pcs_gold, pcs_sd):
    """EV-M detector, bf16 engine, on images the reference's own bf16 runs were taken on (fixture image seed 1 + seeds 2..5): the
    two text prompts (bf16ref_manifest.json + bf16ref_draws.json) and the two geometric-prompt cases (bf16ref_geo.json), every output
    against the fp32 oracle run live; median <= 1.25 x the reference's median and max <= 1.5 x its max per case and output (every
    element of the 200 x 288 x 288 mask logits, as the reference's figures are)."""
    from efficientsam3_amd import Sam3Processor, build_efficientsam3_image_model
    from oracle import ref_model, ref_pcs
    man, g = pcs_gold
    gdir = os.path.join(os.path.dirname(__file__), "golden", "pcs_ev_m")
    with open(os.path.join(gdir, "bf16ref_manifest.json")) as f:
        fix = json.load(f)["cases"]
    with open(os.path.join(gdir, "bf16ref_draws.json")) as f:
        draws = json.load(f)
    with open(os.path.join(gdir, "bf16ref_geo.json")) as f:
        geo_yard = json.load(f)
    assert geo_yard["seeds"] == [1] + list(draws["seeds"]), (geo_yard["seeds"], draws["seeds"])
    seeds = ([1] + list(draws["seeds"]))[:PCS_DIST_IMAGES]   # every image costs four fp32 oracle runs of the detector on the host
    texts = man["prompts"]
    ref = {t: {k: ([fix[t][k]] + [d[k] for d in draws["cases"][t]])[:PCS_DIST_IMAGES] for k in PCS_OUT} for t in texts}
    ref.update({n: {k: [d[k] for d in geo_yard["cases"][n]][:PCS_DIST_IMAGES] for k in PCS_OUT} for n in man["geometric_cases"]})
    model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit",
                                            model_name="b1", dtype="bf16", state_dict=pcs_sd, text_encoder_type="MobileCLIP-S0",
                                            text_encoder_context_length=16)
    proc = Sam3Processor(model, confidence_threshold=man["confidence_threshold"])
    runs = [(t, f"{pi}", None) for pi, t in enumerate(texts)] + [(n, n, n) for n in man["geometric_cases"]]
    eng = {c: {k: [] for k in PCS_OUT} for c, _, _ in runs}
    for seed in seeds:
        img = synth.smooth_image_u8(seed=seed)
        state = proc.set_image(torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0))))
        fpn = state["backbone_out"]["_esam3_nhwc_sam3"]
        with torch.inference_mode():
            bo = ref_model.set_image(pcs_sd, torch.from_numpy(synth.normalise_to_chw_f32(img))[None], (1008, 1008), "b1")["backbone_out"]
        for case, key, geo_name in runs:
            lf, lm = torch.from_numpy(g[f"{key}_language_features"]), torch.from_numpy(g[f"{key}_language_mask"])
            geo = _geo_case(g, geo_name) if geo_name else None
            out = model.engine.ground(fpn, lf.to("cuda"), lm.to("cuda"), geo=geo)
            with torch.inference_mode():
                o = ref_pcs.forward_grounding(pcs_sd, bo["backbone_fpn"], bo["vision_pos_enc"][-1], lf, lm, None, geo)
            for k in PCS_OUT:
                eng[case][k].append(float((out[k].float().cpu() - o[k].float()).abs().max()))
    failures = []
    _pcs_verdicts("pcs ev_m bf16", eng, ref, failures)
    assert not failures, failures


@pytest.mark.gpu
def test_config4_batch_8_on_the_yardstick_pairs():
    """BASELINE config 4 at its full size (ViT-H + MobileCLIP-S0-16 + detector, batch 8): the eight (image, text) pairs are the ones the
    reference's own bf16 runs of this model were taken on (tests/golden/pcs_vit_h/bf16ref_draws.json: image seeds 2..5 x the prompts
    "dog" / "traffic light"), so the bf16 engine is held to the DISTRIBUTION rule on exactly the yardstick's inputs (median <= 1.25 x,
    max <= 1.5 x the reference's, per prompt and output; no flat factor).  The engine runs its own text encoder on the prompts' token
    ids.  f32: every pair within the f32 limits of test_pcs_engine_vs_golden against the ORACLE (oracle/ref_model.py ViT-H + text
    student, oracle/ref_pcs.py detector) run on the host.  Image independence: the two batch entries that share an image have
    bit-identical features, entries that share a prompt bit-identical text memory, and a batch of two of the pairs reproduces their
    batch-8 results (to a tenth of the f32 limits in f32)."""
    from efficientsam3_amd import build_sam3_image_model
    from oracle import ref_model, ref_pcs
    sd = schema.synthetic_state_dict("sam3", "vit_h", seed=0, enable_inst_interactivity=False)
    sd.update(schema.synthetic_text_state_dict("MobileCLIP-S0", 16, seed=0))
    sd.update(schema.synthetic_pcs_state_dict(seed=0))
    with open(os.path.join(os.path.dirname(__file__), "golden", "pcs_vit_h", "bf16ref_draws.json")) as f:
        draws = json.load(f)
    seeds, texts = list(draws["seeds"]), list(YARD_TOKENS)
    assert len(seeds) == 4 and set(draws["cases"]) == set(texts)
    base = [synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s_)) for s_ in seeds]
    pairs = [(i, t) for i in range(4) for t in texts]                    # batch entry 2 i + j = (image i, prompt j)
    tok = np.zeros((8, 16), dtype=np.int64)
    for e, (_, t) in enumerate(pairs):
        tok[e, :len(YARD_TOKENS[t])] = YARD_TOKENS[t]
    oracle = []
    with torch.inference_mode():
        feats = [ref_model.forward_image(sd, torch.from_numpy(base[i])[None], "vit_h") for i in range(4)]
        txt = {t: ref_model.text_encoder_student(sd, torch.from_numpy(tok[j:j + 1])) for j, t in enumerate(texts)}
        for i, t in pairs:
            mask, mem, _ = txt[t]
            o = ref_pcs.forward_grounding(sd, feats[i]["backbone_fpn"], feats[i]["vision_pos_enc"][-1], mem, mask)
            oracle.append({k: o[k].float().numpy() for k in PCS_OUT})
    ref = {t: {k: [d[k] for d in draws["cases"][t]] for k in PCS_OUT} for t in texts}
    f32_lim = dict(pred_logits=1e-4, pred_boxes=1e-4, presence_logit_dec=1e-4, pred_masks=4e-3)
    failures = []
    for mode in ("bf16", "f32"):
        model = build_sam3_image_model(device="cuda", enable_inst_interactivity=False, dtype=mode, state_dict=sd,
                                       text_encoder_type="MobileCLIP-S0", text_encoder_context_length=16)
        eng = model.engine
        x = torch.from_numpy(np.stack([base[i] for i, _ in pairs])).to("cuda")
        tok_d = torch.from_numpy(tok).to("cuda")
        out = eng.encode(x, want_sam3=True, want_sam2=False)
        mem, _ = eng.encode_text(tok_d)
        gr = eng.ground(out["sam3_fpn"], mem, tok_d == 0)
        for lvl in out["sam3_fpn"]:
            for i in range(4):
                assert torch.equal(lvl[2 * i], lvl[2 * i + 1]), (mode, i)
        for j in range(2):
            for i in range(1, 4):
                assert torch.equal(mem[:, j], mem[:, 2 * i + j]), (mode, i, j)
        # a batch of two of the pairs gives what they gave inside the batch of eight (f32: a tenth of the f32 limits); in bf16 the GEMM
        # tiling and the few-query attention's key split follow the batch size -- other summation orders, other bf16 roundings,
        # amplified through 12 layers (measured 0.006 on the class logits, half of the reference's own bf16 distance): it has to stay
        # below 3 / 4 of that distance, i.e. batch composition must matter less than the precision itself
        sub = [1, 6]
        gr2 = eng.ground([lvl[sub].contiguous() for lvl in out["sam3_fpn"]], mem[:, sub].contiguous(), (tok_d == 0)[sub].contiguous())
        for k in PCS_OUT:
            assert torch.isfinite(gr[k]).all(), (mode, k)
            d = float((gr2[k] - gr[k][sub]).abs().max())
            lim = 0.1 * f32_lim[k] if mode == "f32" else 0.75 * max(max(ref[t][k]) for t in texts) + (PRESENCE_BF16_ULP if k == "presence_logit_dec" else 0.0)
            print(f"[cfg4 {mode}] batch of 2 vs the same pairs in the batch of 8: {k} max-abs-diff {d:.3g} (allowed {lim:.3g})")
            assert d <= lim, (mode, k, d, lim)
        assert gr["pred_masks"].shape == (8, 200, 288, 288) and float(gr["pred_boxes"].min()) >= 0.0 and float(gr["pred_boxes"].max()) <= 1.0
        err = {t: {k: [] for k in PCS_OUT} for t in texts}
        for e, (i, t) in enumerate(pairs):
            for k in PCS_OUT:
                err[t][k].append(float(np.abs(gr[k][e].float().cpu().numpy() - oracle[e][k][0]).max()))
        if mode == "f32":
            for t in texts:
                for k in PCS_OUT:
                    print(f"[cfg4 f32] {t:14s} {k:20s} worst of 4 images {max(err[t][k]):.3g} (allowed {f32_lim[k]:.3g})")
                    if max(err[t][k]) > f32_lim[k]:
                        failures.append(("f32", t, k, max(err[t][k])))
        else:
            _pcs_verdicts("cfg4 vit_h bf16", err, ref, failures)
        del model, eng, out, gr, gr2, x
        torch.cuda.empty_cache()
    assert not failures, failures


# ---- the video path's multi-GPU entry around the REAL detector ---------------------------------------------
def _video_engine_worker(q):
    """One rank, RCCL ("nccl") group forced: `VideoGroundingMultiGPU` drives the engine (encode -> ground + SAM2 FPN in bf16) on
    device tensors through the same all-gather code a multi-rank job runs."""
    import torch.distributed as tdist
    from efficientsam3_amd import Sam3Processor, build_efficientsam3_image_model
    from efficientsam3_amd import dist as esdist
    try:
        torch.cuda.set_device(0)
        esdist.init_process_group("nccl", device=torch.device("cuda", 0), force=True)
        sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0, enable_inst_interactivity=True)
        sd.update(schema.synthetic_text_state_dict("MobileCLIP-S0", 16, seed=0))
        sd.update(schema.synthetic_pcs_state_dict(seed=0))
        model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=True, backbone_type="efficientvit",
                                                model_name="b1", dtype="bf16", state_dict=sd, text_encoder_type="MobileCLIP-S0",
                                                text_encoder_context_length=16)
        proc = Sam3Processor(model)
        # text features of the first golden prompt, as the REFERENCE's text encoder produced them (the tokenizer's BPE table is an asset
        # of the reference checkout and does not travel; tests/golden/pcs_ev_m/pcs_cases.npz)
        gp = np.load(os.path.join(os.path.dirname(__file__), "golden", "pcs_ev_m", "pcs_cases.npz"))
        text = {"language_features": torch.from_numpy(gp["0_language_features"]).to("cuda"),
                "language_mask": torch.from_numpy(gp["0_language_mask"]).to("cuda")}
        frames = [torch.from_numpy(np.ascontiguousarray(np.moveaxis(synth.smooth_image_u8(seed=20 + t), -1, 0))) for t in range(3)]
        direct, calls = {}, []

        def detect(t):
            calls.append(t)
            bo = proc.set_image(frames[t])["backbone_out"]
            bo.update(text)
            out = dict(model.forward_grounding(bo, geometric_prompt=model._get_dummy_prompt()))
            from efficientsam3_amd import box_ops
            out["pred_boxes_xyxy"] = box_ops.box_cxcywh_to_xyxy(out["pred_boxes"])
            s2 = bo["sam2_backbone_out"]
            direct[t] = (out, [x.to(torch.bfloat16) for x in s2["backbone_fpn"]])
            return out, s2["backbone_fpn"], s2["vision_pos_enc"]

        v = esdist.VideoGroundingMultiGPU(detect, force_collective=True)
        buf, ok, n_buf = {}, True, []
        for t in range(3):
            out = v.forward(t, 3, buf, return_sam2_backbone_feats=True)
            torch.cuda.synchronize()
            ref_out, ref_fpn = direct[t]
            for k in ("pred_logits", "pred_boxes", "pred_boxes_xyxy", "pred_masks"):
                ok &= bool(torch.equal(out[k], ref_out[k])) and out[k].is_cuda
            for i in range(3):
                g = out[f"tracker_backbone_fpn_{i}"]
                ok &= g.dtype == torch.bfloat16 and bool(torch.equal(g, ref_fpn[i])) and tuple(g.shape) == tuple(ref_fpn[i].shape)
            ok &= "tracker_backbone_pos_enc" in out
            n_buf.append(len(buf))
        q.put({"ok": bool(ok), "calls": calls, "chunks": v.chunks_built, "n_buf": n_buf, "backend": tdist.get_backend(),
               "finite": bool(all(torch.isfinite(direct[t][0]["pred_masks"]).all() for t in direct))})
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put({"ok": False, "error": repr(e), "trace": traceback.format_exc()})
    finally:
        if tdist.is_initialized():
            tdist.destroy_process_group()


@pytest.mark.gpu
def test_video_grounding_entry_drives_the_engine():
    """`dist.VideoGroundingMultiGPU` (the reference's `forward_video_grounding_multigpu`, sam3/sam3/model/sam3_image.py:701-883; its
    bookkeeping is pinned against the reference's own trace by tests/test_dist_gloo.py) around the REAL detector: three frames
    through `set_image` -> `forward_grounding` + the SAM2 FPN cast to bf16, all-gathered over RCCL in a one-rank group.  Every
    frame read back from the buffer equals the detector's direct output bit for bit, the chunks are built one call ahead."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_video_engine_worker, args=(q,))
    p.start()
    rep = q.get(timeout=600)
    p.join(timeout=60)
    assert rep.get("ok"), {k: (v if k != "trace" else v[-1500:]) for k, v in rep.items()}
    assert rep["backend"] == "nccl" and rep["finite"]
    assert rep["calls"] == [0, 1, 2] and rep["chunks"] == [(0, 1), (1, 2), (2, 3)] and max(rep["n_buf"]) <= 2
