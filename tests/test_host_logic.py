"""CPU: host-side logic of the drop-in (no GPU, no compute through the HIP library):
prompt preparation, position encoding, checkpoint key remapping, schema, and the loud failure
of the product path when no HIP device is present."""
import numpy as np
import pytest
import torch

from efficientsam3_amd import model_builder, schema, synth
from efficientsam3_amd.sam3_image import Sam3Image, _sine_position_encoding
from oracle import ref_model


def test_sine_position_encoding_matches_oracle():
    for h, w in ((72, 72), (144, 144), (5, 9)):
        mine = _sine_position_encoding(h, w)
        ref = ref_model.position_embedding_sine(h, w).reshape(256, h, w).numpy()
        assert mine.shape == ref.shape == (256, h, w)
        assert np.abs(mine - ref).max() <= 2e-6


@pytest.mark.parametrize("orig_hw", [(1008, 1008), (600, 800)])
def test_prep_prompts_follows_reference_rules(orig_hw):
    """sam1_task_predictor.py:298-326 + sam1_utils.py:47-75: px -> /orig -> *1008; box corners get
    labels 2,3 and are placed BEFORE the points."""
    h, w = orig_hw
    pts = np.array([[100.0, 50.0], [10.5, 20.25]], np.float32)
    lab = np.array([1, 0])
    box = np.array([30.0, 40.0, 300.0, 400.0], np.float32)
    c, l = Sam3Image._prep_prompts(pts, lab, box, True, orig_hw)
    assert c.shape == (1, 4, 2) and l.shape == (1, 4) and c.dtype == np.float32 and l.dtype == np.int32
    s = np.array([1008.0 / w, 1008.0 / h], np.float32)
    np.testing.assert_allclose(c[0, :2], box.reshape(2, 2) * s, rtol=1e-6)
    np.testing.assert_allclose(c[0, 2:], pts * s, rtol=1e-6)
    assert l.tolist() == [[2, 3, 1, 0]]
    # K boxes -> Bp = K prompts, no points
    boxes = np.array([[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12]], np.float32)
    c, l = Sam3Image._prep_prompts(None, None, boxes, True, orig_hw)
    assert c.shape == (3, 2, 2) and l.tolist() == [[2, 3]] * 3
    # normalize_coords=False: coordinates are already in [0,1] and only scaled by 1008
    c, l = Sam3Image._prep_prompts(np.array([[0.5, 0.25]]), np.array([1]), None, False, orig_hw)
    np.testing.assert_allclose(c, [[[504.0, 252.0]]])
    # nothing given
    assert Sam3Image._prep_prompts(None, None, None, True, orig_hw) == (None, None)
    with pytest.raises(AssertionError):
        Sam3Image._prep_prompts(pts, None, None, True, orig_hw)
    # the inputs are not modified in place
    assert pts[0, 0] == 100.0 and box[0] == 30.0


def test_checkpoint_key_remap():
    """model_builder.py:584-630: 'detector.' and 'student_trunk.' are stripped; tracker.* is
    duplicated under inst_interactive_predictor.model.* when interactivity is on."""
    t = torch.zeros(1)
    ck = {"model": {"detector.backbone.vision_backbone.trunk.model.student_trunk.backbone.x": t,
                    "detector.backbone.vision_backbone.convs.0.conv_1x1.weight": t,
                    "tracker.sam_mask_decoder.iou_token.weight": t}}
    out = model_builder._clean_checkpoint_keys(ck, interactive=True)
    assert "backbone.vision_backbone.trunk.model.backbone.x" in out
    assert "backbone.vision_backbone.convs.0.conv_1x1.weight" in out
    assert "inst_interactive_predictor.model.sam_mask_decoder.iou_token.weight" in out
    out = model_builder._clean_checkpoint_keys(ck, interactive=False)
    assert not any(k.startswith("inst_interactive_predictor") for k in out)


def test_size_aliases_and_schema_names():
    assert model_builder.SIZE_ALIASES["efficientvit"]["m"] == "b1"
    assert model_builder.SIZE_ALIASES["repvit"]["m"] == "m1.1"
    assert model_builder.SIZE_ALIASES["tinyvit"]["m"] == "11m"
    sch = dict(schema.image_path_schema("efficientvit", "b1", True))
    # a few names the reference's checkpoints carry (model_builder.py:764-787, necks.py:13-125)
    for k in ("backbone.vision_backbone.trunk.model.head.0.weight",
              "backbone.vision_backbone.sam2_convs.0.dconv_2x2_0.weight",
              "inst_interactive_predictor.model.sam_mask_decoder.conv_s0.weight",
              "inst_interactive_predictor.model.sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"):
        assert k in sch, k


def test_synthetic_inputs_are_deterministic():
    a, b = synth.smooth_image_u8(seed=3), synth.smooth_image_u8(seed=3)
    assert a.dtype == np.uint8 and a.shape == (1008, 1008, 3) and np.array_equal(a, b)
    x = synth.normalise_to_chw_f32(a)
    assert x.shape == (3, 1008, 1008) and x.dtype == np.float32 and -1.0 <= x.min() and x.max() <= 1.0
    p1, p2 = synth.prompts(4, seed=2), synth.prompts(4, seed=2)
    assert all(np.array_equal(u, v) for u, v in zip(p1, p2))


def test_product_path_fails_loudly_without_hip_device():
    """No CPU fallback: on a box without a GPU building the model must raise, not degrade."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception) as ei:
        model_builder.build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=True,
                                                      backbone_type="efficientvit", model_name="b1")
    assert not isinstance(ei.value, NotImplementedError)
    with pytest.raises(RuntimeError):
        model_builder.build_efficientsam3_image_model(device="cpu")


def test_sam3_import_facade():
    """<repo>/compat on PYTHONPATH gives the reference's import surface."""
    import importlib
    import os
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat")
    saved = {k: v for k, v in sys.modules.items() if k == "sam3" or k.startswith("sam3.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, compat)
    try:
        sam3 = importlib.import_module("sam3")
        proc = importlib.import_module("sam3.model.sam3_image_processor")
        dev = importlib.import_module("sam3.device")
        mb = importlib.import_module("sam3.model_builder")
        import efficientsam3_amd
        assert sam3.build_efficientsam3_image_model is efficientsam3_amd.build_efficientsam3_image_model
        assert mb.build_sam3_image_model is efficientsam3_amd.build_sam3_image_model
        assert proc.Sam3Processor is efficientsam3_amd.Sam3Processor
        assert callable(dev.get_device)
    finally:
        sys.path.remove(compat)
        for k in [k for k in sys.modules if k == "sam3" or k.startswith("sam3.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_checkpoint_merge_and_clean_round_trip(tmp_path):
    """The stage-1 converters' merge format (convert_both_encoders_weights_stage1.py:106-152) followed by the
    loader's key rules (model_builder.py:584-630) gives back the model's own key names; a checkpoint written that
    way loads through ``checkpoint_path`` parsing."""
    from efficientsam3_amd import checkpoint as ck
    full = schema.synthetic_state_dict("efficientvit", "b0", seed=0, enable_inst_interactivity=True)
    full.update(schema.synthetic_text_state_dict("MobileCLIP-S0", 16, seed=0))
    img_p, txt_p = "backbone.vision_backbone.trunk.model.", "backbone.language_backbone."
    image_sd = {"module.student_trunk." + k[len(img_p):]: v for k, v in full.items() if k.startswith(img_p)}
    text_sd = {"module." + k[len(txt_p):]: v for k, v in full.items() if k.startswith(txt_p)}
    assert image_sd and text_sd
    inter_p = "inst_interactive_predictor.model."
    teacher = {}
    for k, v in full.items():
        if k.startswith(inter_p):
            teacher["tracker." + k[len(inter_p):]] = v
        elif k.startswith(img_p) or k.startswith(txt_p):
            continue
        else:
            teacher["detector." + k] = v
    teacher["detector.backbone.vision_backbone.trunk.blocks.0.attn.qkv.weight"] = torch.zeros(1)   # ViT-H: replaced
    teacher["detector.backbone.language_backbone.encoder.transformer.resblocks.0.ln_1.weight"] = torch.zeros(1)
    payload = ck.merge_student_checkpoints(teacher, image_sd, text_sd, text_context_length=16)
    assert payload["meta"]["text_context_length"] == 16
    assert not any("blocks.0.attn.qkv" in k or "resblocks" in k for k in payload["model"])
    assert all(k.startswith(("detector.", "tracker.")) for k in payload["model"])
    path = tmp_path / "merged.pth"
    torch.save(payload, path)
    sd = ck.clean_checkpoint_keys({"model": ck.load_state_dict_file(str(path))}, interactive=True)
    missing = [k for k in full if k not in sd]
    assert not missing, missing[:5]
    assert all(torch.equal(sd[k], full[k]) for k in full)
    # the single-encoder converters: only one subtree replaced
    only_img = ck.merge_student_checkpoints(teacher, image_sd=image_sd)["model"]
    assert "detector.backbone.language_backbone.encoder.transformer.resblocks.0.ln_1.weight" in only_img
    assert ck.normalize_image_student_key("detector.backbone.vision_backbone.trunk.model.backbone.x") == "backbone.x"
    assert ck.extract_state_dict({"state_dict": {"a": torch.zeros(1)}}).keys() == {"a"}
    with pytest.raises(ValueError):
        ck.extract_state_dict({"a": 1})


def test_box_ops_match_the_reference_formulas():
    """sam3.model.box_ops of the facade: the six format conversions are mutually inverse and agree with the reference's
    definitions (box_ops.py:11-45) on hand-computed values; pairwise IoU on a known pair."""
    import os
    import sys
    from efficientsam3_amd import box_ops as bo
    b = torch.tensor([[10.0, 20.0, 30.0, 40.0], [0.25, 0.5, 0.5, 0.25]])          # XYWH
    assert torch.equal(bo.box_xywh_to_xyxy(b), torch.tensor([[10.0, 20.0, 40.0, 60.0], [0.25, 0.5, 0.75, 0.75]]))
    assert torch.equal(bo.box_xywh_to_cxcywh(b), torch.tensor([[25.0, 40.0, 30.0, 40.0], [0.5, 0.625, 0.5, 0.25]]))
    g = torch.rand((3, 5, 4), generator=torch.Generator().manual_seed(0)) + 0.1
    for f, inv in ((bo.box_xywh_to_xyxy, bo.box_xyxy_to_xywh), (bo.box_xywh_to_cxcywh, bo.box_cxcywh_to_xywh),
                   (bo.box_cxcywh_to_xyxy, bo.box_xyxy_to_cxcywh)):
        assert torch.allclose(inv(f(g)), g, atol=1e-6)
    iou, union = bo.box_iou(torch.tensor([[0.0, 0.0, 2.0, 2.0]]), torch.tensor([[1.0, 1.0, 3.0, 3.0], [5.0, 5.0, 6.0, 6.0]]))
    assert torch.allclose(iou, torch.tensor([[1.0 / 7.0, 0.0]])) and torch.allclose(union, torch.tensor([[7.0, 5.0]]))
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat")
    sys.path.insert(0, compat)
    try:
        for m in [k for k in sys.modules if k == "sam3" or k.startswith("sam3.")]:
            del sys.modules[m]
        from sam3.model.box_ops import box_xywh_to_cxcywh
        assert box_xywh_to_cxcywh is bo.box_xywh_to_cxcywh
    finally:
        sys.path.remove(compat)
        for m in [k for k in sys.modules if k == "sam3" or k.startswith("sam3.")]:
            del sys.modules[m]


def _with_reference():
    """Context for importing the REAL reference where it exists (the build container); None elsewhere."""
    import contextlib
    import os
    import sys
    ref_root = "/root/reference/sam3"
    if not os.path.isdir(ref_root):
        return None
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    added = [os.path.join(repo, "oracle", "shims"), ref_root]

    @contextlib.contextmanager
    def ctx():
        purge = lambda: [sys.modules.pop(k) for k in list(sys.modules) if k == "sam3" or k.startswith("sam3.")]
        purge()
        sys.path[:0] = added
        try:
            yield
        finally:
            for p_ in added:
                sys.path.remove(p_)
            purge()
    return ctx()


def test_prompt_coordinates_match_reference_transforms():
    """Row P2: Sam3Image._prep_prompts == SAM2Transforms.transform_coords / transform_boxes
    (sam3/sam3/model/utils/sam1_utils.py:47-75) for random points, boxes and original sizes."""
    ctx = _with_reference()
    if ctx is None:
        pytest.skip("the reference is not present on this machine")
    from efficientsam3_amd.sam3_image import Sam3Image
    with ctx:
        from sam3.model.utils.sam1_utils import SAM2Transforms
        tr = SAM2Transforms(resolution=1008, mask_threshold=0.0, max_hole_area=0.0, max_sprinkle_area=0.0)
        rng = np.random.default_rng(0)
        for _ in range(25):
            h, w = int(rng.integers(50, 3000)), int(rng.integers(50, 3000))
            n = int(rng.integers(1, 6))
            pts = (rng.random((n, 2)) * [w, h]).astype(np.float32)
            lab = rng.integers(0, 2, n).astype(np.int32)
            box = np.sort(rng.random((2, 2)) * [w, h], axis=0).reshape(4).astype(np.float32)
            want_pts = tr.transform_coords(torch.from_numpy(pts)[None], normalize=True, orig_hw=(h, w))[0].numpy()
            want_box = tr.transform_boxes(torch.from_numpy(box)[None], normalize=True, orig_hw=(h, w)).reshape(-1, 2, 2)[0].numpy()
            coords, labels = Sam3Image._prep_prompts(pts, lab, box, True, (h, w))
            # the box corners come first with labels 2 / 3 (sam1_task_predictor.py:385-396), then the points
            assert coords.shape == (1, 2 + n, 2) and labels.tolist() == [[2, 3] + lab.tolist()]
            np.testing.assert_allclose(coords[0, :2], want_box, rtol=1e-6, atol=1e-4)
            np.testing.assert_allclose(coords[0, 2:], want_pts, rtol=1e-6, atol=1e-4)


def test_geometry_prompt_matches_reference_prompt_class():
    """geometry_prompt.Prompt == the reference's Prompt (geometry_encoders.py:82-400) for random append sequences with
    per-image padding: same embeddings, labels and masks after every step."""
    ctx = _with_reference()
    if ctx is None:
        pytest.skip("the reference is not present on this machine")
    from efficientsam3_amd.geometry_prompt import Prompt as Mine
    with ctx:
        from sam3.model.geometry_encoders import Prompt as Ref
        rng = np.random.default_rng(1)
        for trial in range(10):
            b = int(rng.integers(1, 4))
            kw = dict(box_embeddings=torch.zeros(0, b, 4), box_mask=torch.zeros(b, 0, dtype=torch.bool))
            mine, ref = Mine(**kw), Ref(**kw)
            for step in range(int(rng.integers(1, 6))):
                boxes = torch.from_numpy(rng.random((1, b, 4)).astype(np.float32))
                labels = torch.from_numpy(rng.integers(0, 2, (1, b))).bool()
                mask = torch.from_numpy(rng.random((b, 1)) < 0.35)
                mine.append_boxes(boxes, labels, mask=mask.clone())
                ref.append_boxes(boxes.clone(), labels.clone(), mask=mask.clone())
                valid = ~ref.box_mask
                assert torch.equal(mine.box_mask, ref.box_mask), (trial, step)
                assert torch.equal(mine.box_embeddings.transpose(0, 1)[valid], ref.box_embeddings.transpose(0, 1)[valid])
                assert torch.equal(mine.box_labels.transpose(0, 1)[valid].long(), ref.box_labels.transpose(0, 1)[valid].long())


def test_full_training_checkpoint_loads_without_unpickling_foreign_code(tmp_path):
    """stage1/utils.py:287-293 saves {"model", "optimizer", "scaler", "config" (a yacs CfgNode), ...}; the reference's
    converters load it with weights_only=False.  load_state_dict_file must ingest such a file -- the foreign classes are
    not importable here -- WITHOUT executing pickled code, and `trusted=True` is the explicit opt-in to a full unpickle."""
    import sys
    import types

    import torch

    from efficientsam3_amd import checkpoint
    pkg, mod = types.ModuleType("yacs_like"), types.ModuleType("yacs_like.config")

    class CfgNode(dict):
        pass

    CfgNode.__module__, CfgNode.__qualname__ = "yacs_like.config", "CfgNode"
    mod.CfgNode = CfgNode
    sys.modules["yacs_like"], sys.modules["yacs_like.config"] = pkg, mod
    marker = tmp_path / "executed"

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, (f"touch {marker}",))

    # foreign container subclasses next to the weights: a list, a set and a slotted object replay APPENDS / ADDITEMS /
    # a (dict, slots) BUILD state on the stand-in
    class History(list):
        pass

    class Seen(set):
        pass

    class Slotted:
        __slots__ = ("a", "b")

        def __init__(self):
            self.a, self.b = 1, [2, 3]

    for cls in (History, Seen, Slotted):
        cls.__module__, cls.__qualname__ = "yacs_like.config", cls.__name__
        setattr(mod, cls.__name__, cls)

    try:
        ck = {"model": {"student_trunk.w": torch.arange(4.0), "b": torch.ones(2, dtype=torch.bfloat16)},
              "optimizer": {"state": {0: {"exp_avg": torch.zeros(2)}}, "param_groups": [{"lr": 1e-3}]},
              "config": CfgNode(a=1, nested=CfgNode(c="x")), "epoch": 3, "hook": Evil(),
              "history": History([1.0, 2.0, 3.0]), "seen": Seen({"a", "b"}), "slotted": Slotted()}
        path = str(tmp_path / "ckpt_epoch_3.pth")
        torch.save(ck, path)
    finally:
        del sys.modules["yacs_like"], sys.modules["yacs_like.config"]
    sd = checkpoint.load_state_dict_file(path)
    assert sorted(sd) == ["b", "student_trunk.w"] and torch.equal(sd["student_trunk.w"], torch.arange(4.0))
    assert not marker.exists(), "pickled code was executed"
    assert "yacs_like" not in sys.modules
    plain = str(tmp_path / "plain.pth")
    torch.save({"w": torch.zeros(3)}, plain)
    assert list(checkpoint.load_state_dict_file(plain)) == ["w"]


def test_bpe_path_is_resolved_at_build_time(tmp_path, monkeypatch):
    """model_builder.resolve_bpe_path: an explicit path that does not exist raises at BUILD time (the reference fails in
    its tokenizer constructor too, model_builder.py:676-692), $ESAM3_BPE_PATH is honoured, nothing found -> None."""
    from efficientsam3_amd import model_builder
    with pytest.raises(FileNotFoundError):
        model_builder.resolve_bpe_path(str(tmp_path / "missing.txt.gz"))
    f = tmp_path / "bpe.txt.gz"
    f.write_bytes(b"x")
    assert model_builder.resolve_bpe_path(str(f)) == str(f)
    monkeypatch.setenv("ESAM3_BPE_PATH", str(f))
    assert model_builder.resolve_bpe_path(None) == str(f)
    monkeypatch.delenv("ESAM3_BPE_PATH")
    got = model_builder.resolve_bpe_path(None)
    assert got is None or got.endswith("bpe_simple_vocab_16e6.txt.gz")
    if got is None:
        with pytest.raises(FileNotFoundError):
            model_builder.resolve_bpe_path(None, required=True)


def test_rle_string_decoder_rejects_malformed_input():
    """esam3_rle_from_string: characters outside cocoapi's alphabet and endless continuation runs are errors, not
    undefined shifts."""
    import ctypes as C

    from efficientsam3_amd import _lib
    lib = _lib.load()  # signature declared in _lib.py: (void*, int64, void*, int64) -> int64
    buf = (C.c_uint32 * 16)()

    def dec(b):
        return int(lib.esam3_rle_from_string(C.cast(C.c_char_p(b), C.c_void_p), len(b), C.cast(buf, C.c_void_p), 16))

    assert dec(b"52203") == 5                 # five single-character counts
    assert dec(b"\x10\x11") == -1             # below '0'
    assert dec(b"o" * 20 + b"0") == -1        # 'o' = continuation bit set, 20 times
    assert dec(b"o") == -1                    # truncated


def test_selection_tie_fixtures_and_matching_rule():
    """tests/util.py: errors_with_ties -- a single-mask prompt may match a recorded alternative of the reference ONLY if the
    tie manifest lists it, and only within the same limits; the committed manifests are consistent with their arrays."""
    import json
    import os
    from tests import util as U
    root = os.path.join(os.path.dirname(__file__), "golden")
    n_alt = 0
    for d in ("", "repvit_m1.1", "repvit_m2.3", "tinyvit_11m", "sam3_vit_h"):
        cases, arr = U.load_ties(os.path.join(root, d))
        assert cases is not None
        for name, c in cases.items():
            assert len(c["stability_mask0"]) == len(c["selected"]) == len(c["iou_pred"])
            for i, alts in c["alternatives"].items():
                st, sel, iou = c["stability_mask0"][int(i)], c["selected"][int(i)], c["iou_pred"][int(i)]
                assert sel not in alts and (abs(st - 0.98) < 5e-3 or st < 0.98)
                for k in alts:
                    assert arr[f"{name}/alt_low_res/{i}/{k}"].shape == (288, 288)
                    assert arr[f"{name}/alt_mask_bits/{i}/{k}"].dtype == np.uint8 and arr[f"{name}/alt_mask_bits/{i}/{k}"].size >= 480 * 640 // 8
                    assert k == 0 or iou[k] >= max(iou[1:]) - 1e-2
                    n_alt += 1
    assert n_alt >= 4
    # the matching rule on synthetic data: prompt 1 of a 2-prompt case has one alternative (mask 3)
    rng = np.random.default_rng(0)
    g_low = rng.standard_normal((2, 1, 8, 8)).astype(np.float32)
    g_iou = np.array([[0.5], [0.6]], np.float32)
    alt = rng.standard_normal((8, 8)).astype(np.float32)
    ties = ({"case": {"alternatives": {"1": [3]}}}, {"case/alt_low_res/1/3": alt, "case/alt_iou/1/3": np.float32(0.61)})
    low = g_low.copy()
    low[1, 0] = alt + 0.01
    iou = np.array([[0.5], [0.612]], np.float32)
    e_low, e_iou, flipped = U.errors_with_ties("case", low, iou, g_low, g_iou, 0.05, 0.01, ties)
    assert flipped == {1: 3} and e_low <= 0.0101 and e_iou <= 0.0021
    # the mask of a prompt that took an alternative is compared with THAT candidate's mask, not dropped
    ref_bits = np.zeros((2, 1, 4, 4), bool)
    alt_bits = np.ones(16, bool)
    ties_m = (ties[0], dict(ties[1], **{"case/alt_mask_bits/1/3": np.packbits(alt_bits)}))
    rb = U.tie_reference_bits("case", ref_bits, flipped, ties_m)
    assert rb[1].all() and not rb[0].any() and not ref_bits.any()
    assert U.tie_reference_bits("case", ref_bits, {}, ties_m) is ref_bits
    e_low, _, flipped = U.errors_with_ties("case", low, iou, g_low, g_iou, 0.05, 0.01, (None, None))       # f32 mode: no allowance
    assert flipped == {} and e_low > 0.5
    low[0, 0] = alt                                                                                       # prompt 0 has no alternative listed
    e_low, _, flipped = U.errors_with_ties("case", low, iou, g_low, g_iou, 0.05, 0.01, ties)
    assert flipped == {1: 3} and e_low > 0.5
    low[0, 0] = g_low[0, 0]
    low[1, 0] = alt + 0.2                                                                                 # outside the limit: no match
    e_low, _, flipped = U.errors_with_ties("case", low, iou, g_low, g_iou, 0.05, 0.01, ties)
    assert flipped == {} and e_low > 0.5
    # per-case score limit: 1.5 x the case's own reference-bf16 distance + half a bf16 ulp at the score
    yard = {"cases": {"a": {"low_res": 0.2, "iou": 1e-3, "mask_iou": 0.99}, "b": {"low_res": 0.3, "iou": 3e-3, "mask_iou": 0.98}}}
    la, lb = U.bf16_case_limits(yard, "a", score_peak=0.7), U.bf16_case_limits(yard, "b", score_peak=0.3)
    assert abs(la[1] - (1.5e-3 + 2.0 ** -9)) < 1e-12 and abs(lb[1] - (4.5e-3 + 2.0 ** -10)) < 1e-12 and la[1] < lb[1]
    # the yardstick as a distribution: the worst of the fixture image's draw and the further draws, selection flips left out
    draws = {"cases": {"a": {"low_res": [0.25, 12.7, 0.18], "iou": [2e-3, 0.15, 5e-4], "mask_iou": [0.985, 0.69, 0.995]}}}
    ya = U.bf16_case_yard(yard, "a", draws)
    assert ya == {"low_res": 0.25, "iou": 2e-3, "mask_iou": 0.985, "n_draws": 3}          # the 12.7 draw is a flip (> 5 x 0.2)
    assert U.bf16_case_yard(yard, "b", draws) == {"low_res": 0.3, "iou": 3e-3, "mask_iou": 0.98, "n_draws": 1}
    assert U.BF16_EXCEPTIONS == {}


def test_host_result_buffers_are_never_reused_while_referenced():
    """predict_inst_batch's host result pool (sam3_image._pool_get, used by _d2h_begin for masks, low-res logits and scores): a buffer is
    handed out again only when no view of it is alive -- ONE live view (a single-image group of a batch) protects it; an
    `out = step()` loop settles on two buffers; at most three are kept."""
    import torch
    from efficientsam3_amd import sam3_image as S

    def get(pool, shape):
        return S._pool_get(pool, lambda: torch.empty(shape))[1]

    pool = []
    a = get(pool, (2, 3))[0]
    b = get(pool, (2, 3))[0]
    assert a.base is not b.base                      # one live view keeps its buffer out of circulation
    del a
    c = get(pool, (2, 3))[0]
    assert c.base is pool[0][1]                      # released -> handed out again
    del b, c
    seen, out = set(), None
    for _ in range(6):
        out = [get(pool, (2, 3))[0]]
        seen.add(id(out[0].base))
    assert len(seen) == 2 and len(pool) <= 3
    kept = [get(pool, (2, 3)) for _ in range(5)]     # a caller that keeps everything: fresh buffers every time, none shared
    assert len({id(k) for k in kept}) == 5 and len(pool) <= 3
    # the engine-double path of the hand-back (a host tensor): begin / end return the values, uint8 widened to float32
    m = S.Sam3Image.__new__(S.Sam3Image)
    h = m._d2h_begin(torch.tensor([[0, 1], [1, 0]], dtype=torch.uint8))
    got = m._d2h_end(h)
    assert got.dtype == np.float32 and got.tolist() == [[0.0, 1.0], [1.0, 0.0]]


def test_pil_rgbx_view_is_the_image_and_staging_falls_back():
    """Sam3Processor._rgbx_view: the zero-copy export of a PIL "RGB" image's storage holds exactly the image's pixels in its first
    three bytes; images of other modes (and a Pillow / pyarrow that cannot export) take the packed 3-byte path; the staged batch says
    which layout it carries."""
    from PIL import Image
    from efficientsam3_amd.sam3_image_processor import Sam3Processor
    import types
    rng = np.random.default_rng(3)
    arr = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    im = Image.fromarray(arr)
    v = Sam3Processor._rgbx_view(im)
    if v is not None:
        assert v.shape == (37, 53, 4) and v.dtype == np.uint8 and np.array_equal(v[..., :3], arr)
    assert Sam3Processor._rgbx_view(Image.fromarray(arr[..., 0])) is None          # mode "L"
    assert Sam3Processor._rgbx_view(im.convert("RGBA")) is None
    proc = Sam3Processor(types.SimpleNamespace(device=torch.device("cpu")), device="cpu")
    ims = [Image.fromarray(rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)) for _ in range(5)]
    packed = proc._stage_pil_batch(ims, rgbx=False)
    assert tuple(packed.shape) == (5, 37, 53, 3)
    wide = proc._stage_pil_batch(ims, rgbx=True)
    assert wide.shape[-1] == (4 if v is not None else 3) and torch.equal(wide[..., :3], packed)
    mixed = proc._stage_pil_batch(ims[:4] + [ims[4].convert("L")], slot=1, rgbx=True)  # one image cannot export: whole batch packs
    assert mixed.shape[-1] == 3 and torch.equal(mixed[:4], packed[:4])


@pytest.mark.parametrize("n_in,n_out", [(32, 72), (16, 16), (9, 20), (64, 192), (20, 9), (1, 3), (33, 34)])
def test_resize_axis_tables_match_aten(n_in, n_out):
    """the per-axis maps the row-persistent resize_shuffle kernel gets from its launcher (csrc/kernels_backbone.hip: rs_build_tables) against
    torch's own bilinear resize (align_corners=False, the head's F.interpolate) on the CPU: every output index belongs to exactly one source
    cell's run, and (1 - frac) x[cell] + frac x[cell + 1] reproduces F.interpolate on random signals"""
    import ctypes as C

    import torch.nn.functional as F

    from efficientsam3_amd import _lib
    lib = _lib.load()
    first, count, frac = (C.c_int * n_in)(), (C.c_int * n_in)(), (C.c_float * n_out)()
    _lib.check(lib.esam3_resize_axis_tables(n_in, n_out, first, count, frac), "esam3_resize_axis_tables")
    owner = np.full(n_out, -1)
    for c in range(n_in):
        assert 0 <= count[c] <= 4
        for o in range(first[c], first[c] + count[c]):
            assert owner[o] == -1
            owner[o] = c
    assert (owner >= 0).all() and (np.diff(owner) >= 0).all()
    x = torch.randn(3, n_in, generator=torch.Generator().manual_seed(n_in * 131 + n_out))
    ref = F.interpolate(x[None, :, :, None], size=(n_out, 1), mode="bilinear", align_corners=False)[0, :, :, 0].numpy()
    fr = np.asarray(list(frac), dtype=np.float32)
    nxt = np.minimum(owner + 1, n_in - 1)
    got = (1.0 - fr) * x.numpy()[:, owner] + fr * x.numpy()[:, nxt]
    assert np.abs(got - ref).max() <= 5e-6      # the blend itself is rounded differently (fused multiply-adds in ATen's vectorised kernel)
    assert lib.esam3_resize_axis_tables(65, 72, first, count, frac) != 0     # more source cells than the kernel's tables hold


def test_widen_into_matches_numpy_on_every_split():
    """the host-side widening of the uint8 masks into the float32 result array (sam3_image._widen_into): serial and worker-thread paths,
    sizes that do not divide by the number of parts, float32 -> float32 for return_logits"""
    from efficientsam3_amd.sam3_image import _widen_into
    rng = np.random.default_rng(0)
    for shape, parts, below in (((3, 5, 7), 16, 1 << 22), ((2, 1, 257, 129), 16, 1000), ((1, 1, 1000, 1003), 7, 1000), ((5,), 16, 1)):
        src = rng.integers(0, 2, shape, dtype=np.uint8)
        dst = np.full(shape, -1.0, np.float32)
        _widen_into(dst, src, parts=parts, serial_below=below)
        assert dst.dtype == np.float32 and np.array_equal(dst, src.astype(np.float32))
        srcf = rng.standard_normal(shape).astype(np.float32)
        _widen_into(dst, srcf, parts=parts, serial_below=below)
        assert np.array_equal(dst, srcf)


def test_bench_cpulist_parser():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_cpulist("0-3,8-9\n") == {0, 1, 2, 3, 8, 9}
    assert bench.parse_cpulist("5") == {5} and bench.parse_cpulist("") == set()
    assert bench.bind_to_gpu_numa_node(0) is None or isinstance(bench.bind_to_gpu_numa_node(0), str)   # no GPU here: unbound, no exception
