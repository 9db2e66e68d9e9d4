"""MobileCLIP-S0 student text encoder.

CPU: the oracle restatement reproduces the REAL reference's outputs (fixtures written by
oracle/gen_golden_text.py) and the product tokenizer reproduces the reference tokenizer's ids
(needs the reference's BPE merge table: ESAM3_BPE_PATH or /root/reference/...; skipped elsewhere).
GPU: the HIP engine (esam3_encode_text through the C ABI) vs the same fixtures, from token ids."""
import json
import os

import numpy as np
import pytest
import torch

from efficientsam3_amd import schema

BPE_CANDIDATES = [os.environ.get("ESAM3_BPE_PATH"), "/root/reference/sam3/assets/bpe_simple_vocab_16e6.txt.gz"]


@pytest.fixture(scope="module")
def text_gold(golden_dir):
    d = os.path.join(golden_dir, "text_s0")
    with open(os.path.join(d, "manifest.json")) as f:
        man = json.load(f)
    return man, np.load(os.path.join(d, "text_cases.npz"))


@pytest.fixture(scope="module")
def text_sd(text_gold):
    return schema.synthetic_text_state_dict("MobileCLIP-S0", text_gold[0]["context_length"], seed=0)


def test_text_oracle_pinned_and_reproduces_golden(text_gold, text_sd):
    import hashlib
    from oracle import ref_model
    man, g = text_gold
    assert man["oracle_vs_reference_maxabs"]["memory"] <= 1e-5 and man["oracle_vs_reference_maxabs"]["mask"] == 0
    h = hashlib.sha256()
    for k, v in text_sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.numpy()).tobytes())
    assert h.hexdigest() == man["weights_sha256"]
    with torch.inference_mode():
        mask, mem, emb = ref_model.text_encoder_student(text_sd, torch.from_numpy(g["ids_ctx16"]))
    assert np.array_equal(mask.numpy(), g["mask"])
    assert float(np.abs(mem.numpy() - g["memory"]).max()) <= 1e-5
    assert float(np.abs(emb.numpy() - g["embeds"]).max()) <= 1e-6


def test_tokenizer_matches_reference_ids(text_gold):
    bpe = next((p for p in BPE_CANDIDATES if p and os.path.exists(p)), None)
    if bpe is None:
        pytest.skip("the reference's BPE merge table is not available on this machine")
    from efficientsam3_amd.tokenizer import ClipBpeTokenizer
    man, g = text_gold
    tok = ClipBpeTokenizer(bpe)
    for ctx in (16, 77):
        ids = tok(man["prompts"], context_length=ctx)
        assert ids.dtype == np.int64 and np.array_equal(ids, g[f"ids_ctx{ctx}"]), ctx
    one = tok("a dog", context_length=16)
    assert one.shape == (1, 16) and one[0, 0] == 49406 and one[0, 3] == 49407 and one[0, 4:].sum() == 0
    long = tok(" ".join(["word"] * 100), context_length=16)
    assert long[0, -1] == 49407 and (long != 0).all()  # truncated, last token forced to EOT


def test_tokenizer_requires_the_merge_table():
    from efficientsam3_amd.tokenizer import ClipBpeTokenizer
    with pytest.raises(FileNotFoundError):
        ClipBpeTokenizer("/nonexistent/bpe.txt.gz")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_text_encoder_engine_vs_golden(text_gold, text_sd, mode):
    """Tolerances: f32 1e-3 (BASELINE north_star); bf16 3e-2 of the unit-variance output (the reference keeps
    LayerNorm and softmax in fp32 under autocast, mobile_clip.py:267-269,403 -- so does the engine)."""
    from efficientsam3_amd import build_efficientsam3_image_model
    man, g = text_gold
    sd = schema.synthetic_state_dict("efficientvit", "b0", seed=0, enable_inst_interactivity=False)
    sd.update(text_sd)
    model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit",
                                            model_name="b0", dtype=mode, state_dict=sd, text_encoder_type="MobileCLIP-S0",
                                            text_encoder_context_length=man["context_length"])
    lb = model.backbone.language_backbone
    with pytest.raises(RuntimeError, match="grounding-detector"):   # this state dict has no detector weights
        model.forward_grounding({"language_features": None})
    ids = torch.from_numpy(g["ids_ctx16"])
    mask, mem, emb = lb.encode_tokens(ids)
    assert tuple(mem.shape) == g["memory"].shape and tuple(emb.shape) == g["embeds"].shape
    assert np.array_equal(mask.cpu().numpy(), g["mask"])
    assert float(np.abs(emb.cpu().numpy() - g["embeds"]).max()) <= 1e-6
    err = float(np.abs(mem.cpu().numpy() - g["memory"]).max())
    print(f"[{mode}] language_features max-abs-err {err:.3e} (range {g['memory'].min():.2f}..{g['memory'].max():.2f})")
    assert err <= (1e-3 if mode == "f32" else 0.12)
    # batch of one / a sub-batch gives the same rows (no cross-prompt coupling)
    _, mem1, _ = lb.encode_tokens(ids[3:4])
    assert float((mem1[:, 0] - mem[:, 3]).abs().max()) <= (1e-5 if mode == "f32" else 2e-2)
    # forward_text dictionary from token ids is exercised through encode_tokens; the string path
    # needs the merge table:
    bpe = next((p for p in BPE_CANDIDATES if p and os.path.exists(p)), None)
    if bpe is not None:
        lb._bpe_path = bpe
        out = model.backbone.forward_text(man["prompts"][:4])
        assert tuple(out["language_features"].shape) == (16, 4, 256)


# ---- the 12-layer "base" students (MobileCLIP-S1 / -B / MobileCLIP2-*; model_builder.py:525-546) ----
VARIANTS = ["MobileCLIP-S1", "MobileCLIP-B", "MobileCLIP2-L"]


@pytest.fixture(scope="module")
def variants_gold(golden_dir):
    d = os.path.join(golden_dir, "text_variants")
    with open(os.path.join(d, "manifest.json")) as f:
        man = json.load(f)
    return man, np.load(os.path.join(d, "text_variants.npz"))


def test_text_variants_oracle_reproduces_golden(variants_gold):
    from oracle import ref_model
    man, g = variants_gold
    assert man["kinds"] == VARIANTS
    ids = torch.from_numpy(g["ids"])
    for kind in VARIANTS:
        assert man["cases"][kind]["oracle_vs_reference_maxabs"]["memory"] <= 1e-5
        dim, n_layers, heads, variant, causal = schema.TEXT_ENCODER_CFG[kind]
        sd = schema.synthetic_text_state_dict(kind, man["context_length"], seed=0)
        with torch.inference_mode():
            _, mem, emb = ref_model.text_encoder_student(sd, ids, n_layers, heads, variant, causal)
        assert tuple(emb.shape) == (16, ids.shape[0], dim)
        assert float(np.abs(mem.numpy() - g[kind + "_memory"]).max()) <= 1e-5
        assert float(np.abs(emb.numpy()[:, :, ::8] - g[kind + "_embeds_sample"]).max()) <= 1e-6
    # same seeded weights, only the causal mask differs: the fixtures must tell the two apart
    assert float(np.abs(g["MobileCLIP-S1_memory"] - g["MobileCLIP-B_memory"]).max()) > 0.05
    # the aliases share their configuration
    assert schema.TEXT_ENCODER_CFG["MobileCLIP2-S0"] == schema.TEXT_ENCODER_CFG["MobileCLIP-S1"]
    assert schema.TEXT_ENCODER_CFG["MobileCLIP2-S4"] == schema.TEXT_ENCODER_CFG["MobileCLIP2-L"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", VARIANTS)
def test_text_variants_engine_vs_golden(variants_gold, kind):
    """esam3_encode_text on the "base" students (12 encoder layers, no RepMixer blocks; width 512 or 768; causal
    self-attention for MobileCLIP-B) vs the reference's outputs.  Tolerances as for MobileCLIP-S0."""
    from efficientsam3_amd import build_efficientsam3_image_model
    man, g = variants_gold
    ids = torch.from_numpy(g["ids"])
    dim = schema.TEXT_ENCODER_CFG[kind][0]
    for mode, tol in (("f32", 1e-3), ("bf16", 0.15)):
        sd = schema.synthetic_state_dict("efficientvit", "b0", seed=0, enable_inst_interactivity=False)
        sd.update(schema.synthetic_text_state_dict(kind, man["context_length"], seed=0))
        model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit",
                                                model_name="b0", dtype=mode, state_dict=sd, text_encoder_type=kind,
                                                text_encoder_context_length=man["context_length"])
        mask, mem, emb = model.backbone.language_backbone.encode_tokens(ids)
        assert tuple(mem.shape) == g[kind + "_memory"].shape and tuple(emb.shape) == (16, ids.shape[0], dim)
        assert np.array_equal(mask.cpu().numpy(), g["ids"] == 0)
        assert float(np.abs(emb.cpu().numpy()[:, :, ::8] - g[kind + "_embeds_sample"]).max()) <= 1e-6
        err = float(np.abs(mem.cpu().numpy() - g[kind + "_memory"]).max())
        print(f"[{kind} {mode}] language_features max-abs-err {err:.3e}")
        assert err <= tol, (kind, mode, err)
        del model


def test_tokenizer_equals_reference_on_random_text():
    """Property check of row T0 where the reference is present (the build container): for random unicode / ASCII /
    punctuation-heavy strings the product tokenizer emits exactly the reference tokenizer's ids
    (sam3/sam3/model/tokenizer_ve.py:128-253).  Skipped on machines without /root/reference."""
    import sys
    ref_root = "/root/reference/sam3"
    bpe = "/root/reference/sam3/assets/bpe_simple_vocab_16e6.txt.gz"
    if not (os.path.isdir(ref_root) and os.path.exists(bpe)):
        pytest.skip("the reference is not present on this machine")
    from hypothesis import given, settings, strategies as st
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    added = [os.path.join(repo, "oracle", "shims"), ref_root]
    sys.path[:0] = added
    try:
        for m in [k for k in sys.modules if k == "sam3" or k.startswith("sam3.")]:
            del sys.modules[m]
        from sam3.model.tokenizer_ve import SimpleTokenizer
        ref = SimpleTokenizer(bpe_path=bpe)
    finally:
        for p_ in added:
            sys.path.remove(p_)
        for m in [k for k in sys.modules if k == "sam3" or k.startswith("sam3.")]:
            del sys.modules[m]
    from efficientsam3_amd.tokenizer import ClipBpeTokenizer
    mine = ClipBpeTokenizer(bpe)
    alphabet = st.one_of(st.characters(min_codepoint=32, max_codepoint=126), st.characters(min_codepoint=0xA0, max_codepoint=0x2FFF),
                         st.sampled_from(list(" \t\n'\".,!?-_&<>;#@$%0123456789")))

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.text(alphabet=alphabet, min_size=0, max_size=60), min_size=1, max_size=4), st.sampled_from([16, 32, 77]))
    def check(texts, ctx):
        want = ref(texts, context_length=ctx).numpy()
        got = mine(texts, context_length=ctx)
        assert np.array_equal(got, want), (texts, ctx)

    check()
