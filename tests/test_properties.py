"""Property tests (hypothesis) of the host-side pieces that have exact, size-independent invariants."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from efficientsam3_amd import checkpoint as ck
from efficientsam3_amd import dist as esdist
from efficientsam3_amd import rle, stage1


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=2 ** 32 - 1), min_size=1, max_size=64))
def test_rle_string_codec_round_trips_and_matches_oracle(counts):
    """esam3_rle_to_string / esam3_rle_from_string (library host code) are inverse of each other and agree with the
    oracle's restatement of cocoapi for ANY counts, including zeros, 32-bit values and negative deltas."""
    from oracle import ref_rle
    c = np.asarray(counts, dtype=np.uint32)
    s = rle.counts_to_string(c)
    assert s == ref_rle.counts_to_string(counts)
    assert np.array_equal(rle.string_to_counts(s), c)
    assert all(48 <= ord(ch) < 48 + 64 for ch in s)


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 12), st.integers(1, 12), st.integers(0, 2 ** 31 - 1))
def test_rle_oracle_encode_decode_round_trip(h, w, seed):
    from oracle import ref_rle
    m = (np.random.default_rng(seed).random((h, w)) < 0.5).astype(np.uint8)
    counts = ref_rle.rle_counts(m)
    assert sum(counts) == h * w and all(c > 0 for c in counts[1:])   # only the first run may be empty
    assert np.array_equal(ref_rle.decode(ref_rle.string_to_counts(ref_rle.counts_to_string(counts)), h, w), m)


@settings(max_examples=50, deadline=None)
@given(st.integers(8, 300), st.integers(1, 40), st.data())
def test_valid_mask_closed_form_equals_interpolate(img, hw, data):
    """stage1.valid_mask (separable closed form) == bilinear interpolate + threshold of the reference, for any image
    size, grid size (down- or up-sampling) and un-padded extent."""
    from oracle import ref_stage1
    sizes = [(data.draw(st.integers(1, img)), data.draw(st.integers(1, img))) for _ in range(3)]
    want = ref_stage1.build_valid_mask(img, sizes, (hw, hw)).numpy().reshape(3, -1).astype(np.uint8)
    assert np.array_equal(stage1.valid_mask(img, sizes, (hw, hw)), want)


@settings(max_examples=100, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 16))
def test_shard_bounds_partition(n, world):
    spans = [esdist.shard_bounds(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a <= b for a, b in spans) and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes


@settings(max_examples=50, deadline=None)
@given(st.sampled_from(["", "module.", "student_trunk.", "module.student_trunk.", "detector.backbone.vision_backbone.trunk.model.",
                        "backbone.vision_backbone.trunk."]), st.text(alphabet="abcdefgh.", min_size=1, max_size=12))
def test_student_key_normalisation_is_idempotent(prefix, tail):
    k = ck.normalize_image_student_key(prefix + "x" + tail)
    assert ck.normalize_image_student_key(k) == k
    assert ck.normalize_text_student_key(ck.normalize_text_student_key("module." + tail)) == ck.normalize_text_student_key("module." + tail)


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 3), st.integers(0, 4), st.integers(0, 4), st.integers(0, 2 ** 31 - 1))
def test_geometry_prompt_stays_right_padded(b, n_box_appends, n_point_appends, seed):
    """Prompt.append_boxes / append_points with random per-image padding keep every sequence right-padded and keep
    the valid entries of each image in insertion order (geometry_encoders.py:22-79)."""
    from efficientsam3_amd.geometry_prompt import Prompt
    rng = np.random.default_rng(seed)
    p = Prompt(box_embeddings=torch.zeros(0, b, 4), box_mask=torch.zeros(b, 0, dtype=torch.bool))
    want_boxes = [[] for _ in range(b)]
    for _ in range(n_box_appends):
        boxes = torch.from_numpy(rng.random((1, b, 4)).astype(np.float32))
        mask = torch.from_numpy(rng.random((b, 1)) < 0.4)
        p.append_boxes(boxes, torch.ones((1, b), dtype=torch.bool), mask=mask)
        for i in range(b):
            if not bool(mask[i, 0]):
                want_boxes[i].append(boxes[0, i])
    for _ in range(n_point_appends):
        p.append_points(torch.from_numpy(rng.random((1, b, 2)).astype(np.float32)), torch.ones((1, b), dtype=torch.bool))
    bf = p.batch_first()
    for i in range(b):
        valid = (bf["box_mask"][i] == 0)
        n = int(valid.sum())
        assert n == len(want_boxes[i]) and bool(valid[:n].all()) and not bool(valid[n:].any())
        for j, wb in enumerate(want_boxes[i]):
            assert torch.equal(bf["boxes"][i, j], wb)
    assert bf["points"].shape == (b, n_point_appends, 2) and int(bf["point_mask"].sum()) == 0
