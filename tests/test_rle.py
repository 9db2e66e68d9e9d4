"""Mask -> COCO RLE (the evaluation writers' encode step).

CPU: the oracle (oracle/ref_rle.py) reproduces the run lengths the REAL reference's ``rle_encode`` produced
(tests/golden/rle, oracle/gen_golden_rle.py), its string codec round-trips, and the library's host codec
(esam3_rle_to_string / esam3_rle_from_string, no device work) agrees with it character for character.
GPU: the HIP kernels (esam3_rle_encode through the C ABI) are bit-exact against the fixtures and the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from efficientsam3_amd import synth


@pytest.fixture(scope="module")
def rle_gold(golden_dir):
    d = os.path.join(golden_dir, "rle")
    with open(os.path.join(d, "manifest.json")) as f:
        man = json.load(f)
    return man, np.load(os.path.join(d, "rle_cases.npz"))


def test_oracle_counts_match_reference_fixtures(rle_gold):
    from oracle import ref_rle
    man, g = rle_gold
    masks = synth.rle_test_masks()
    assert set(masks) == set(man["cases"])
    for name, mk in masks.items():
        assert man["cases"][name]["oracle_equals_reference_counts"]
        counts, offs = g[name + "_counts"], g[name + "_offsets"]
        for i in range(mk.shape[0]):
            c = ref_rle.rle_counts(mk[i])
            assert c == [int(v) for v in counts[offs[i]:offs[i + 1]]], (name, i)
            assert sum(c) == mk[i].size and len(c) >= 1
            s = ref_rle.counts_to_string(c)
            assert s == man["cases"][name]["strings"][i]
            assert ref_rle.string_to_counts(s) == c
            assert np.array_equal(ref_rle.decode(c, *mk[i].shape), (mk[i] != 0).astype(np.uint8))


def test_library_string_codec_matches_oracle(rle_gold):
    """The host codec of the C-ABI library (no GPU needed) vs the oracle, including counts whose deltas are
    negative and counts that need the full 32 bits."""
    from efficientsam3_amd import rle
    from oracle import ref_rle
    man, g = rle_gold
    for name, case in man["cases"].items():
        counts, offs = g[name + "_counts"], g[name + "_offsets"]
        for i, s_ref in enumerate(case["strings"]):
            c = counts[offs[i]:offs[i + 1]]
            assert rle.counts_to_string(c) == s_ref
            assert np.array_equal(rle.string_to_counts(s_ref), c)
    rng = np.random.default_rng(3)
    for _ in range(50):
        c = rng.integers(0, [3, 40, 2000, 2 ** 31][rng.integers(4)], size=rng.integers(1, 40)).astype(np.uint32)
        c[rng.integers(c.size)] = np.uint32(rng.integers(0, 2 ** 32 - 1, dtype=np.uint64))
        s = ref_rle.counts_to_string([int(v) for v in c])
        assert rle.counts_to_string(c) == s
        assert np.array_equal(rle.string_to_counts(s), c)


@pytest.mark.gpu
def test_rle_kernels_bit_exact(rle_gold):
    from efficientsam3_amd import rle
    from oracle import ref_rle
    man, g = rle_gold
    for name, mk in synth.rle_test_masks().items():
        counts, offs = rle.rle_counts_device(torch.from_numpy(mk).cuda())
        assert np.array_equal(offs, g[name + "_offsets"]), name
        assert np.array_equal(counts, g[name + "_counts"]), name
        out = rle.rle_encode(torch.from_numpy(mk != 0).cuda(), return_areas=True)
        assert [r["counts"] for r in out] == man["cases"][name]["strings"]
        assert [r["area"] for r in out] == [int((m != 0).sum()) for m in mk]
        assert out[0]["size"] == list(mk.shape[1:])
    # a deliberately small capacity: truncated first pass, exact retry
    mk = synth.rle_test_masks()["noise_37x53"]
    counts, offs = rle.rle_counts_device(torch.from_numpy(mk).cuda(), capacity=8)
    assert np.array_equal(counts, g["noise_37x53_counts"])
    # full-size property: many masks at 1008^2 decode back to themselves; total = H*W per mask
    big = (torch.rand((24, 1008, 1008), generator=torch.Generator().manual_seed(0)) < 0.5)
    big[::2] = torch.from_numpy(synth.rle_test_masks()["blobs_1008"][0] != 0)
    counts, offs = rle.rle_counts_device(big.cuda())
    for i in (0, 1, 23):
        c = [int(v) for v in counts[offs[i]:offs[i + 1]]]
        assert sum(c) == 1008 * 1008
        assert np.array_equal(ref_rle.decode(c, 1008, 1008), big[i].numpy().astype(np.uint8))
    assert rle.rle_encode(torch.zeros((0, 4, 4), dtype=torch.bool, device="cuda")) == []
