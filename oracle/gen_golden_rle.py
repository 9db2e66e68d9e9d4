"""ORACLE tooling (test infrastructure only): pin oracle/ref_rle.py's run lengths against the REAL reference's
``rle_encode`` (sam3/sam3/train/masks_ops.py:161-230, run on CPU tensors) and write tests/golden/rle/.
The compressed string form comes from the pycocotools shim (= the oracle's restatement of cocoapi; unpinned).

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_rle.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientsam3_amd import synth  # noqa: E402
from oracle import ref_rle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "rle")


def main():
    os.makedirs(GOLD, exist_ok=True)
    import pycocotools.mask as shim_mask
    from sam3.train.masks_ops import rle_encode  # the REAL reference
    cases = synth.rle_test_masks()
    manifest = {"cases": {}}
    arrays = {}
    for name, masks in cases.items():
        del shim_mask.LAST_UNCOMPRESSED[:]
        out = rle_encode(torch.from_numpy(masks.astype(bool)), return_areas=True)
        assert len(out) == masks.shape[0] == len(shim_mask.LAST_UNCOMPRESSED)
        flat, offs, strings, ok = [], [0], [], True
        for i in range(masks.shape[0]):
            ref_counts = shim_mask.LAST_UNCOMPRESSED[i]["counts"]
            ora_counts = ref_rle.rle_counts(masks[i])
            ok = ok and ref_counts == ora_counts and out[i]["area"] == int(masks[i].astype(bool).sum())
            assert np.array_equal(ref_rle.decode(ref_counts, *masks[i].shape), (masks[i] != 0).astype(np.uint8))
            flat += ref_counts
            offs.append(len(flat))
            strings.append(out[i]["counts"])
        print(name, masks.shape, "runs", len(flat), "oracle == reference:", ok)
        assert ok
        manifest["cases"][name] = {"shape": list(masks.shape), "oracle_equals_reference_counts": ok, "strings": strings}
        arrays[name + "_counts"] = np.asarray(flat, dtype=np.uint32)
        arrays[name + "_offsets"] = np.asarray(offs, dtype=np.int32)
    np.savez_compressed(os.path.join(GOLD, "rle_cases.npz"), **arrays)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
