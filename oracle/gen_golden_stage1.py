"""ORACLE tooling (test infrastructure only): pin oracle/ref_stage1.py against the reference's own loss functions.
The training script cannot be imported (its module-level imports need the training stack), so the three function
definitions are compiled from its source text where it lies under /root/reference and executed here; nothing is
copied into this repository.  Writes tests/golden/stage1/.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_stage1.py
"""
import ast
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientsam3_amd import synth  # noqa: E402
from oracle import ref_stage1  # noqa: E402

SRC = "/root/reference/stage1/train_image_encoder_stage1.py"
GOLD = os.path.join(ROOT, "tests", "golden", "stage1")


def reference_functions():
    tree = ast.parse(open(SRC).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("build_valid_mask", "masked_mse", "masked_cosine_loss")]
    assert len(keep) == 3
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=keep, type_ignores=[]), SRC, "exec"), ns)
    return ns


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = reference_functions()
    out = {}
    for name, (b, c, hw, img, sizes) in synth.stage1_cases().items():
        preds, teacher = synth.stage1_embeddings(name)
        cfg = types.SimpleNamespace(DATA=types.SimpleNamespace(IMG_SIZE=img))
        before_pad = [(3, h, w) for h, w in sizes]
        m_ref = ref["build_valid_mask"](cfg, before_pad, (b, c, hw, hw), "cpu")
        m_ora = ref_stage1.build_valid_mask(img, sizes, (hw, hw))
        assert torch.equal(m_ref, m_ora)
        p, t = torch.from_numpy(preds), torch.from_numpy(teacher)
        mse_r, cos_r = float(ref["masked_mse"](p, t, m_ref)), float(ref["masked_cosine_loss"](p, t, m_ref))
        mse_o, cos_o = float(ref_stage1.masked_mse(p, t, m_ora)), float(ref_stage1.masked_cosine_loss(p, t, m_ora))
        print(name, "ref", mse_r, cos_r, "oracle-ref", mse_o - mse_r, cos_o - cos_r, "valid", int(m_ref.sum()))
        assert mse_o == mse_r and cos_o == cos_r
        out[name] = {"mse": mse_r, "cosine": cos_r, "valid_pixels": [int(v) for v in m_ref.sum(dim=(1, 2, 3))]}
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump({"cases": out, "source": "stage1/train_image_encoder_stage1.py:271-307"}, f, indent=1, sort_keys=True)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
