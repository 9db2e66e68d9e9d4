"""ORACLE tooling (test infrastructure only): pin the text-encoder restatement
(``oracle/ref_model.text_encoder_student``) and the product tokenizer
(``efficientsam3_amd/tokenizer.py``) against the REAL reference, and write the fixtures under
``tests/golden/text_s0/``.  Runs only where ``/root/reference`` exists:

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_text.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from efficientsam3_amd import schema  # noqa: E402
from efficientsam3_amd.tokenizer import ClipBpeTokenizer  # noqa: E402
from oracle import ref_model  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "text_s0")
BPE = "/root/reference/sam3/assets/bpe_simple_vocab_16e6.txt.gz"
CTX = 16
# noun phrases (fallback list) + inputs that exercise cleaning, BPE merges,
# contractions, unicode bytes, truncation (> 16 tokens) and the empty string
PROMPTS = ["dog", "person", "car", "tree", "chair", "bottle", "window", "shoe",
           "a dog", "The quick brown fox's jumped over 13 lazy dogs!!", "  multiple   spaces\tand\nnewlines ",
           "café au lait — naïve coöperation", "traffic light", "hello &amp; goodbye &lt;tag&gt;",
           "it's we're they've I'm you'll he'd can't",
           "supercalifragilisticexpialidocious antidisestablishmentarianism pneumonoultramicroscopicsilicovolcanoconiosis",
           "$100.50 (approx.) #hashtag @user", "", "日本語のテキスト", "emoji 😀 test"]


def main():
    torch.manual_seed(0)
    os.makedirs(GOLD, exist_ok=True)
    from sam3 import build_efficientsam3_image_model  # the REAL reference
    model = build_efficientsam3_image_model(
        device="cpu", checkpoint_path=None, load_from_HF=False, enable_inst_interactivity=False,
        backbone_type="efficientvit", model_name="b0", text_encoder_type="MobileCLIP-S0",
        text_encoder_context_length=CTX)
    sd = schema.synthetic_text_state_dict("MobileCLIP-S0", CTX, seed=0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected[:5]
    assert not [k for k in missing if "language_backbone" in k], [k for k in missing if "language_backbone" in k][:5]
    model.eval()
    lb = model.backbone.language_backbone

    digest = hashlib.sha256()
    for k, v in sd.items():
        digest.update(k.encode())
        digest.update(np.ascontiguousarray(v.numpy()).tobytes())

    # ---- tokenizer --------------------------------------------------------------------
    mine = ClipBpeTokenizer(BPE)
    tok = {}
    for ctx in (CTX, 77):
        ref_ids = lb.tokenizer(PROMPTS, context_length=ctx).numpy()
        assert np.array_equal(ref_ids, mine(PROMPTS, context_length=ctx)), f"tokenizer mismatch at ctx {ctx}"
        tok[f"ids_ctx{ctx}"] = ref_ids
    # ---- encoder ------------------------------------------------------------------------
    with torch.inference_mode():
        mask_r, mem_r, emb_r = lb(PROMPTS, None, torch.device("cpu"))
        mask_o, mem_o, emb_o = ref_model.text_encoder_student(sd, torch.from_numpy(tok[f"ids_ctx{CTX}"]))
    errs = {"mask": int((mask_r != mask_o).sum()), "memory": float((mem_r - mem_o).abs().max()),
            "embeds": float((emb_r - emb_o).abs().max())}
    print("oracle vs reference:", errs, "| memory range", float(mem_r.min()), float(mem_r.max()),
          "std", float(mem_r.std()))
    np.savez_compressed(os.path.join(GOLD, "text_cases.npz"), mask=mask_r.numpy(), memory=mem_r.numpy(),
                        embeds=emb_r.numpy(), **tok)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump({"weights_sha256": digest.hexdigest(), "weights_seed": 0, "model": "MobileCLIP-S0", "context_length": CTX,
                   "prompts": PROMPTS, "oracle_vs_reference_maxabs": errs}, f, indent=1, sort_keys=True, ensure_ascii=True)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
