"""ORACLE tooling (test infrastructure only): pin the text-encoder restatement for the 12-layer "base" students
(MobileCLIP-S1 = the config shared with MobileCLIP2-S0/S2, MobileCLIP-B with causal masking, MobileCLIP2-L = the
768-wide config shared with MobileCLIP2-S3/S4; model_builder.py:525-546) against the REAL reference and write
``tests/golden/text_variants/``.  Runs only where ``/root/reference`` exists:

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_text_variants.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientsam3_amd import schema  # noqa: E402
from oracle import ref_model  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "text_variants")
CTX = 16
KINDS = ["MobileCLIP-S1", "MobileCLIP-B", "MobileCLIP2-L"]


def main():
    os.makedirs(GOLD, exist_ok=True)
    from sam3 import build_efficientsam3_image_model  # the REAL reference
    ids = np.load(os.path.join(ROOT, "tests", "golden", "text_s0", "text_cases.npz"))["ids_ctx16"][:8]
    manifest = {"context_length": CTX, "kinds": KINDS, "cases": {}}
    arrays = {"ids": ids}
    for kind in KINDS:
        model = build_efficientsam3_image_model(
            device="cpu", checkpoint_path=None, load_from_HF=False, enable_inst_interactivity=False,
            backbone_type="efficientvit", model_name="b0", text_encoder_type=kind, text_encoder_context_length=CTX)
        sd = schema.synthetic_text_state_dict(kind, CTX, seed=0)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected[:5]
        assert not [k for k in missing if "language_backbone" in k], [k for k in missing if "language_backbone" in k][:5]
        model.eval()
        lb = model.backbone.language_backbone
        dim, n_layers, heads, variant, causal = schema.TEXT_ENCODER_CFG[kind]
        assert lb.encoder.causal_masking == causal and len(lb.encoder.transformer) == n_layers
        with torch.inference_mode():
            tok = torch.from_numpy(ids)
            emb_r = lb.encoder.forward_embedding(tok)
            mem_r = lb.projector(lb.encoder(emb_r, return_all_tokens=True, input_is_embeddings=True)).transpose(0, 1)
            mask_o, mem_o, emb_o = ref_model.text_encoder_student(sd, tok, n_layers, heads, variant, causal)
        errs = {"memory": float((mem_r - mem_o).abs().max()), "embeds": float((emb_r.transpose(0, 1) - emb_o).abs().max())}
        print(kind, "oracle vs reference:", errs, "| memory std", float(mem_r.std()))
        manifest["cases"][kind] = {"oracle_vs_reference_maxabs": errs, "dim": dim, "layers": n_layers, "causal": causal}
        arrays[kind + "_memory"] = mem_r.numpy()
        arrays[kind + "_embeds_sample"] = emb_r.transpose(0, 1).numpy()[:, :, ::8]
        del model
    np.savez_compressed(os.path.join(GOLD, "text_variants.npz"), **arrays)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
