"""ORACLE tooling (test infrastructure only): the REFERENCE's own bf16 behaviour on the text-grounding path.

Same idea as oracle/gen_golden_bf16ref.py, for `Sam3Processor.set_text_prompt` (config 4): the real reference is run in
fp32 and under `torch.autocast("cpu", dtype=torch.bfloat16)` on the same seeded weights, image and prompts, and the
distance between the two runs is recorded per output of `forward_grounding`.  Two models:

  * pcs_ev_m      EV-M (EfficientViT-B1) + MobileCLIP-S0-16 + detector  -- the model of tests/golden/pcs_ev_m
  * pcs_vit_h     BASELINE config 4: ViT-H + MobileCLIP-S0-16 + detector (`build_sam3_image_model(text_encoder_type=...)`)

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference/sam3:. python oracle/gen_golden_pcs_bf16ref.py [--model ev_m|vit_h]

`--draws N` adds N further seeded images (seeds 2 ..): the yardstick of a test is then the worst distance over the fixture image and
the draws, not one sample of it (a single draw is a noisy estimate of the reference's own bf16 distance).

`--geo N` measures the same distance for the two GEOMETRIC-prompt cases of oracle/gen_golden_pcs.py on image seeds 1..N and writes only
bf16ref_geo.json (round 5: the geometric cases used to borrow 2 x the text cases' figure).

Output: tests/golden/pcs_<model>/bf16ref_manifest.json (+ bf16ref_draws.json, bf16ref_geo.json)
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from efficientsam3_amd import schema, synth  # noqa: E402

CTX = 16
PROMPTS = ["dog", "traffic light"]
THRESH = 0.05
KEYS = ("pred_logits", "pred_boxes", "presence_logit_dec", "pred_masks")


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ev_m", choices=["ev_m", "vit_h"])
    ap.add_argument("--draws", type=int, default=0)
    ap.add_argument("--draws-only", action="store_true", help="leave bf16ref_manifest.json as it is")
    ap.add_argument("--geo", type=int, default=0, help="N images (seeds 1..N) of the geometric-prompt cases -> bf16ref_geo.json, nothing else")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    gold = os.path.join(ROOT, "tests", "golden", f"pcs_{args.model}")
    os.makedirs(gold, exist_ok=True)
    from sam3 import build_efficientsam3_image_model, build_sam3_image_model  # the REAL reference
    from sam3.model.sam3_image_processor import Sam3Processor
    if args.model == "ev_m":
        model = build_efficientsam3_image_model(
            device="cpu", checkpoint_path=None, load_from_HF=False, enable_inst_interactivity=False,
            backbone_type="efficientvit", model_name="b1", text_encoder_type="MobileCLIP-S0", text_encoder_context_length=CTX)
        sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0, enable_inst_interactivity=False)
    else:
        model = build_sam3_image_model(device="cpu", checkpoint_path=None, load_from_HF=False, enable_inst_interactivity=False,
                                       text_encoder_type="MobileCLIP-S0", text_encoder_context_length=CTX)
        sd = schema.synthetic_state_dict("sam3", "vit_h", seed=0, enable_inst_interactivity=False)
    sd.update(schema.synthetic_text_state_dict("MobileCLIP-S0", CTX, seed=0))
    sd.update(schema.synthetic_pcs_state_dict(seed=0))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected[:5]
    model.eval()
    if args.model == "vit_h":  # the builder creates the student at ctx 77 and truncates after the checkpoint load
        model.backbone.language_backbone.set_context_length(CTX)
    proc = Sam3Processor(model, device="cpu", confidence_threshold=THRESH)
    image = lambda seed: torch.from_numpy(np.ascontiguousarray(np.moveaxis(synth.smooth_image_u8(seed=seed), -1, 0)))
    chw = image(1)
    captured = {}
    orig = model.forward_grounding

    def wrapped(*a, **k):
        out = orig(*a, **k)
        captured["out"] = out
        return out

    model.forward_grounding = wrapped

    def run(amp: bool, chw=chw):
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if amp else torch.autocast("cpu", enabled=False)
        res = []
        with torch.inference_mode(), ctx:
            state = proc.set_image(chw)
            for text in PROMPTS:
                proc.reset_all_prompts(state)
                state = proc.set_text_prompt(text, state)
                out = {k: captured["out"][k].float().clone() for k in KEYS}
                out["n_kept"] = int(state["scores"].numel())
                res.append(out)
        return res

    t0 = time.time()
    if args.geo:
        # the two geometric-prompt cases of oracle/gen_golden_pcs.py (add_geometric_prompt / add_point_prompt,
        # sam3_image_processor.py:130-190) in fp32 and under bf16 autocast: image seed 1 (the fixture image) first, then --geo - 1 draws
        def run_geo(amp: bool, chw):
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if amp else torch.autocast("cpu", enabled=False)
            res = {}
            with torch.inference_mode(), ctx:
                state = proc.set_image(chw)
                proc.reset_all_prompts(state)
                state = proc.set_text_prompt("dog", state)
                state = proc.add_geometric_prompt([0.45, 0.5, 0.3, 0.4], True, state)
                res["geo_text_box"] = {k: captured["out"][k].float().clone() for k in KEYS}
                proc.reset_all_prompts(state)
                state = proc.add_point_prompt([300.0, 420.0], 1, state)
                state = proc.add_geometric_prompt([0.6, 0.4, 0.2, 0.25], False, state)
                state = proc.add_point_prompt([700.5, 200.0], 0, state)
                state = proc.add_geometric_prompt([0.25, 0.7, 0.45, 0.5], True, state)
                res["geo_visual_mixed"] = {k: captured["out"][k].float().clone() for k in KEYS}
            return res
        geo = {"model": f"pcs_{args.model}", "seeds": list(range(1, 1 + args.geo)), "cases": {"geo_text_box": [], "geo_visual_mixed": []}}
        for seed in geo["seeds"]:
            a32, a16 = run_geo(False, image(seed)), run_geo(True, image(seed))
            for name in geo["cases"]:
                e = {k: float((a32[name][k] - a16[name][k]).abs().max()) for k in KEYS}
                e["seed"] = seed
                geo["cases"][name].append(e)
                print("geo", seed, name, e, f"[{time.time() - t0:.0f}s]", flush=True)
            with open(os.path.join(gold, "bf16ref_geo.json"), "w") as f:
                json.dump(geo, f, indent=1, sort_keys=True)
        return
    if args.draws:
        draws = {"model": f"pcs_{args.model}", "seeds": list(range(2, 2 + args.draws)), "cases": {t: [] for t in PROMPTS}}
        for seed in draws["seeds"]:
            a32, a16 = run(False, image(seed)), run(True, image(seed))
            for text, a, b in zip(PROMPTS, a32, a16):
                e = {k: float((a[k] - b[k]).abs().max()) for k in KEYS}
                e["seed"] = seed
                draws["cases"][text].append(e)
                print("draw", seed, text, e, f"[{time.time() - t0:.0f}s]", flush=True)
        with open(os.path.join(gold, "bf16ref_draws.json"), "w") as f:
            json.dump(draws, f, indent=1, sort_keys=True)
        if args.draws_only:
            return
    r32 = run(False)
    r16 = run(True)
    manifest = {"model": f"pcs_{args.model}", "autocast": "torch.autocast('cpu', dtype=torch.bfloat16)", "torch": torch.__version__,
                "prompts": PROMPTS, "confidence_threshold": THRESH, "cases": {}}
    for text, a, b in zip(PROMPTS, r32, r16):
        e = {k: float((a[k] - b[k]).abs().max()) for k in KEYS}
        e["ranges"] = {k: [float(a[k].min()), float(a[k].max())] for k in KEYS}
        e["n_kept_fp32"], e["n_kept_bf16"] = a["n_kept"], b["n_kept"]
        manifest["cases"][text] = e
        print(text, e)
    with open(os.path.join(gold, "bf16ref_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(f"wrote {gold}/bf16ref_manifest.json in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
