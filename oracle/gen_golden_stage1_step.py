"""ORACLE tooling (test infrastructure only): TWO stage-1 training steps of the REAL reference stack as a fixture.

What runs, all of it the reference's own code on the CPU in fp32 (AMP off):

  * the student `ImageStudentEncoder` built by `stage1/model.py: build_image_student_model` (MODEL.BACKBONE efficientvit_b1,
    DATA.IMG_SIZE 1008, DISTILL.EMBED_DIM 1024, DISTILL.EMBED_SIZE 72) in `model.train()` mode;
  * the optimizer built by `stage1/optimizer.py: build_optimizer` (AdamW, the two weight-decay groups of `set_weight_decay`);
  * `masked_mse` + COSINE x `masked_cosine_loss` with `build_valid_mask` (compiled from the source text of
    stage1/train_image_encoder_stage1.py:271-307: that module's imports need the training stack);
  * the loss-scaler call sequence of `NativeScalerWithGradNormCount.__call__` (stage1/utils.py:347-362) with AMP disabled:
    `loss.backward()`, `clip_grad_norm_(parameters, clip_grad)`, `optimizer.step()`, then `optimizer.zero_grad()`
    (train_image_encoder_stage1.py:210-219).

Weights are the seeded synthetic state dict of `efficientsam3_amd/schema.py` (the reference's key names; loaded strict), the two
images come from `efficientsam3_amd/synth.py` and the teacher embeddings from a seeded CPU generator -- the GPU test regenerates all
three, so the fixture holds only results: the losses, the gradient norms, strided samples (<= 256 values per tensor) of every
parameter gradient of step 1 and of every parameter after steps 1 and 2, and the BatchNorm running statistics after each step.
A second run under `torch.autocast("cpu", bfloat16)` gives the bf16 yardstick (loss and gradient-norm distance of the reference
from itself).

    PYTHONDONTWRITEBYTECODE=1 CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" \
    PYTHONPATH=oracle/shims:/root/reference:. python oracle/gen_golden_stage1_step.py     (the OUTER sam3 package: stage1/model.py imports sam3.sam3.*)

Output: tests/golden/stage1/step.npz + step_manifest.json (`--model b2`: step_b2.npz + step_b2_manifest.json, round 5;
`--model repvit_m0_9` / `repvit_m1_1`: step_repvit_m0_9.* / step_repvit_m1_1.*, the RepViT students of stage1/model.py:386-395;
`--model tiny_vit_5m` / `tiny_vit_11m`: step_tiny_vit_5m.* / step_tiny_vit_11m.*, the TinyViT students of stage1/model.py:397-406)
"""
from __future__ import annotations

import ast
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/stage1")

from efficientsam3_amd import schema, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "stage1")
SRC = "/root/reference/stage1/train_image_encoder_stage1.py"
PREFIX = "backbone.vision_backbone.trunk.model."       # the student encoder inside the full SAM3 state dict
HYPER = dict(lr=1e-4, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, clip_grad=5.0, cosine=0.5, accumulation_steps=1)
IMG, EMBED_DIM, EMBED_SIZE = 1008, 1024, 72
SIZES = [(1008, 1008), (756, 1008)]        # img_size_before_pad of the two samples (the second one is padded at the bottom)
IMAGE_SEEDS, TEACHER_SEED = (11, 12), 7
NSAMP = 256


MODEL = "b1"          # --model b0 | b1 | b2 (the default b1 writes step.npz / step_manifest.json, the others step_<model>.*)
                      # | repvit_m0_9 | repvit_m1_1 | repvit_m2_3 (round 5: the RepViT students, stage1/model.py:386-395)
                      # | tiny_vit_5m | tiny_vit_11m | tiny_vit_21m (the TinyViT students, stage1/model.py:397-406; 11m / 21m train with
                      #   stochastic depth: the per-sample keep masks every DropPath drew are part of the fixture)


def _family():
    """(schema backbone type, schema model name, MODEL.BACKBONE of the reference's config)"""
    if MODEL.startswith("repvit_"):
        return "repvit", MODEL[len("repvit_"):].replace("_", "."), MODEL
    if MODEL.startswith("tiny_vit_"):
        return "tinyvit", MODEL[len("tiny_vit_"):], MODEL
    return "efficientvit", MODEL, f"efficientvit_{MODEL}"


def student_state_dict():
    family, name, _ = _family()
    sd = schema.synthetic_state_dict(family, name, seed=0)
    return {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}


def inputs():
    imgs = torch.stack([torch.from_numpy(synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s))) for s in IMAGE_SEEDS])
    for i, (h, w) in enumerate(SIZES):          # the dataset pads with zeros below / right of the resized image
        imgs[i, :, h:, :] = 0
        imgs[i, :, :, w:] = 0
    g = torch.Generator().manual_seed(TEACHER_SEED)
    teacher = torch.randn((len(SIZES), EMBED_DIM, EMBED_SIZE, EMBED_SIZE), generator=g) * 0.5
    return imgs, teacher


def reference_functions():
    tree = ast.parse(open(SRC).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("build_valid_mask", "masked_mse", "masked_cosine_loss")]
    assert len(keep) == 3
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=keep, type_ignores=[]), SRC, "exec"), ns)
    return ns


def sample(t: torch.Tensor) -> np.ndarray:
    flat = t.detach().float().reshape(-1)
    step = max(1, flat.numel() // NSAMP)
    return flat[::step][:NSAMP].numpy().copy()


def run(amp: bool):
    import model as ref_model            # stage1/model.py
    import optimizer as ref_optimizer    # stage1/optimizer.py
    ref = reference_functions()
    cfg = SimpleNamespace(MODEL=SimpleNamespace(BACKBONE=_family()[2]), DATA=SimpleNamespace(IMG_SIZE=IMG),
                          DISTILL=SimpleNamespace(EMBED_DIM=EMBED_DIM, EMBED_SIZE=EMBED_SIZE, COSINE=HYPER["cosine"]),
                          TRAIN=SimpleNamespace(OPTIMIZER=SimpleNamespace(NAME="adamw", EPS=HYPER["eps"], BETAS=HYPER["betas"], MOMENTUM=0.9),
                                                BASE_LR=HYPER["lr"], WEIGHT_DECAY=HYPER["weight_decay"], CLIP_GRAD=HYPER["clip_grad"],
                                                ACCUMULATION_STEPS=HYPER["accumulation_steps"]))
    torch.manual_seed(0)
    net = ref_model.build_image_student_model(cfg)
    missing, unexpected = net.load_state_dict(student_state_dict(), strict=True)
    net.train()
    drawn = []           # (module name, call index within the forward pass, per-sample factor) of every DropPath call, in call order
    calls = {}

    def recording(name, mod):
        def forward(x):
            if mod.drop_prob == 0.0 or not mod.training:
                return x
            keep = 1 - mod.drop_prob                       # timm.layers.drop_path (scale_by_keep=True), as the shim restates it
            # drawn in fp32 whatever the activation dtype, so that the fp32 and the bf16-autocast run drop the same branches
            mask = torch.empty((x.shape[0],) + (1,) * (x.ndim - 1), dtype=torch.float32).bernoulli_(keep)
            if keep > 0.0:
                mask.div_(keep)
            k = calls.get(name, 0)
            calls[name] = k + 1
            drawn.append((name, k, mask.reshape(-1).clone()))
            return x * mask.to(x.dtype)
        return forward

    for name, mod in net.named_modules():
        if type(mod).__name__ == "DropPath":
            mod.forward = recording(name, mod)
    opt = ref_optimizer.build_optimizer(cfg, net)
    opt.zero_grad()
    imgs, teacher = inputs()
    named = dict(net.named_parameters())
    rec = {"losses": [], "mse": [], "cosine": [], "grad_norms": []}
    arrays = {}
    torch.manual_seed(1234)      # the stream the DropPath masks are drawn from (the same in the fp32 and the bf16 run)
    for step in range(2):
        t0 = time.time()
        calls.clear()
        del drawn[:]
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if amp else torch.autocast("cpu", enabled=False)
        with ctx:
            preds = net(imgs)
        valid = ref["build_valid_mask"](cfg, [(3, h, w) for h, w in SIZES], preds.shape, preds.device)
        mse = ref["masked_mse"](preds, teacher, valid)
        cos = ref["masked_cosine_loss"](preds, teacher, valid)
        loss = (mse + cfg.DISTILL.COSINE * cos) / cfg.TRAIN.ACCUMULATION_STEPS
        # NativeScalerWithGradNormCount.__call__ with the GradScaler disabled (AMP off): backward, clip_grad_norm_, step
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(net.parameters(), cfg.TRAIN.CLIP_GRAD)
        for name, k, mask in drawn:
            arrays[f"droppath{step + 1}/{name}/{k}"] = mask.numpy()
        if step == 0 and not amp:
            for n, p in named.items():
                arrays[f"grad1/{n}"] = sample(p.grad)    # after clipping: what AdamW sees
                arrays[f"gradmax1/{n}"] = np.asarray(float(p.grad.abs().max()), dtype=np.float32)
        opt.step()
        opt.zero_grad()
        rec["losses"].append(float(loss)); rec["mse"].append(float(mse)); rec["cosine"].append(float(cos)); rec["grad_norms"].append(float(norm))
        if not amp:
            for n, p in named.items():
                arrays[f"param{step + 1}/{n}"] = sample(p)
            for k, v in net.state_dict().items():
                if k.endswith(("running_mean", "running_var")):
                    arrays[f"buffer{step + 1}/{k}"] = sample(v)
        print(f"{'bf16-autocast' if amp else 'fp32'} step {step + 1}: loss {float(loss):.6f} (mse {float(mse):.6f}, cos {float(cos):.6f}) "
              f"grad norm {float(norm):.6f}  [{time.time() - t0:.0f}s]", flush=True)
    if not amp:
        rec["names"] = list(named)
        rec["shapes"] = {n: list(p.shape) for n, p in named.items()}
        rec["valid_pixels"] = [int(v) for v in valid.sum(dim=(1, 2, 3))]
    return rec, arrays


def main():
    global MODEL
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=NSAMP, help="strided samples kept per tensor (256; the fixtures of the largest students keep 64)")
    ap.add_argument("--model", default="b1", choices=["b0", "b1", "b2", "repvit_m0_9", "repvit_m1_1", "repvit_m2_3", "tiny_vit_5m", "tiny_vit_11m", "tiny_vit_21m"])
    a_ = ap.parse_args()
    MODEL = a_.model
    globals()["NSAMP"] = a_.samples
    suffix = "" if MODEL == "b1" else f"_{MODEL}"
    os.makedirs(GOLD, exist_ok=True)
    fp32, arrays = run(False)
    bf16, arrays16 = run(True)
    for k in [k for k in arrays if k.startswith("droppath")]:
        assert np.array_equal(arrays[k], arrays16[k]), k      # both runs dropped the same residual branches
    man = {"source": "stage1/train_image_encoder_stage1.py:165-226, stage1/model.py:28-37,188-211, stage1/optimizer.py:6-46, stage1/utils.py:341-368",
           "hyper": HYPER, "img_size": IMG, "embed_dim": EMBED_DIM, "embed_size": EMBED_SIZE, "sizes_before_pad": SIZES,
           "image_seeds": list(IMAGE_SEEDS), "teacher_seed": TEACHER_SEED, "samples_per_tensor": NSAMP, "torch": torch.__version__,
           "fp32": fp32, "bf16_autocast": bf16, "model": MODEL}
    np.savez_compressed(os.path.join(GOLD, f"step{suffix}.npz"), **arrays)
    with open(os.path.join(GOLD, f"step{suffix}_manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print("wrote", GOLD, f"step{suffix}.npz", os.path.getsize(os.path.join(GOLD, f"step{suffix}.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
