#!/usr/bin/env python3
"""Which Python lines of a stage-1 training step issue device copies (``__amd_rocclr_copyBuffer``: contiguous ``copy_`` / ``clone`` /
host uploads) and ATen kernels: one step under ``torch.profiler`` with stacks, grouped by the innermost frame inside this package.

    python tools/step_copy_census.py [--model b1] [--batch 32]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from efficientsam3_amd import schema                      # noqa: E402
from efficientsam3_amd.stage1_train import Stage1Trainer  # noqa: E402

PREFIX = "backbone.vision_backbone.trunk.model."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="b1")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    if a.model.startswith("repvit_"):
        family, name = "repvit", a.model[len("repvit_"):].replace("_", ".")
    elif a.model.startswith("tiny_vit_"):
        family, name = "tinyvit", a.model[len("tiny_vit_"):]
    else:
        family, name = "efficientvit", a.model
    sd = schema.synthetic_state_dict(family, name, seed=0)
    sd = {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}
    tr = Stage1Trainer(sd, a.model, embed_size=72, dtype="bf16", lr=1e-4, weight_decay=0.05, clip_grad=5.0, cosine_weight=0.5)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn((a.batch, 3, 1008, 1008), generator=g).cuda()
    teacher = (torch.randn((a.batch, 72, 72, 1024), generator=g) * 0.5).to("cuda", torch.bfloat16)
    sizes = [(1008, 1008)] * a.batch
    for _ in range(2):
        tr.step(imgs, teacher, sizes)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.step(imgs, teacher, sizes)
        torch.cuda.synchronize()
    by, kern = collections.Counter(), collections.defaultdict(collections.Counter)

    def device_kernels(ev):
        out = [k.name for k in ev.kernels]
        for c in ev.cpu_children:
            out += device_kernels(c)
        return out

    for ev in prof.events():
        if not ev.name.startswith("aten::") or (ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::")):
            continue                                        # outermost ATen calls only
        ks = device_kernels(ev)
        if not ks:
            continue
        by[ev.name] += 1
        for k in ks:
            kern[ev.name][k[:70]] += 1
    print("outermost ATen calls with device work in one step -> the kernels / copies they launch:")
    for name, n in by.most_common(40):
        print(f"{n:5d}  {name:24s} " + ", ".join(f"{c} x {k}" for k, c in kern[name].most_common(4)))
    names = collections.Counter(ev.name[:70] for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    print("device events of the step:", sum(names.values()), "; copies:", {k: v for k, v in names.items() if "copy" in k.lower() or "Memcpy" in k})


if __name__ == "__main__":
    main()
