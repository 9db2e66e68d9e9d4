"""Development aid (round 5): where does the TRAINING-mode EfficientViT trunk on the HIP kernels (fp32) leave the same composition run with
plain torch stand-ins on the host?  Forward activations after every block, then the input gradient after every block's backward (same
random upstream gradient), relative max-abs difference per block.  Written to localise the EfficientViT-B2 step discrepancy
(tests/test_stage1_step.py::test_b2_training_step_matches_the_reference_run).     python tools/trunk_train_layer_diff.py [b2]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientsam3_amd import schema, synth, train_blocks as tb  # noqa: E402
from efficientsam3_amd.stage1_train import EFFICIENTVIT  # noqa: E402

MODEL = sys.argv[1] if len(sys.argv) > 1 else "b2"
PREFIX = "backbone.vision_backbone.trunk.model."
sd = schema.synthetic_state_dict("efficientvit", MODEL, seed=0)
sd = {k[len(PREFIX):]: v.clone().float() for k, v in sd.items() if k.startswith(PREFIX)}
widths, depths, dim = EFFICIENTVIT[MODEL]
imgs = torch.stack([torch.from_numpy(synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s))) for s in (11, 12)])
imgs[1, :, 756:, :] = 0


def run(device):
    tb.DEVICE = device
    params = {k: v.to(device).contiguous() for k, v in sd.items()}
    trunk = tb.EfficientViTTrunkTrain(params, widths, depths, dim, dtype=torch.float32, prefix="backbone.model.")
    outs = []
    x = trunk.stem.forward(imgs.to(device))
    outs.append(("stem", x.float().cpu()))
    for i, (blk, back) in enumerate(trunk.layers):
        x = blk.forward(x)
        outs.append((f"{i}:{type(blk).__name__}:{next(iter(back.values())).rsplit('.', 3)[0]}", x.float().cpu()))
    g = torch.Generator().manual_seed(3)
    d = (torch.randn(x.shape, generator=g) * 1e-2).to(device)
    douts = []
    for i, (blk, back) in reversed(list(enumerate(trunk.layers))):
        d, _ = blk.backward(d)
        douts.append((f"{i}:{type(blk).__name__}", d.float().cpu()))
    return outs, douts


gpu_f, gpu_b = run("cuda")
torch.cuda.synchronize()
import tests.test_train_blocks_host as H  # noqa: E402


class MP:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


H.cpu_kernels.__wrapped__(MP())
tb.stem_forward = lambda img, w, dtype: F.conv2d(img, w.float(), None, stride=2, padding=1).permute(0, 2, 3, 1).contiguous()
cpu_f, cpu_b = run("cpu")
print(f"EfficientViT-{MODEL} training-mode trunk, fp32, batch 2 @1008^2: HIP kernels vs torch stand-ins")
for (n, a), (_, b) in zip(gpu_f, cpu_f):
    print(f"  fwd {n:70s} {tuple(a.shape)!s:22s} rel max-abs diff {float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):.3e}")
for (n, a), (_, b) in zip(gpu_b, cpu_b):
    print(f"  bwd {n:30s} {tuple(a.shape)!s:22s} rel max-abs diff {float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):.3e}")
