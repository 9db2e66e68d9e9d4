"""Per-operator roofline table of ONE stage-1 training step ("so the 67 ms has an owner per kernel").

The training step is sequenced from Python (efficientsam3_amd/train_blocks.py, stage1_train.py, stage1.py): every operator wrapper there is
one C-ABI call = one kernel or a short fixed group of kernels.  This tool wraps those functions for ONE step with a pair of HIP events each
(on the launch stream) and prices every call at its algorithmic work:

  * bytes  = every tensor argument read once + the result written once (fp32 parameters / statistics included);
  * flops  = 2 M N K for the linear / conv3x3 forward, data-gradient and weight-gradient GEMMs, 2 k^2 per output element for the depthwise
             convolutions and their gradients (everything else is priced by bytes alone);
  * floor  = max(flops / 2.5 PFLOP/s, bytes / 8 TB/s)  (MI355X_MICROARCH.md),

and prints the table sorted by time, plus the step's total against the sum of the floors.  The events serialise nothing (same stream), but
an instrumented step is a few per cent slower than a plain one: the plain step time is printed beside it.

    python tools/stage1_step_roofline.py [--model b1] [--batch 32] [--dtype bf16] > profiles/r06/roofline_stage1_step_b1_b32.md
"""
import argparse
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientsam3_amd import schema, stage1, stage1_train, train_blocks  # noqa: E402
from efficientsam3_amd.stage1_train import Stage1Trainer  # noqa: E402

PREFIX = "backbone.vision_backbone.trunk.model."
PEAK_TF, PEAK_TB = 2500.0, 8.0
REC = []          # (name, start event, end event, bytes, flops)
ON = [False]
DEPTH = [0]       # only the OUTERMOST wrapped call is recorded (bn_act_forward may call bn_train_forward + act_forward)


def _bytes(obj) -> float:
    if torch.is_tensor(obj):
        return float(obj.numel() * obj.element_size())
    if isinstance(obj, (list, tuple)):
        return sum(_bytes(o) for o in obj)
    if isinstance(obj, dict):
        return sum(_bytes(o) for o in obj.values())
    return 0.0


def _flops(name, args, out) -> float:
    t = [a for a in args if torch.is_tensor(a)]
    try:
        if name in ("linear_forward", "linear_dgrad"):            # (x | dy [.., K], w [N, K] | [K, N]) -> [.., N]
            return 2.0 * t[0].numel() / t[0].shape[-1] * t[1].numel()
        if name == "linear_wgrad":                                  # (dy [.., N], x [.., K]) -> [N, K]
            return 2.0 * (t[0].numel() / t[0].shape[-1]) * t[0].shape[-1] * t[1].shape[-1]
        if name in ("conv3x3_forward", "conv3x3_dgrad"):          # (x NHWC, w [Co, Ci, 3, 3])
            return 2.0 * t[0].numel() / t[0].shape[-1] * t[1].numel()
        if name == "conv3x3_wgrad":                                 # (dy NHWC [.., Co], x NHWC [.., Ci])
            return 2.0 * (t[0].numel() / t[0].shape[-1]) * t[0].shape[-1] * t[1].shape[-1] * 9
        if name in ("dwconv_forward", "dwconv_dgrad", "dwconv_wgrad"):
            k2 = 25.0 if any(torch.is_tensor(a) and a.dim() == 4 and a.shape[-1] == 5 for a in args) else 9.0
            big = max(t, key=lambda a: a.numel())
            return 2.0 * k2 * big.numel()
    except Exception:  # noqa: BLE001
        return 0.0
    return 0.0


def wrap(mod, name):
    fn = getattr(mod, name)

    def inner(*args, **kw):
        if not ON[0] or DEPTH[0] > 0:
            return fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        DEPTH[0] += 1
        try:
            out = fn(*args, **kw)
        finally:
            DEPTH[0] -= 1
        e1.record()
        shapes = " ".join("x".join(str(d) for d in t_.shape) for t_ in list(args) + list(kw.values()) if torch.is_tensor(t_))
        REC.append((name, e0, e1, _bytes(args) + _bytes(kw) + _bytes(out), _flops(name, list(args) + list(kw.values()), out), shapes))
        return out

    inner.__name__ = name
    setattr(mod, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="b1")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--calls", type=int, default=0, help="also list the N slowest single calls with their tensor shapes")
    a = ap.parse_args()
    # leaf operators only (a wrapper that calls other wrapped functions would be counted twice): the C-ABI callers of train_blocks /
    # stage1_train, the loss and the update
    leaves = {train_blocks: ["bn_train_forward", "bn_train_backward", "bn_act_forward", "bn_act_backward", "act_forward", "act_backward",
                             "linear_forward", "linear_dgrad", "linear_wgrad", "dwconv_forward", "dwconv_dgrad", "dwconv_wgrad", "colsum",
                             "channel_scale", "add", "batched_coldot", "lite_mla_forward", "lite_mla_backward", "stem_forward", "stem_im2col"],
              stage1_train: ["conv3x3_forward", "conv3x3_dgrad", "conv3x3_wgrad", "resize_forward", "resize_backward", "distill_loss",
                             "distill_loss_backward"]}
    for mod, names in leaves.items():
        for n in names:
            if hasattr(mod, n):
                wrap(mod, n)
    if a.model.startswith("repvit_"):
        family, name = "repvit", a.model[len("repvit_"):].replace("_", ".")
    elif a.model.startswith("tiny_vit_"):
        family, name = "tinyvit", a.model[len("tiny_vit_"):]
    else:
        family, name = "efficientvit", a.model
    sd = schema.synthetic_state_dict(family, name, seed=0)
    sd = {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}
    tr = Stage1Trainer(sd, a.model, embed_size=72, dtype=a.dtype, lr=1e-4, weight_decay=0.05, clip_grad=5.0, cosine_weight=0.5)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn((a.batch, 3, 1008, 1008), generator=g).cuda()
    teacher = (torch.randn((a.batch, 72, 72, 1024), generator=g) * 0.5).to("cuda", torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    sizes = [(1008, 1008)] * a.batch
    for _ in range(2):
        tr.step(imgs, teacher, sizes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        tr.step(imgs, teacher, sizes)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / 3
    # the update (norm + AdamW launches) is one call on the updater object
    upd = tr.updater.step
    ue = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]

    def upd_timed(*x, **k):
        ue[0].record()
        r = upd(*x, **k)
        ue[1].record()
        return r

    tr.updater.step = upd_timed
    ON[0] = True
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    tr.step(imgs, teacher, sizes)
    s1.record()
    torch.cuda.synchronize()
    ON[0] = False
    total = s0.elapsed_time(s1)
    rows = collections.OrderedDict()
    for nm, e0, e1, by, fl, _shapes in REC:
        r = rows.setdefault(nm, dict(ms=0.0, n=0, by=0.0, fl=0.0))
        r["ms"] += e0.elapsed_time(e1); r["n"] += 1; r["by"] += by; r["fl"] += fl
    n_par = sum(v.numel() for k, v in sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    rows["update (grad norm + clip + AdamW)"] = dict(ms=ue[0].elapsed_time(ue[1]), n=1, by=float(n_par) * (4 + 16 + 12 + 2), fl=0.0)
    covered = sum(r["ms"] for r in rows.values())
    print(f"per-operator roofline of ONE stage-1 training step: student {a.model}, batch {a.batch}, {a.dtype}; instrumented step {total:.2f} ms "
          f"(plain step {plain * 1e3:.2f} ms = {a.batch / plain:.0f} img/s); the operators below cover {covered:.2f} ms, the rest "
          f"({total - covered:.2f} ms) is torch data movement between them (views / cat / copies) and launch gaps\n")
    print("| ms | calls | TFLOP/s | TB/s | bound | floor ms | fraction of the floor | operator (train_blocks.py / stage1_train.py wrapper) |")
    print("|---|---|---|---|---|---|---|---|")
    fsum = 0.0
    for nm, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        t = r["ms"] * 1e-3
        f_m, f_h = r["fl"] / (PEAK_TF * 1e12), r["by"] / (PEAK_TB * 1e12)
        floor = max(f_m, f_h)
        fsum += floor
        print(f"| {r['ms']:.3f} | {r['n']} | {r['fl'] / t / 1e12 if t else 0:.0f} | {r['by'] / t / 1e12 if t else 0:.2f} | "
              f"{'mfma' if f_m >= f_h else 'hbm'} | {floor * 1e3:.3f} | {floor / t if t else 0:.2f} | `{nm}` |")
    print(f"\nsum of the floors {fsum * 1e3:.2f} ms = {fsum * 1e3 / total:.3f} of the instrumented step")
    if a.calls:
        print(f"\nthe {a.calls} slowest single calls (ms, GB/s of algorithmic bytes, operator, tensor argument shapes):\n")
        for nm, e0, e1, by, fl, shapes in sorted(REC, key=lambda r: -r[1].elapsed_time(r[2]))[:a.calls]:
            t = e0.elapsed_time(e1)
            print(f"    {t:7.3f} ms  {by / t / 1e6:7.0f} GB/s  {nm:20s} {shapes}")


if __name__ == "__main__":
    main()
