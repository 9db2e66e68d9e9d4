"""Development aid: the bf16 decoder with / without its fused kernels on the same inputs (dev library: the A/B switches are
environment variables read once per process, so every variant runs in its own process and the results are compared from files).

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so python tools/decoder_fused_check.py run /tmp/a.npz
    ESAM3_DEV_LIB=... ESAM3_NO_TOK_FUSED=1 python tools/decoder_fused_check.py run /tmp/b.npz
    python tools/decoder_fused_check.py cmp /tmp/a.npz /tmp/b.npz
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(path):
    import torch
    from efficientsam3_amd import _lib
    if os.environ.get("ESAM3_DEV_LIB"):
        _lib.LIB_PATH = os.environ["ESAM3_DEV_LIB"]
    from efficientsam3_amd import Sam3Processor, build_efficientsam3_image_model, schema, synth
    sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0)
    model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=True, backbone_type="efficientvit", model_name="b1",
                                            dtype="bf16", state_dict=sd)
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        cases = json.load(f)["cases"]
    out = {}
    proc = Sam3Processor(model)
    for seed in (1, 2):
        img = synth.smooth_image_u8(seed=seed)
        state = proc.set_image(torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0))))
        for name, case in cases.items():
            kw = {k: (np.asarray(v, dtype=np.int32 if k == "point_labels" else np.float32) if isinstance(v, list) else v)
                  for k, v in case["kw"].items() if k in ("point_coords", "point_labels", "box", "multimask_output")}
            try:
                masks, iou, low = model.predict_inst(state, **kw)
            except Exception as e:  # noqa: BLE001
                print("skip", name, e)
                continue
            out[f"{seed}/{name}/low"] = np.asarray(low, dtype=np.float32)
            out[f"{seed}/{name}/iou"] = np.asarray(iou, dtype=np.float32)
    torch.cuda.synchronize()
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    worst = {}
    for k in A.files:
        d = float(np.abs(A[k] - B[k]).max())
        kind = k.rsplit("/", 1)[1]
        worst[kind] = max(worst.get(kind, 0.0), d)
        if kind == "low":
            fa, fb = A[k] > 0, B[k] > 0
            miou = float((fa & fb).sum() / max((fa | fb).sum(), 1))
            worst["1-mask_iou"] = max(worst.get("1-mask_iou", 0.0), 1 - miou)
    print(os.path.basename(a), "vs", os.path.basename(b), {k: round(v, 6) for k, v in worst.items()},
          "logit range", float(min(A[k].min() for k in A.files if k.endswith("low"))), float(max(A[k].max() for k in A.files if k.endswith("low"))))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
