"""GPU timeline of API-level steps (Sam3Processor.set_image_batch + model.predict_inst_batch): where the device idles.

    cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d <dir> -o t --output-format csv -- python tools/api_timeline.py run
    python tools/api_timeline.py report <dir>

`run` is the workload (3 warm + 4 measured steps, the host-side wall time of each printed); `report` merges the kernel and
memory-copy traces, splits them into steps at the first host-to-device copy of each set_image_batch, and prints per step: span,
busy time, and every idle gap above 40 us with the activities on both sides."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    import numpy as np
    import torch
    from PIL import Image
    sys.path.insert(0, ROOT)
    from efficientsam3_amd import Sam3Processor, build_efficientsam3_image_model, schema, synth
    sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0)
    model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=True, backbone_type="efficientvit", model_name="b1",
                                            dtype="bf16", state_dict=sd)
    B = 32
    rng_img = np.random.default_rng(0).integers(0, 256, (4, 1024, 1024, 3), dtype=np.uint8)
    pil = [Image.fromarray(rng_img[i % 4]) for i in range(B)]
    proc = Sam3Processor(model)
    pts, labels, boxes = synth.prompts(B, seed=2)
    sx = 1024.0 / 1008.0
    pcs, bxs, lbl = [pts[i] * sx for i in range(B)], [boxes[i] * sx for i in range(B)], [labels[i] for i in range(B)]
    out = None
    for i in range(7):
        t0 = time.perf_counter()
        st = proc.set_image_batch(pil)
        t1 = time.perf_counter()
        out = model.predict_inst_batch(st, point_coords_batch=pcs, point_labels_batch=lbl, box_batch=bxs, multimask_output=False)
        t2 = time.perf_counter()
        print(f"host step {i}: set_image_batch {1e3 * (t1 - t0):.2f} ms, predict_inst_batch {1e3 * (t2 - t1):.2f} ms", flush=True)
    torch.cuda.synchronize()


def report(d):
    acts = []
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + name[:60]))
    for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:40]))
    acts.sort()
    # a step starts at a host-to-device copy that follows a device-to-host copy (the previous step's result hand-back)
    starts, seen_d2h = [], True
    for i, a in enumerate(acts):
        if a[2].startswith("C") and "HOST_TO_DEVICE" in a[2].upper() and seen_d2h and (a[1] - a[0]) > 100_000:
            starts.append(i)
            seen_d2h = False
        if a[2].startswith("C") and "DEVICE_TO_HOST" in a[2].upper() and (a[1] - a[0]) > 100_000:
            seen_d2h = True
    print(f"{len(acts)} activities, {len(starts)} steps found")
    for k in range(max(0, len(starts) - 4), len(starts)):
        lo, hi = starts[k], (starts[k + 1] if k + 1 < len(starts) else len(acts))
        seg = acts[lo:hi]
        t0 = seg[0][0]
        end, busy, gaps = seg[0][1], 0, []
        cur_s, cur_e = seg[0][0], seg[0][1]
        for s, e, n in seg[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                if s - cur_e > 40_000:
                    gaps.append((cur_e - t0, s - cur_e, n))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        nxt = acts[hi][0] if hi < len(acts) else cur_e
        print(f"step {k}: first activity -> last end {1e-6 * (cur_e - t0):.2f} ms, busy {1e-6 * busy:.2f} ms, to the next step's first copy "
              f"{1e-6 * (nxt - t0):.2f} ms")
        copies = [(s - t0, e - s, n) for s, e, n in seg if n.startswith("C") and e - s > 50_000]
        for s, dur, n in copies:
            print(f"    copy at {1e-6 * s:7.2f} ms for {1e-6 * dur:5.2f} ms  {n}")
        for at, g, n in gaps:
            print(f"    idle at {1e-6 * at:7.2f} ms for {1e-6 * g:5.2f} ms  before {n}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
