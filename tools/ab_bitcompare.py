"""Development aid: are this round's backbone kernels BIT-identical to the ones they replace on real (non-lattice) data?  The EV-M and
TinyViT-11M encoders run on smooth synthetic images through the development library, once per set of dev-build switches
(esam3_dev_flag reads the environment at every launch), and every output tensor is compared bit for bit against the default dispatch.

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so python tools/ab_bitcompare.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("ESAM3_DEV_LIB"), "needs the development library (make -C efficientsam3_amd/csrc dev)"
from efficientsam3_amd import build_efficientsam3_image_model, schema, synth  # noqa: E402

SWITCHES = ["ESAM3_MLA1_OLD", "ESAM3_MB3S", "ESAM3_STEM_OLD", "ESAM3_KVPREP_OLD"]


def run(eng, x):
    o = eng.encode(x, want_sam3=True, want_sam2=True, want_trunk=True)
    torch.cuda.synchronize()
    flat = {}
    for k, v in o.items():
        if isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                flat[f"{k}[{i}]"] = t.clone()
        elif torch.is_tensor(v):
            flat[k] = v.clone()
    return flat


for backbone, name, B in (("efficientvit", "b1", 4), ("tinyvit", "11m", 2)):
    sd = schema.synthetic_state_dict(backbone, name, seed=0)
    model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=True, backbone_type=backbone, model_name=name,
                                            dtype="bf16", state_dict=sd)
    x = torch.from_numpy(np.stack([synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s)) for s in range(1, B + 1)])).cuda()
    for s in SWITCHES:
        os.environ.pop(s, None)
    ref = run(model.engine, x)
    for sw in SWITCHES + ["ALL"]:
        for s in SWITCHES:
            os.environ.pop(s, None)
        for s in (SWITCHES if sw == "ALL" else [sw]):
            os.environ[s] = "1"
        got = run(model.engine, x)
        diffs = {k: int((got[k].view(torch.int16) != ref[k].view(torch.int16)).sum()) for k in ref}
        bad = {k: v for k, v in diffs.items() if v}
        print(f"{backbone}-{name} B{B}  {sw:18s}: " + ("bit-identical on all %d outputs" % len(ref) if not bad else f"DIFFERS {bad}"))
    for s in SWITCHES:
        os.environ.pop(s, None)


# ---- the PCS detector (text prompts): GroupNorm of the pixel decoder finalised once per (image, group) instead of per thread ----
sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0, enable_inst_interactivity=False)
sd.update(schema.synthetic_text_state_dict("MobileCLIP-S0", 16, seed=0))
sd.update(schema.synthetic_pcs_state_dict(seed=0))
model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit", model_name="b1",
                                        dtype="bf16", state_dict=sd, text_encoder_type="MobileCLIP-S0", text_encoder_context_length=16)
eng = model.engine
B = 3
x = torch.from_numpy(np.stack([synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s)) for s in range(2, 2 + B)])).cuda()
tok = np.zeros((B, 16), dtype=np.int64)
for i in range(B):
    tok[i, :4] = [49406, 1929 + 37 * i, 2368 + i, 49407]
tok_d = torch.from_numpy(tok).cuda()


def ground():
    out = eng.encode(x, want_sam3=True, want_sam2=False)
    mem_t, _ = eng.encode_text(tok_d)
    g = eng.ground(out["sam3_fpn"], mem_t, tok_d == 0)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in g.items() if torch.is_tensor(v)}


os.environ.pop("ESAM3_GN_OLD", None)
ref = ground()
os.environ["ESAM3_GN_OLD"] = "1"
got = ground()
os.environ.pop("ESAM3_GN_OLD", None)
bad = {k: int((got[k].contiguous().view(torch.uint8) != ref[k].contiguous().view(torch.uint8)).sum()) for k in ref}
bad = {k: v for k, v in bad.items() if v}
print(f"PCS ground B{B}  ESAM3_GN_OLD      : " + ("bit-identical on all %d outputs" % len(ref) if not bad else f"DIFFERS {bad}"))
