"""Development aid: the mask upsampling kernel (288^2 logits -> thresholded uint8 / float32 masks at the image size) in its two forms --
ESAM3_UPSAMPLE_OLD=1 selects the one-thread-per-pixel kernel in dev builds -- compared bit for bit and timed with HIP events.

    ESAM3_DEV_LIB=build_dev/libesam3_dev.so python tools/post_ab.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.environ["ESAM3_DEV_LIB"])
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def run(m, oh, ow, want_f32, old):
    os.environ["ESAM3_UPSAMPLE_OLD"] = "1" if old else "0"
    n = m.shape[0]
    f = torch.full((n, oh, ow), float("nan"), dtype=torch.float32, device="cuda") if want_f32 else None
    u = torch.full((n, oh, ow), 7, dtype=torch.uint8, device="cuda")
    rc = lib.esam3_op_upsample_masks(P(m), P(f), P(u), n, m.shape[1], m.shape[2], oh, ow, C.c_float(0.0), None)
    assert rc == 0
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        lib.esam3_op_upsample_masks(P(m), P(f), P(u), n, m.shape[1], m.shape[2], oh, ow, C.c_float(0.0), None)
    ev[1].record()
    torch.cuda.synchronize()
    return f, u, ev[0].elapsed_time(ev[1]) / 20


for (n, oh, ow, f32) in [(32, 1008, 1008, False), (32, 1024, 1024, True), (3, 600, 800, True), (3, 333, 517, True), (2, 1200, 1801, False)]:
    m = torch.randn(n, 288, 288, generator=torch.Generator().manual_seed(n + oh)).cuda()
    fo, uo, to = run(m, oh, ow, f32, True)
    fn, un, tn = run(m, oh, ow, f32, False)
    same = bool(torch.equal(uo, un)) and (not f32 or bool(torch.equal(fo.view(torch.int32), fn.view(torch.int32))))
    print(f"{n} x {oh} x {ow} f32={f32}: {'bit-identical' if same else 'DIFFERS'}; per-pixel kernel {to * 1e3:.1f} us (incl. op overhead), row kernel {tn * 1e3:.1f} us")
