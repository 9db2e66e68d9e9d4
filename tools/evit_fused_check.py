"""Development check of the round-4 fused EfficientViT kernels (csrc/evit_fused.hip) on a GPU box: runs the op-level cases of
tests/test_ops_gpu.py (test_mbconv3, test_lite_mla_block) and, instead of asserting, prints where the error sits (by output
channel group, by pixel row / column inside the 8 x 16 tile) so that one gpurun call localises an indexing bug."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from tests import test_ops_gpu as T  # noqa: E402
from tests import util as U  # noqa: E402


def report(name, got, ref):
    diff = (got.double() - ref.double()).abs()
    rel = T._rel_l2(got, ref)
    bound = 6e-2 + 3e-2 * ref.double().abs()
    bad = diff > bound
    print(f"{name}: rel_l2 {rel:.3e} max_abs {float(diff.max()):.3e} ref_max {float(ref.abs().max()):.2f} bad {int(bad.sum())}/{bad.numel()}"
          f" finite {bool(torch.isfinite(got).all())}")
    if rel > 8e-3 or not torch.isfinite(got).all():
        B, C, H, W = got.shape
        d = torch.nan_to_num(diff, nan=1e3)
        print("   per 8-channel group max:", [f"{float(d[:, c:c + 8].max()):.2g}" for c in range(0, C, 8)][:40])
        print("   per row%8 max:", [f"{float(d[:, :, r::8].max()):.2g}" for r in range(min(8, H))])
        print("   per col%16 max:", [f"{float(d[:, :, :, c::16].max()):.2g}" for c in range(min(16, W))])
        print("   per image max:", [f"{float(d[b].max()):.2g}" for b in range(B)])
        print("   per tile-row (H/8) max:", [f"{float(d[:, :, r:r + 8].max()):.2g}" for r in range(0, H, 8)])
        print("   per tile-col (W/16) max:", [f"{float(d[:, :, :, c:c + 16].max()):.2g}" for c in range(0, W, 16)])


def mb(B, H, W, Cin, Cmid, Cout, stride, res):
    mode = "bf16"
    tdt = torch.bfloat16
    x = T._rand(B, Cin, H, W, seed=1)
    w1, b1 = T._rand(Cmid, Cin, 1, 1, seed=2) * (2.0 / Cin) ** 0.5, T._rand(Cmid, seed=3) * 0.1
    wd, bd = T._rand(Cmid, 1, 3, 3, seed=4) * 0.4, T._rand(Cmid, seed=5) * 0.1
    w2, b2 = T._rand(Cout, Cmid, 1, 1, seed=6) / Cmid ** 0.5, T._rand(Cout, seed=7) * 0.1
    q = lambda t: T._q(t, mode)
    xq = q(x)
    m = q(F.hardswish(F.conv2d(xq, q(w1), b1)))
    m = q(F.hardswish(F.conv2d(m, q(wd), bd, stride=stride, padding=1, groups=Cmid)))
    ref = F.conv2d(m, q(w2), b2)
    if res:
        ref = ref + xq
    x_d = U.to_dev_nhwc(x, tdt)
    OH, OW = ref.shape[-2:]
    out = torch.full((B, OH, OW, Cout), float("nan"), dtype=tdt, device="cuda")
    rc = U.lib().esam3_op_mbconv3(U.P(x_d), U.H(U.np32(w1)), U.H(U.np32(b1)), U.H(U.np32(wd)), U.H(U.np32(bd)), U.H(U.np32(w2)),
                                  U.H(U.np32(b2)), U.P(out), B, H, W, Cin, Cmid, Cout, stride, res, None)
    if rc:
        print(f"mbconv3 {Cin}->{Cmid}->{Cout} s{stride}: rc {rc}", U.lib().esam3_last_error())
        return
    report(f"mbconv3 {Cin}->{Cmid}->{Cout} s{stride} {B}x{H}x{W}", U.from_dev_nhwc(out), ref)


def mla(B, H, W, C):
    tdt = torch.bfloat16
    x = T._rand(B, C, H, W, seed=1)
    wqkv = T._rand(3 * C, C, seed=2) / C ** 0.5
    wdw = T._rand(3 * C, 1, 5, 5, seed=3) * 0.2
    wgrp = T._rand(3 * C, 16, seed=4) / 4.0
    wproj = T._rand(C, 2 * C, seed=5) / (2 * C) ** 0.5
    bproj = T._rand(C, seed=6) * 0.1
    ref = T._lite_mla_block_ref(x, wqkv, wdw, wgrp, wproj, bproj)
    x_d = U.to_dev_nhwc(x, tdt)
    out = torch.full((B, H, W, C), float("nan"), dtype=tdt, device="cuda")
    rc = U.lib().esam3_op_lite_mla_block(U.P(x_d), U.H(U.np32(wqkv)), U.H(U.np32(wdw)), U.H(U.np32(wgrp)), U.H(U.np32(wproj)),
                                         U.H(U.np32(bproj)), U.P(out), B, H, W, C, None)
    if rc:
        print(f"lite_mla_block C={C}: rc {rc}", U.lib().esam3_last_error())
        return
    got = U.from_dev_nhwc(out)
    report(f"lite_mla_block C={C} {B}x{H}x{W}", got, ref)
    # the attention branch alone (out - x): the shortcut hides a wrong branch behind a large common term
    report(f"   branch only", got - T._q(x, "bf16"), ref - T._q(x, "bf16"))


if __name__ == "__main__":
    for case in [(2, 40, 40, 16, 64, 32, 2, 0), (1, 30, 26, 32, 128, 32, 1, 1), (1, 41, 23, 32, 128, 64, 2, 0),
                 (2, 33, 50, 64, 256, 64, 1, 1), (2, 21, 19, 64, 256, 128, 2, 0), (1, 16, 16, 128, 512, 128, 1, 1),
                 (2, 63, 63, 128, 512, 128, 1, 1), (1, 31, 29, 128, 512, 256, 2, 0), (1, 9, 9, 256, 1024, 256, 1, 1),
                 (2, 32, 32, 256, 1024, 256, 1, 1)]:
        mb(*case)
    for case in [(1, 8, 16, 128), (2, 21, 19, 128), (1, 63, 63, 128), (1, 9, 9, 256), (2, 32, 32, 256)]:
        mla(*case)
