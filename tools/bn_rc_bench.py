#!/usr/bin/env python3
"""BatchNorm + activation training passes, saved-output form against the recomputing form (esam3_bn_act_train_backward_rc), timed per call.

    python tools/bn_rc_bench.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from efficientsam3_amd import stage1  # noqa: E402


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (rows, c, act) in [(32 * 504 * 504, 64, "hswish"), (32 * 252 * 252, 128, "hswish"), (32 * 126 * 126, 256, "hswish"), (32 * 63 * 63, 512, "hswish"),
                       (32 * 252 * 252, 96, "gelu"), (32 * 126 * 126, 224, "gelu"), (32 * 252 * 252, 128, "gelu")]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(rows // 64, 64, c, generator=g).to(torch.bfloat16).cuda()
    dy = torch.randn(rows // 64, 64, c, generator=g).to(torch.bfloat16).cuda()
    gamma, beta = (torch.rand(c) + 0.5).cuda(), (torch.randn(c) * 0.3).cuda()
    rm, rv = torch.zeros(c).cuda(), torch.ones(c).cuda()
    y, a_, m, r = stage1.bn_act_train_forward(x, gamma, beta, rm, rv, 0.1, 1e-5, act)
    gb = x.numel() * 2 / 1e9
    t_f1 = timed(lambda: stage1.bn_act_train_forward(x, gamma, beta, rm, rv, 0.1, 1e-5, act))
    t_f0 = timed(lambda: stage1.bn_act_train_forward(x, gamma, beta, rm, rv, 0.1, 1e-5, act, keep_pre=False))
    t_b1 = timed(lambda: stage1.bn_act_train_backward(x, dy, y, act, gamma, m, r))
    t_b0 = timed(lambda: stage1.bn_act_train_backward(x, dy, None, act, gamma, m, r, beta=beta))
    print(f"rows {rows} C {c} {act}: tensor {gb:.3f} GB | forward saved {t_f1:.3f} ms, no y {t_f0:.3f} ms | backward saved {t_b1:.3f} ms, recompute {t_b0:.3f} ms")
