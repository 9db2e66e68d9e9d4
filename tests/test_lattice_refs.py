"""CPU: the exact-lattice references of tests/test_lattice_gpu.py are the SAME functions as the rounding-point formulas of
tests/test_ops_gpu.py evaluated on lattice inputs -- the gather / ReLU / integer shortcuts the lattice tests use to state the
expected output are checked here against the full formulas (true softmax, Hardswish, fp64 division), so a GPU failure of a lattice
test is a kernel defect, not a mis-stated expectation."""
import torch
import torch.nn.functional as F

from tests import test_lattice_gpu as L
from tests import test_ops_gpu as O


def test_mbconv_lattice_is_the_rounding_point_formula():
    B, H, W, Cin, Cmid, Cout, stride, res = 1, 17, 13, 32, 128, 64, 2, 0
    x, w1, b1, wd, bd, w2, b2, ref = L.mbconv_lattice(B, H, W, Cin, Cmid, Cout, stride, res)
    q = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    m = q(F.hardswish(F.conv2d(q(x), q(w1)[:, :, None, None], b1)))
    m = q(F.hardswish(F.conv2d(m, q(wd), bd, stride=stride, padding=1, groups=Cmid)))
    assert torch.equal(F.conv2d(m, q(w2)[:, :, None, None], b2), ref.float())


def test_mla_lattice_is_the_rounding_point_formula():
    x, wqkv, wdw, wgrp, wsel, ref = L.mla_lattice(1, 21, 19, 128)
    full = O._lite_mla_block_ref(x, wqkv, wdw, wgrp, wsel, torch.zeros(128)).to(torch.bfloat16).float()
    # the formula divides in fp64 and rounds once, the lattice reference (like the kernel) divides in fp32 first: equal except where the
    # fp32 quotient lands on a bf16 rounding tie
    assert float((full != ref).float().mean()) < 1e-3 and float((full - ref).abs().max()) <= 2.0 ** -6


def test_t2i_lattice_is_softmax_attention():
    q, k, v, ref = L.t2i_lattice(2, 5, 300)
    split = lambda t: t.double().reshape(t.shape[0], t.shape[1], 8, 16).transpose(1, 2)  # noqa: E731
    full = F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(2, 5, 128)
    assert float((full - ref.double()).abs().max()) < 1e-12


def test_i2t_lattice_is_the_block_formula():
    for q_from in ("x", "table"):
        a = L.i2t_lattice(2, 48, 9, q_from)
        full = O._i2t_block_ref(a["x"], a["wq"], a["bq"], a["peq"], a["wo"], a["bo"], a["gamma"], a["beta"], a["tk"], a["tv"])
        assert float((full.double() - a["ref"]).abs().max()) < 1e-5, q_from
