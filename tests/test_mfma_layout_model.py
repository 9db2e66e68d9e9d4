"""A CPU model of the lane layouts of v_mfma_f32_16x16x16_bf16 (A: lane l holds row l % 16, k = 4 (l // 16) .. + 3; B: lane l holds column l % 16,
k = 4 (l // 16) .. + 3; C / D: lane l holds column l % 16, rows 4 (l // 16) .. + 3) and of the chain the fused decoder kernels build on it
(csrc/decoder_fused.hip: i2t_block_kernel, t2i_mfma_kernel): in the TRANSPOSED formulation the accumulator registers of one product are, lane by
lane, the B-operand registers of the next, so nothing has to move between lanes.  The test runs the image -> token chain
    Q^T = Wq X^T  ->  S^T = K Q^T  ->  P = softmax over the keys  ->  O^T = V^T P  ->  Y^T = Wo O^T
with every product evaluated THROUGH the lane model (registers in, registers out, no reshuffling) and compares with dense numpy.  It pins the
layout algebra the kernels rely on; the hardware side of it is covered by the -m gpu operator tests."""
import numpy as np

LANES = 64


def a_regs(A):           # A [16 x 16] (M x K) -> per-lane 4 values
    return np.stack([A[l % 16, 4 * (l // 16):4 * (l // 16) + 4] for l in range(LANES)])


def b_regs(B):           # B [16 x 16] (K x N) -> per-lane 4 values
    return np.stack([B[4 * (l // 16):4 * (l // 16) + 4, l % 16] for l in range(LANES)])


def mfma(a, b, c=None):  # registers in, registers out: D = A B + C in the C layout
    A = np.zeros((16, 16)); B = np.zeros((16, 16))
    for l in range(LANES):
        A[l % 16, 4 * (l // 16):4 * (l // 16) + 4] = a[l]
        B[4 * (l // 16):4 * (l // 16) + 4, l % 16] = b[l]
    D = A @ B
    d = np.stack([D[4 * (l // 16):4 * (l // 16) + 4, l % 16] for l in range(LANES)])
    return d if c is None else d + c


def c_to_dense(c):       # C layout registers -> [16 x 16]
    D = np.zeros((16, 16))
    for l in range(LANES):
        D[4 * (l // 16):4 * (l // 16) + 4, l % 16] = c[l]
    return D


def test_c_layout_is_the_next_b_layout():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((16, 16))
    assert np.array_equal(b_regs(X), np.stack([c for c in mfma(a_regs(np.eye(16)), b_regs(X))]))   # D = I X comes back in B's own layout


def test_image_to_token_chain_stays_in_lane():
    rng = np.random.default_rng(1)
    C, D, HEADS, HD, T, PIX = 64, 32, 2, 16, 10, 16        # a small instance: 64 channels, 2 heads x 16, 10 tokens, one group of 16 pixels
    x = rng.standard_normal((PIX, C))
    wq, wo = rng.standard_normal((D, C)) / 8, rng.standard_normal((C, D)) / 6
    k, v = rng.standard_normal((T, D)), rng.standard_normal((T, D))
    # dense reference
    q = x @ wq.T
    o = np.zeros((PIX, D))
    for h in range(HEADS):
        s = q[:, h * HD:(h + 1) * HD] @ k[:, h * HD:(h + 1) * HD].T * 0.25
        p = np.exp(s - s.max(1, keepdims=True))
        o[:, h * HD:(h + 1) * HD] = (p / p.sum(1, keepdims=True)) @ v[:, h * HD:(h + 1) * HD]
    y_ref = o @ wo.T + x
    # lane model: B fragments of X^T per 16-channel block = the residual in the C layout of Y^T
    xf = [b_regs(x[:, 16 * kb:16 * kb + 16].T) for kb in range(C // 16)]
    ob = []
    for h in range(HEADS):
        acc = None
        for kb in range(C // 16):                                                   # Q_h^T = Wq[16 h ..] X^T
            acc = mfma(a_regs(wq[16 * h:16 * h + 16, 16 * kb:16 * kb + 16]), xf[kb], acc)
        qb = acc * 0.25                                                              # C layout == B layout of the next product
        kpad = np.zeros((16, 16)); kpad[:T] = k[:, 16 * h:16 * h + 16]
        s = mfma(a_regs(kpad), qb)                                                   # S^T[t, pixel]: rows t = 4 g + i
        t_of = np.array([[4 * (l // 16) + i for i in range(4)] for l in range(LANES)])
        s = np.where(t_of < T, s, -np.inf)
        mx = np.full(LANES, -np.inf)
        for l in range(LANES):                                                       # max over t: in-lane + the lanes l, l ^ 16, l ^ 32, l ^ 48
            mx[l] = max(s[m].max() for m in (l, l ^ 16, l ^ 32, l ^ 48))
        pb = np.exp(s - mx[:, None])
        vt = np.zeros((16, 16)); vt[:, :T] = v[:, 16 * h:16 * h + 16].T              # A = V_h^T [dims x tokens]
        num = mfma(a_regs(vt), pb)                                                   # O^T[d, pixel]
        den = mfma(a_regs(np.ones((16, 16))), pb)                                    # every row = sum_t p
        ob.append(num / den[:, :1])
    y = np.zeros((PIX, C))
    for mb in range(C // 16):                                                        # Y^T = Wo O^T + x
        acc = None
        for h in range(HEADS):
            acc = mfma(a_regs(wo[16 * mb:16 * mb + 16, 16 * h:16 * h + 16]), ob[h], acc)
        y[:, 16 * mb:16 * mb + 16] = c_to_dense(acc + xf[mb]).T                      # the residual is the B fragment the first product held
    assert np.abs(y - y_ref).max() <= 1e-10
