"""The thread -> element mapping of integral_seq.hip's tilted-sum kernels, replayed in Python against oracle/integral.c (CPU; no GPU needed): k_iseq_tbuf walks the
anti-diagonals j + y = a, k_iseq_tcol0 the first column, k_iseq_tdiag the diagonals j - y = d, all three stepping through the rows in lockstep with per-wave row ranges.
The replay uses exact integers (CV_8U -> CV_32S), so it checks WHICH elements meet, not the float order (that is the GPU suite's bit-for-bit test); it is what the
kernels were designed against, kept as a test of the index arithmetic (wave size, channel interleave, widths 1 / 2, images taller than wide and the reverse)."""
import numpy as np
import pytest

import orc

WAVE = 64


def replay(src):
    H, W = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    P = src.reshape(H, W * cn).astype(np.int64)
    Wn, n = W * cn, (W + H - 1) * cn
    B = np.zeros((H, Wn), np.int64)
    T = np.zeros((H + 1, (W + 1) * cn), np.int64)
    for i0 in range(0, n, WAVE):                                   # k_iseq_tbuf
        aLo, aHi = i0 // cn, min(n - 1, i0 + WAVE - 1) // cn
        yLo, yHi = max(0, aLo - W + 1), min(H - 1, aHi)
        for i in range(i0, min(n, i0 + WAVE)):
            b = 0
            for y in range(yLo, yHi + 1):
                e = i - y * cn
                if e < 0 or e >= Wn:
                    continue
                p = P[y, e]
                b = p if (y == 0 or e >= Wn - cn) else b + p
                B[y, e] = b
    for k in range(cn):                                            # k_iseq_tcol0
        r = 0
        for y in range(H):
            T[y + 1, k] = r
            r = P[0, k] if y == 0 else (r + P[y, k]) + (B[y - 1, cn + k] if W > 1 else 0)
            T[y + 1, cn + k] = r
    for i0 in range(0, n, WAVE):                                   # k_iseq_tdiag
        dLo, dHi = i0 // cn - (H - 1), min(n - 1, i0 + WAVE - 1) // cn - (H - 1)
        yLo, yHi = max(0, -dHi), min(H - 1, W - 1 - dLo)
        for i in range(i0, min(n, i0 + WAVE)):
            r = 0
            for y in range(yLo, yHi + 1):
                e = i - (H - 1 - y) * cn
                if e < 0 or e >= Wn:
                    continue
                if e < cn:
                    r = T[y + 1, cn + e]
                    continue
                p = P[y, e]
                if y == 0:
                    r = p
                else:
                    r = B[y - 1, e] + ((B[y - 1, e + cn] + p) + r) if e < Wn - cn else (p + B[y - 1, e]) + r
                T[y + 1, cn + e] = r
    return T


@pytest.mark.parametrize("shape", [(1, 1), (1, 5), (5, 1, 2), (3, 2), (7, 9, 3), (40, 150), (150, 40, 2), (70, 130), (66, 5, 4)])
def test_the_tilted_kernels_visit_the_right_elements(shape):
    src = np.random.default_rng(sum(shape)).integers(0, 256, shape).astype(np.uint8)
    want = orc.orc_integral(src, 4, 6, False, True)[2].reshape(shape[0] + 1, -1)
    assert np.array_equal(replay(src), want)
