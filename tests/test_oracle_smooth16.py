"""The CV_16U sigma-0 Gaussian restated as plain integer binomial sums with one rounding, pinned against the real reference (oracle/_ref, built from
/root/reference by oracle/ref/Makefile) -- TEST INFRASTRUCTURE: this is the checker of tests/test_filters_gpu.py::test_gaussian_16u_binomial_on_the_rolling_kernel."""
import numpy as np
import pytest

import orc


W16 = {3: [1, 2, 1], 5: [1, 4, 6, 4, 1], 7: [2, 7, 14, 18, 14, 7, 2], 9: [4, 13, 30, 51, 60, 51, 30, 13, 4]}     # smooth.dispatch.cpp:89-145 times 4 / 16 / 64 / 256


def border_index(p, n, border):
    """cv::borderInterpolate (core/src/copy.cpp:748-793) for an index array; -1 = the constant border"""
    p = np.asarray(p, np.int64).copy()
    if border == 0:
        return np.where((p >= 0) & (p < n), p, -1)
    if border == 1:
        return np.clip(p, 0, n - 1)
    if border == 3:
        return np.mod(p, n)
    if n == 1:
        return np.zeros_like(p)
    d = 1 if border == 4 else 0
    for _ in range(64):
        lo, hi = p < 0, p >= n
        if not (lo.any() or hi.any()):
            break
        p = np.where(lo, -p - 1 + d, p)
        p = np.where(p >= n, n - 1 - (p - n) - d, p)
    return p


def np_binom16(src, k, border):
    """CV_16U sigma-0 Gaussian of any channel count, k in {3, 5, 7, 9}: the plain integer sums with ONE rounding, (S + 2^(2s-1)) >> 2s with the taps' sum 2^s -- what the
    reference's Q16.16 passes (fixedSmoothInvoker<uint16_t, ufixedpoint32>, exact products, one rounding in the column pass) evaluate to; every border rule incl. WRAP,
    images smaller than the kernel (repeated reflection)"""
    w = np.array(W16[k], np.int64)
    sh = {3: 2, 5: 4, 7: 6, 9: 8}[k]
    r = k // 2
    a = src.astype(np.int64)
    H, W = a.shape[:2]
    xi = [border_index(np.arange(W) + i - r, W, border) for i in range(k)]
    yi = [border_index(np.arange(H) + j - r, H, border) for j in range(k)]

    def take(arr, idx, axis):
        t = np.take(arr, np.maximum(idx, 0), axis=axis)
        m = (idx >= 0).reshape([-1 if ax == axis else 1 for ax in range(arr.ndim)])
        return t * m
    h = sum(w[i] * take(a, xi[i], 1) for i in range(k))
    v = sum(w[j] * take(h, yi[j], 0) for j in range(k))
    return ((v + (1 << (2 * sh - 1))) >> (2 * sh)).astype(np.uint16)


def test_16u_binomial_restatement_equals_the_reference():
    if orc.load_ref() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(1)
    for (w, h) in [(37, 23), (64, 5), (5, 64), (333, 41), (1, 9), (9, 1), (2, 2), (3, 3), (1, 1), (2, 1), (3, 1), (1, 3), (5, 5), (7, 7), (256, 128)]:
        for cn in (1, 2, 3, 4):
            if cn > 1 and (w, h) not in ((37, 23), (3, 3), (256, 128), (2, 2)):
                continue
            src = rng.integers(0, 65536, (h, w, cn) if cn > 1 else (h, w)).astype(np.uint16)
            src.flat[:: max(1, src.size // 7)] = 65535
            for k in (3, 5, 7, 9):
                for border in (0, 1, 2, 3, 4):
                    want = orc.ref_GaussianBlur(src, k, 0.0, 0.0, border | 16)          # BORDER_ISOLATED, as the reference's own bit-exact test calls it
                    kw = 1 if (w == 1 and border != 0) else k              # smooth.dispatch.cpp:623-630: a one-pixel dimension clamps the kernel
                    kh = 1 if (h == 1 and border != 0) else k
                    if kw != k or kh != k:
                        continue                                            # (not a square kernel any more: the hook is not asked, the reference's separable path is not restated)
                    assert np.array_equal(np_binom16(src, k, border), want), (w, h, cn, k, border)
    full = np.full((100, 100), 65535, np.uint16)                            # GaussianBlur_Bitexact.overflow_20121
    assert np_binom16(full, 9, 4).min() == 65535 and np.array_equal(np_binom16(full, 9, 4), orc.ref_GaussianBlur(full, 9, 0.0, 0.0, 4))
