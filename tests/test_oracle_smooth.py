"""Pins oracle/smooth.c (the CPU restatement) -- CPU only.

 1. against the reference's own known-answer definition: GaussianBlur_Bitexact.Linear8U
    (modules/imgproc/test/test_smooth_bitexact.cpp:139-173) -- same sizes, channels, kernels
    (incl. the literal sigma!=0 Q8.8 kernels :14-27), all five isolated border modes, `eval` :37-50;
 2. against the real reference (oracle/_ref) on RNG inputs;
 3. against committed golden vectors generated from the real reference (tests/golden/gen_golden.py).
"""
import os

import numpy as np
import pytest

import refpatterns as rp

V_U8 = {1: [256], 3: [64, 128, 64], 5: [16, 64, 96, 64, 16], 7: [8, 28, 56, 72, 56, 28, 8],
        9: [4, 13, 30, 51, 60, 51, 30, 13, 4]}
S175, S0875, S0375, S075 = [81, 94, 81], [65, 126, 65], [0, 7, 242, 7, 0], [4, 56, 136, 56, 4]

# (cn, (w,h), kx, ky)  -- test_smooth_bitexact.cpp:141-166
LINEAR8U = [(1, (1, 1), 3), (1, (2, 2), 3), (1, (3, 1), 3), (1, (1, 3), 3), (1, (3, 3), 3), (1, (3, 3), 5),
            (1, (3, 3), 7), (1, (5, 5), 3), (1, (5, 5), 5), (1, (3, 5), 5), (1, (5, 5), 7), (1, (7, 7), 7),
            (1, (256, 128), 3), (2, (256, 128), 3), (3, (256, 128), 3), (4, (256, 128), 3),
            (1, (256, 128), 5), (1, (256, 128), 7), (1, (256, 128), 9)]
LINEAR8U = [(cn, sz, V_U8[k], V_U8[k]) for cn, sz, k in LINEAR8U] + \
           [(cn, (256, 128), S175, S0875) for cn in (1, 2, 3, 4)] + [(1, (256, 128), S0375, S075)]


@pytest.mark.parametrize("border", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("case", range(len(LINEAR8U)))
def test_oracle_matches_reference_known_answer(orc, case, border):
    cn, (w, h), kx, ky = LINEAR8U[case]
    big = rp.smooth_bitexact_pattern(h + 20, w + 20, cn)
    roi = np.ascontiguousarray(big[10:10 + h, 10:10 + w])
    want = rp.eval_fixed(roi, kx, ky, border)
    got = orc.orc_sepSmoothFixedU8(roi, kx, ky, border)
    assert np.array_equal(got, want)


def test_border_interpolate_matches_reference(orc, ref):
    o = orc.oracle()
    for border in range(5):
        for length in (1, 2, 3, 5, 8, 17):
            for p in range(-40, 60):
                if border == 3 and False:
                    continue
                assert o.orc_borderInterpolate(p, length, border) == ref.ref_borderInterpolate(p, length, border), (p, length, border)


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
@pytest.mark.parametrize("ksize", [3, 5, 7, 9])
def test_oracle_matches_real_reference(orc, ref, cn, ksize):
    for (w, h) in [(1, 1), (2, 3), (7, 5), (33, 17), (64, 48), (131, 67)]:
        shape = (h, w, cn) if cn > 1 else (h, w)
        src = orc.ref_rng_fill(shape, np.uint8, 809564 + w * 31 + h, 0, 256)
        for border in (0, 1, 2, 3, 4):
            want = orc.ref_GaussianBlur(src, ksize, 0, 0, border | 16)
            kw = ksize if (w > 1 or border == 0) else 1     # smooth.dispatch.cpp:623-630 (1-pixel dims clamp the kernel)
            kh = ksize if (h > 1 or border == 0) else 1
            got = orc.orc_sepSmoothFixedU8(src, V_U8[kw], V_U8[kh], border)
            assert np.array_equal(got, want), (w, h, cn, ksize, border)


def test_oracle_matches_golden_vectors(orc):
    path = os.path.join(os.path.dirname(__file__), "golden", "gaussian_u8.npz")
    g = np.load(path)
    n = int(g["n"])
    assert n >= 20
    for i in range(n):
        src, want = g[f"src{i}"], g[f"dst{i}"]
        ksize, border = int(g[f"ksize{i}"]), int(g[f"border{i}"])
        got = orc.orc_gaussianBlurBinomialU8(src, ksize, border)
        assert np.array_equal(got, want), i


SIGMAS = [0.3, 0.375, 0.5, 0.75, 0.8, 0.875, 1.0, 1.1, 1.5, 1.75, 2.0, 2.5, 3.3, 4.0, 5.0, 7.7, 0.0, -1.0]


def test_gaussian_taps_bit_identical_to_reference(orc, ref):
    """a2: oracle tap generator == cv::getGaussianKernel(CV_64F) bit for bit."""
    for n in (1, 3, 5, 7, 9, 11, 13, 15, 21, 31, 4, 20):
        for s in SIGMAS:
            a, b = orc.orc_getGaussianKernel(n, s), orc.ref_getGaussianKernel(n, s)
            assert a.tobytes() == b.tobytes(), (n, s)


def test_q8_taps_match_reference_literals(orc):
    """test_smooth_bitexact.cpp:22-26 literal Q8.8 kernels for sigma 1.75, 0.875, 0.375, 0.75."""
    assert list(orc.orc_getGaussianKernelQ(3, 1.75)) == S175
    assert list(orc.orc_getGaussianKernelQ(3, 0.875)) == S0875
    assert list(orc.orc_getGaussianKernelQ(5, 0.375)) == S0375
    assert list(orc.orc_getGaussianKernelQ(5, 0.75)) == S075
    for k, taps in V_U8.items():
        assert list(orc.orc_getGaussianKernelQ(k, 0)) == taps


def test_product_tap_generator_matches_oracle(orc):
    """host logic of the product (runs without a GPU): mi355cv_getGaussianKernel[Q] == oracle."""
    import opencv_amd as cv
    for n in (1, 3, 5, 7, 9, 11, 13, 15, 21, 31):
        for s in SIGMAS:
            assert cv.getGaussianKernel(n, s, cv.CV_64F).tobytes() == orc.orc_getGaussianKernel(n, s).tobytes()
            assert list(cv.getGaussianKernelQ(n, s, 8)) == list(orc.orc_getGaussianKernelQ(n, s, 8))
            assert list(cv.getGaussianKernelQ(n, s, 16)) == list(orc.orc_getGaussianKernelQ(n, s, 16))


@pytest.mark.parametrize("cn", [1, 3])
def test_oracle_sigma_path_matches_real_reference(orc, ref, cn):
    """cv::GaussianBlur(8U, sigma>0) end to end == oracle taps + oracle fixed-point smoother."""
    for (w, h) in [(33, 17), (64, 48)]:
        shape = (h, w, cn) if cn > 1 else (h, w)
        src = orc.ref_rng_fill(shape, np.uint8, 4242 + w, 0, 256)
        for (kw, kh, s1, s2) in [(3, 3, 1.75, 0.875), (5, 5, 0.375, 0.75), (7, 5, 1.2, 0), (11, 11, 2.0, 2.0),
                                 (0, 0, 1.5, 0), (0, 0, 0.8, 2.2), (9, 3, 0, 0)]:
            for border in (0, 1, 2, 4):
                want = orc.ref_GaussianBlur(src, (kw, kh), s1, s2, border | 16)
                sy = s2 if s2 > 0 else s1
                kkw = kw if kw > 0 else (int(np.rint(s1 * 6 + 1)) | 1)
                kkh = kh if kh > 0 else (int(np.rint(sy * 6 + 1)) | 1)
                kx = orc.orc_getGaussianKernelQ(kkw, max(s1, 0))
                ky = orc.orc_getGaussianKernelQ(kkh, max(sy, 0))
                got = orc.orc_sepSmoothFixedU8(src, kx, ky, border)
                assert np.array_equal(got, want), (kw, kh, s1, s2, border)


@pytest.mark.parametrize("cn", [1, 3])
def test_oracle_sigma_path_long_kernels_match_real_reference(orc, ref, cn):
    """the same for the kernel lengths the LDS-ring kernel serves since round 6 (11 .. 129 taps: cv::GaussianBlur on CV_8U with sigma >= 1.7 is an everyday call),
    on images larger and smaller than the kernel"""
    for (w, h) in [(200, 150), (40, 23)]:
        shape = (h, w, cn) if cn > 1 else (h, w)
        src = orc.ref_rng_fill(shape, np.uint8, 99 + w, 0, 256)
        for (kw, kh, s1, s2) in [(0, 0, 3.0, 0), (33, 33, 5.5, 0), (0, 0, 10.0, 4.0), (129, 65, 20.0, 0), (19, 19, 0, 0)]:
            if max(kw, kh, int(s1 * 6 + 1)) // 2 >= min(w, h) and cn == 3:
                continue
            for border in (0, 1, 2, 4):
                want = orc.ref_GaussianBlur(src, (kw, kh), s1, s2, border | 16)
                sy = s2 if s2 > 0 else s1
                kkw = kw if kw > 0 else (int(np.rint(s1 * 6 + 1)) | 1)
                kkh = kh if kh > 0 else (int(np.rint(sy * 6 + 1)) | 1)
                kx = orc.orc_getGaussianKernelQ(kkw, max(s1, 0))
                ky = orc.orc_getGaussianKernelQ(kkh, max(sy, 0))
                assert sum(kx) == 256 and sum(ky) == 256
                got = orc.orc_sepSmoothFixedU8(src, kx, ky, border)
                assert np.array_equal(got, want), (kw, kh, s1, s2, border)
