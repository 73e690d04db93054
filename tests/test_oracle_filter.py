"""Pins oracle/filter.c against the real reference (CPU only).

Integer paths (8U filter2D via the FMA chain on exactly representable taps, sepFilter2D bit-exact modes, Sobel 8U->16S,
box 8U) must agree bit for bit; float paths within the 1e-4 relative norm of the parity contract."""
import numpy as np
import pytest

SHARPEN = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32)
BORDERS = [0, 1, 2, 4]          # filter engines reject BORDER_WRAP (FilterEngine__start asserts)


def rnd(orc, shape, dtype, seed):
    hi = {np.uint8: 256, np.uint16: 65536, np.int16: 32767, np.float32: 1.0}[dtype]
    lo = -32768 if dtype == np.int16 else 0
    return orc.ref_rng_fill(shape, dtype, seed, lo, hi)


@pytest.mark.parametrize("cn", [1, 3])
def test_filter2d_8u_exact_taps(orc, ref, cn):
    for (w, h) in [(64, 9), (96, 33), (128, 5)]:        # widths the reference processes entirely in its SIMD body
        src = rnd(orc, (h, w, cn) if cn > 1 else (h, w), np.uint8, 11 + w)
        for k, anchor, delta in [(SHARPEN, (-1, -1), 0.0), (np.ones((3, 3), np.float32) * 0.125, (-1, -1), 3.0),
                                 (np.array([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], np.float32) * 0.25, (0, 2), 128.0),
                                 (np.array([[0.5, 0.25, 0.125, 0.0625, 0.0625]], np.float32), (1, 0), 0.0)]:
            for border in BORDERS:
                want = orc.ref_filter2D(src, -1, k, anchor, delta, border)
                got = orc.orc_filter2D(src, -1, k, anchor, delta, border)
                assert np.array_equal(got, want), (w, h, cn, border, k.shape)


def test_filter2d_other_depths(orc, ref):
    rng = np.random.default_rng(5)
    k5 = (rng.uniform(-3, 10, (5, 5)) / 37.0).astype(np.float32)      # perf_filter2d.cpp:31-34 style kernel
    for dtype, ddepth, tol in [(np.uint8, 3, 0), (np.uint8, 5, 1e-6), (np.uint16, -1, 0), (np.int16, -1, 0), (np.float32, -1, 1e-6),
                               (np.uint8, -1, 0)]:
        src = rnd(orc, (37, 64, 3), dtype, 77)
        for border in BORDERS:
            want = orc.ref_filter2D(src, ddepth, k5, (-1, -1), 0.5, border)
            got = orc.orc_filter2D(src, ddepth, k5, (-1, -1), 0.5, border)
            if want.dtype == np.float32:
                assert orc.rel_err(got, want) <= 1e-6
            else:   # general float taps: an exact tie may round the other way in the CPU's scalar tail
                assert int(np.max(np.abs(got.astype(np.int64) - want.astype(np.int64)))) <= 1
                assert np.mean(got != want) < 1e-3


def test_sepfilter_bitexact_modes(orc, ref):
    src = rnd(orc, (41, 80, 3), np.uint8, 9)
    smooth3, smooth5 = [0.25, 0.5, 0.25], [0.0625, 0.25, 0.375, 0.25, 0.0625]
    for kx, ky in [(smooth3, smooth3), (smooth5, smooth3), (smooth5, smooth5)]:
        for border in BORDERS:
            want = orc.ref_sepFilter2D(src, -1, kx, ky, (-1, -1), 0.0, border)
            got = orc.orc_sepFilter2D(src, -1, kx, ky, (-1, -1), 0.0, border)
            assert np.array_equal(got, want)
    for kx, ky in [([-1, 0, 1], [1, 2, 1]), ([1, 2, 1], [-1, 0, 1]), ([1, -2, 1], [3, 10, 3])]:
        for border in BORDERS:
            want = orc.ref_sepFilter2D(src, 3, kx, ky, (-1, -1), 2.0, border)
            got = orc.orc_sepFilter2D(src, 3, kx, ky, (-1, -1), 2.0, border)
            assert np.array_equal(got, want)


@pytest.mark.parametrize("ksize", [1, 3, 5, 7, -1])
def test_sobel_scharr(orc, ref, ksize):
    src8 = rnd(orc, (33, 70), np.uint8, 3)
    srcf = rnd(orc, (33, 70, 3), np.float32, 4)
    for dx, dy in [(1, 0), (0, 1)] + ([(1, 1), (2, 0)] if ksize >= 3 else []):
        for border in (1, 2, 4):
            want = orc.ref_Sobel(src8, 3, dx, dy, ksize, 1.0, 0.0, border)          # 8U -> 16S exact
            got = orc.orc_Sobel(src8, 3, dx, dy, ksize, 1.0, 0.0, border)
            assert np.array_equal(got, want), (ksize, dx, dy, border)
            sc = 1.0 / (255.0 * 2 * 4)
            want = orc.ref_Sobel(src8, 5, dx, dy, ksize, sc, 0.0, border)            # 8U -> 32F scaled (cornerHarris)
            got = orc.orc_Sobel(src8, 5, dx, dy, ksize, sc, 0.0, border)
            assert orc.rel_err(got, want) <= 1e-6
            want = orc.ref_Sobel(srcf, -1, dx, dy, ksize, 1.0, 0.25, border)
            got = orc.orc_Sobel(srcf, -1, dx, dy, ksize, 1.0, 0.25, border)
            assert orc.rel_err(got, want) <= 1e-6


def test_sepfilter_float_general(orc, ref):
    rng = np.random.default_rng(1)
    src = rnd(orc, (29, 64, 3), np.float32, 8)
    kx = rng.uniform(-1, 1, 7); ky = rng.uniform(-1, 1, 5)
    for border in BORDERS:
        want = orc.ref_sepFilter2D(src, -1, kx, ky, (2, 1), 0.1, border)
        got = orc.orc_sepFilter2D(src, -1, kx.astype(np.float32), ky.astype(np.float32), (2, 1), 0.1, border)
        assert orc.rel_err(got, want) <= 1e-6


def test_boxfilter(orc, ref):
    for dtype, ddepth in [(np.uint8, -1), (np.uint8, 5), (np.float32, -1), (np.uint16, -1), (np.int16, -1),
                          (np.uint8, 3), (np.uint8, 2), (np.uint16, 0), (np.uint16, 3), (np.uint16, 5), (np.int16, 5)]:
        for wdt in (66, 67, 69):
            src = rnd(orc, (31, wdt, 3), dtype, 21)
            for ksize, anchor in [((3, 3), (-1, -1)), ((5, 5), (-1, -1)), ((2, 2), (-1, -1)), ((7, 3), (1, 2)), ((16, 16), (-1, -1)), ((17, 17), (-1, -1))]:
                for normalize in (True, False):
                    for border in (0, 1, 4):
                        want = orc.ref_boxFilter(src, ddepth, ksize, anchor, normalize, border)
                        got = orc.orc_boxFilter(src, ddepth, ksize, anchor, normalize, border)
                        if want.dtype == np.float32 and dtype == np.float32:
                            assert orc.rel_err(got, want) <= 2e-7, (dtype, ksize, normalize, border)
                        else:   # incl. the int32-sum normalisation: SIMD body in float, the last (w*cn) % 8 elements of a row in double
                            assert np.array_equal(got, want), (dtype, ksize, normalize, border)


def test_boxfilter_more_than_four_channels(orc, ref):
    """5 and 9 channels (Imgproc_FilterSupportedFormats blurs 5): the sums are per element, the normalisation's vector body / scalar tail split is by element count"""
    for dtype, ddepth in [(np.uint8, -1), (np.uint8, 5), (np.float32, -1), (np.uint16, -1), (np.int16, 5)]:
        for cn in (5, 9):
            src = rnd(orc, (31, 47, cn), dtype, 30 + cn)
            for ksize in [(3, 3), (11, 11), (4, 7)]:
                for normalize in (True, False):
                    for border in (0, 1, 4):
                        want = orc.ref_boxFilter(src, ddepth, ksize, (-1, -1), normalize, border)
                        got = orc.orc_boxFilter(src, ddepth, ksize, (-1, -1), normalize, border)
                        if want.dtype == np.float32 and dtype == np.float32:
                            assert orc.rel_err(got, want) <= 2e-7, (dtype, cn, ksize, normalize, border)
                        else:
                            assert np.array_equal(got, want), (dtype, cn, ksize, normalize, border)


def test_boxfilter_into_64f(orc, ref):
    """CV_64F destinations: integer sources are the exact int sum times 1 / area in double (ColumnSum<int, double>) -- bit for bit; float / double sources are summed
    in double, where the reference's running row / column sums differ from a direct window sum in the last bits (1e-13 relative)"""
    rng = np.random.default_rng(1)
    for dtype, dd in [(np.uint8, 6), (np.uint16, 6), (np.int16, 6), (np.float32, 6), (np.float64, 6), (np.float64, -1)]:
        for cn in (1, 3, 5):
            src = (rng.random((31, 47, cn)) * 200 - 50).astype(dtype) if dtype in (np.float32, np.float64) else rng.integers(0, 200, (31, 47, cn)).astype(dtype)
            for ks in ((3, 3), (11, 11), (4, 7)):
                for norm in (True, False):
                    for border in (0, 1, 4):
                        a = orc.orc_boxFilter(src, dd, ks, normalize=norm, border=border)
                        b = orc.ref_boxFilter(src, dd, ks, normalize=norm, border=border)
                        assert a.dtype == b.dtype == np.float64
                        if dtype in (np.float32, np.float64):
                            assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(b).max()), (dtype, cn, ks, norm, border)
                        else:
                            assert np.array_equal(a, b), (dtype, cn, ks, norm, border)


def test_sepfilter_long_kernels(orc, ref):
    """34-129 taps per axis (Imgproc_GaussianBlur.regression_11303 asks for 71): general float kernels, a Gaussian on CV_32F, and a symmetric smoothing kernel on CV_8U (the
    fixed-point engine with its float vector body)"""
    rng = np.random.default_rng(9)
    g71 = np.exp(-0.5 * ((np.arange(71) - 35) / 8.64421) ** 2); g71 = (g71 / g71.sum()).astype(np.float32)
    g41 = np.exp(-0.5 * ((np.arange(41) - 20) / 6.5) ** 2); g41 = (g41 / g41.sum()).astype(np.float32)
    kx, ky = rng.uniform(-1, 1, 41).astype(np.float32) / 8, rng.uniform(-1, 1, 37).astype(np.float32) / 8
    for dtype, ddepth in [(np.float32, -1), (np.uint8, -1), (np.uint8, 5), (np.uint16, 5), (np.int16, -1)]:
        for cn in (1, 3):
            src = rnd(orc, (53, 90, cn) if cn > 1 else (53, 90), dtype, 60 + cn)
            for (a, b) in ((g71, g71), (g41, g71), (kx, ky), (g41, ky[:5])):
                for border in (4, 0, 1):
                    want = orc.ref_sepFilter2D(src, ddepth, a, b, (-1, -1), 0.0, border)
                    got = orc.orc_sepFilter2D(src, ddepth, a, b, (-1, -1), 0.0, border)
                    if want.dtype == np.float32:
                        assert orc.rel_err(got, want) <= 1e-6, (dtype, ddepth, cn, len(a), len(b), border)
                    else:
                        assert np.array_equal(got, want), (dtype, ddepth, cn, len(a), len(b), border)
