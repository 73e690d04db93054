"""Pinning of the cv::threshold restatement (oracle/thresh.c) against the real reference (oracle/_ref): every depth on the path,
every fixed-level type, thresholds inside, at and beyond the range (the degenerate shortcuts of thresh.cpp:1595-1609)."""
import numpy as np
import pytest

import orc as O

CASES = [(np.uint8, [-3.0, 0.0, 0.5, 17.9, 127.0, 254.0, 254.9, 255.0, 300.0], [200.0, 255.4, 300.0, -5.0]),
         (np.uint16, [-1.0, 0.0, 1000.3, 65534.0, 65535.0, 70000.0], [60000.0, 65535.9, 1e6]),
         (np.int16, [-40000.0, -32768.0, -5.5, 0.0, 12345.6, 32766.0, 32767.0, 40000.0], [30000.0, -123.0, 1e6]),
         (np.float32, [-1.5, 0.0, 0.25, 0.999, 7.0], [1.0, -2.5, 255.0]),
         (np.float64, [-1.5, 0.0, 0.25, 0.999, 7.0], [1.0, -2.5, 255.0])]


def _src(dtype, seed, shape=(37, 61, 3)):
    rng = np.random.default_rng(seed)
    if dtype == np.float32:
        return (rng.random(shape, dtype=np.float32) * 2 - 0.5).astype(np.float32)
    if dtype == np.float64:
        return rng.random(shape) * 2 - 0.5
    info = np.iinfo(dtype)
    a = rng.integers(info.min, int(info.max) + 1, shape, dtype=dtype)
    a.flat[:4] = [info.min, info.max, info.max - 1, info.min + 1]
    return a


@pytest.mark.ref
@pytest.mark.parametrize("dtype,threshes,maxvals", CASES)
def test_threshold_matches_reference(ref, dtype, threshes, maxvals):
    src = _src(dtype, 11)
    for t in threshes:
        for m in maxvals:
            for ttype in range(5):
                rv_r, want = O.ref_threshold(src, t, m, ttype)
                rv_o, got = O.orc_threshold(src, t, m, ttype)
                assert rv_r == rv_o, (dtype, t, m, ttype)
                assert np.array_equal(got, want), (dtype, t, m, ttype)


def test_threshold_known_answers():
    """hand-checkable vector (the rule of thresh_8u, thresh.cpp:112-): src > thresh"""
    src = np.array([[0, 9, 10, 11, 255]], np.uint8)
    exp = {0: [0, 0, 0, 200, 200], 1: [200, 200, 200, 0, 0], 2: [0, 9, 10, 10, 10], 3: [0, 0, 0, 11, 255], 4: [0, 9, 10, 0, 0]}
    for ttype, e in exp.items():
        rv, got = O.orc_threshold(src, 10.7, 200.2, ttype)
        assert rv == 10.0 and got.tolist() == [e], ttype


@pytest.mark.ref
def test_adaptive_threshold_mean_matches_reference(ref):
    rng = np.random.default_rng(21)
    for shape in [(37, 61), (64, 64), (5, 9), (1, 20)]:
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        for bs in (3, 5, 7, 11, 15):
            for ttype in (0, 1):
                for C in (0.0, 2.0, -3.5, 7.25):
                    for mv in (255.0, 100.4):
                        want = O.ref_adaptiveThreshold(src, mv, 0, ttype, bs, C)
                        got = O.orc_adaptiveThreshold(src, mv, ttype, bs, C)
                        assert np.array_equal(got, want), (shape, bs, ttype, C, mv)


@pytest.mark.ref
def test_adaptive_threshold_gaussian_and_large_blocks_match_reference(ref):
    """ADAPTIVE_THRESH_GAUSSIAN_C (float blur of the float image, back to 8 bits, thresh.cpp:1720-1727) and MEAN_C beyond 15 x 15 (int32 box
    sums with the reference's float body / double tail) against the real reference"""
    rng = np.random.default_rng(33)
    for shape in [(37, 61), (64, 64), (5, 9), (1, 20), (80, 141)]:
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        smooth = np.clip(np.add.outer(np.arange(shape[0]) * 3, np.arange(shape[1]) * 2) % 256 + rng.integers(-2, 3, shape), 0, 255).astype(np.uint8)
        for img in (src, smooth):
            for bs in (3, 5, 7, 9, 11, 15, 21, 33):
                for ttype in (0, 1):
                    for C in (0.0, 2.0, -3.5):
                        want = O.ref_adaptiveThreshold(img, 255.0, 1, ttype, bs, C)
                        got = O.orc_adaptiveThreshold(img, 255.0, ttype, bs, C, method=1)
                        assert np.array_equal(got, want), ("gaussian", shape, bs, ttype, C)
            for bs in (17, 21, 31, 51):
                want = O.ref_adaptiveThreshold(img, 200.0, 0, 0, bs, 1.5)
                got = O.orc_adaptiveThreshold(img, 200.0, 0, bs, 1.5)
                assert np.array_equal(got, want), ("mean", shape, bs)
            if shape in [(37, 61), (80, 141)]:
                # the block sizes the GPU path serves since rounds 5 / 6 (35 .. 129 Gaussian, up to 255 mean): tests/test_thresh_gpu.py asserts parity at exactly these
                for bs in (35, 65, 101, 129):
                    for ttype, C in ((0, 0.0), (1, -3.5)):
                        assert np.array_equal(O.orc_adaptiveThreshold(img, 255.0, ttype, bs, C, method=1), O.ref_adaptiveThreshold(img, 255.0, 1, ttype, bs, C)), ("gaussian", shape, bs, ttype)
                for bs in (101, 255):
                    assert np.array_equal(O.orc_adaptiveThreshold(img, 200.0, 0, bs, 1.5), O.ref_adaptiveThreshold(img, 200.0, 0, 0, bs, 1.5)), ("mean", shape, bs)


@pytest.mark.ref
def test_bilateral_filter_matches_reference(ref):
    """cv::bilateralFilter 8UC1 / 8UC3: the three summation forms of the AVX2 build (vector body, 4-group 128-bit tail, scalar rest), the division /
    reciprocal split between one and three channels, every border -- bit for bit against the real reference"""
    rng = np.random.default_rng(3)
    for cn in (1, 3):
        for (w, h) in [(97, 33), (64, 20), (41, 17), (130, 9)]:
            shape = (h, w, cn) if cn > 1 else (h, w)
            src = rng.integers(0, 256, shape, dtype=np.uint8)
            sm = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256).astype(np.uint8)
            smooth = np.repeat(sm[..., None], cn, -1).copy() if cn > 1 else sm
            noisy = np.clip(smooth.astype(int) + rng.integers(-6, 7, shape), 0, 255).astype(np.uint8)
            for img in (src, noisy):
                for d, sc, ss in [(5, 25.0, 3.0), (9, 75.0, 75.0), (0, 30.0, 2.0), (3, 10.0, 1.0), (15, 40.0, 4.0)]:
                    for border in (4, 1, 0, 2):
                        assert np.array_equal(O.orc_bilateralFilter(img, d, sc, ss, border), O.ref_bilateralFilter(img, d, sc, ss, border)), (cn, w, h, d, sc, ss, border)


@pytest.mark.ref
def test_image_moments_match_reference(ref):
    """cv::moments (spatial moments m00 .. m03) for CV_8U / CV_16U / CV_16S, plain and binary: exact integer tile moments, the reference's double accumulation
    over the 32 x 32 tiles in its order and grouping -- equal as doubles, a 4K frame included"""
    rng = np.random.default_rng(4)
    for dt in (np.uint8, np.uint16, np.int16):
        info = np.iinfo(dt)
        for (w, h) in [(1, 1), (31, 5), (32, 32), (33, 65), (200, 97), (3840, 2160), (1000, 37)]:
            for src in (rng.integers(info.min, int(info.max) + 1, (h, w), dtype=dt), np.full((h, w), info.max, dtype=dt)):
                for binary in (False, True):
                    assert np.array_equal(O.orc_moments(src, binary), O.ref_moments(src, binary)), (dt, w, h, binary)


def test_image_moments_float_match_reference(ref):
    """cv::moments of CV_32F / CV_64F images: chains of double additions in raster order inside every 32 x 32 tile (no vector form in the reference) --
    the restatement (oracle/moments.c orc_imageMomentsF) gives the reference's doubles bit for bit; the GPU kernel for these depths is not built yet
    (the hook declines them), this pins what it will have to reproduce"""
    rng = np.random.default_rng(2)
    for dt in (np.float32, np.float64):
        for (h, w) in [(1, 1), (31, 33), (64, 64), (100, 257), (480, 641), (1080, 1920)]:
            img = (rng.random((h, w)) * 1000 - 300).astype(dt)
            img[rng.random((h, w)) < 0.2] = 0
            for binary in (False, True):
                a, b = O.orc_moments(img, binary), O.ref_moments(img, binary)
                assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (dt, h, w, binary)


@pytest.mark.ref
def test_gaussian_c_float_blur_is_bit_identical_on_8bit_valued_images(ref):
    """the float blur inside ADAPTIVE_THRESH_GAUSSIAN_C (CV_32F GaussianBlur of an 8-bit valued image): the restated separable float path gives the
    reference's floats bit for bit -- so the rounded mean, and with it the thresholded image, cannot differ by a tie"""
    rng = np.random.default_rng(7)
    src = rng.integers(0, 256, (257, 333), dtype=np.uint8).astype(np.float32)
    for bs in (3, 5, 7, 11, 21, 33):
        k = O.ref_getGaussianKernel(bs, 0.0).astype(np.float32).ravel()
        a = O.orc_sepFilter2D(src, 5, k, k, border=1)
        b = O.ref_GaussianBlur(src, bs, 0.0, 0.0, 1 | 16)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), bs


@pytest.mark.ref
def test_bilateral_filter_32f_matches_reference(ref):
    """cv::bilateralFilter CV_32FC1 / CV_32FC3 (bilateralFilter_32f): colour weights from the interpolated table over the image's value range, the centre with weight 1,
    NaN pixels skipped, a constant image copied -- the restatement's scalar form against the reference's vector bodies: 1e-6 (north_star: 1e-4)"""
    rng = np.random.default_rng(3)
    for shape in [(40, 53), (33, 47, 3), (9, 8), (5, 70, 3)]:
        src = (rng.random(shape, dtype=np.float32) * 3 - 1).astype(np.float32)
        for d, sc, ss, border in [(5, 0.3, 2.0, 4), (0, 1.5, 1.2, 1), (9, 0.05, 3.0, 2), (3, 10.0, 1.0, 0), (15, 0.5, 4.0, 4)]:
            want = O.ref_bilateralFilter(src, d, sc, ss, border)
            got = O.orc_bilateralFilter(src, d, sc, ss, border)
            assert O.rel_err(got, want) <= 1e-6 and np.abs(got - want).max() <= 2e-6, (shape, d, sc, ss, border, O.rel_err(got, want))
    src = rng.random((30, 40), dtype=np.float32); src[5, 7] = np.nan; src[10:12, 20] = np.nan
    want = O.ref_bilateralFilter(src, 5, 0.3, 2.0, 4); got = O.orc_bilateralFilter(src, 5, 0.3, 2.0, 4)
    assert not np.isnan(want).any() and not np.isnan(got).any() and np.abs(got - want).max() <= 2e-6
    flat = np.full((20, 30, 3), 0.37, np.float32)
    assert np.array_equal(O.orc_bilateralFilter(flat, 5, 0.3, 2.0, 4), O.ref_bilateralFilter(flat, 5, 0.3, 2.0, 4))
