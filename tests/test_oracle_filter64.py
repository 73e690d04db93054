"""oracle/filter64.c (cv::filter2D / cv::sepFilter2D / Sobel / Scharr into CV_64F: double kernels, double rows, every `s += k * v` fused as the reference's AVX2 + FMA
copy of filter.simd.hpp compiles it) against cv::filter2D etc. themselves (oracle/_ref): bit for bit, 1-5 channels (Imgproc_FilterSupportedFormats runs 5)."""
import numpy as np
import pytest

import orc

needs_ref = pytest.mark.skipif(orc.load_ref() is None, reason="oracle/_ref is not built")
SRC = [np.uint8, np.uint16, np.int16, np.float32, np.float64]


def source(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    if dtype in (np.float32, np.float64):
        return ((rng.random(shape) - 0.3) * 100).astype(dtype)
    info = np.iinfo(dtype)
    return rng.integers(max(info.min, -3000), min(int(info.max), 3000) + 1, shape).astype(dtype)


def bits(a):
    return a.view(np.uint64)


@needs_ref
@pytest.mark.parametrize("dtype", [d for d in SRC if d != np.float32])       # (CV_32F -> CV_64F: getLinearFilter has no such engine, filter.simd.hpp:3250)
def test_filter2d_into_64f(dtype):
    rng = np.random.default_rng(4)
    for cn in (1, 3, 5):
        src = source((31, 47, cn) if cn > 1 else (31, 47), dtype, 7 + cn)
        for k in (rng.uniform(-10, 10, (5, 5)).astype(np.float32), rng.uniform(-1, 1, (3, 7)).astype(np.float32), np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], np.float32),
                  rng.uniform(-1, 1, (4, 2))):                                                      # (the last one a CV_64F kernel)
            for border, delta, anchor in ((4, 0.0, (-1, -1)), (0, 0.5, (-1, -1)), (1, -3.25, (0, 1)), (2, 0.0, (-1, -1))):
                got = orc.orc_filter2D(src, 6, k, anchor, delta, border)
                want = orc.ref_filter2D(src, 6, k, anchor, delta, border)
                assert np.array_equal(bits(got), bits(want)), (dtype, cn, k.shape, border, float(np.abs(got - want).max()))


@needs_ref
@pytest.mark.parametrize("dtype", SRC)
def test_sepfilter_and_derivatives_into_64f(dtype):
    rng = np.random.default_rng(5)
    kx, ky = rng.uniform(-1, 1, 11).astype(np.float32), rng.uniform(-1, 1, 7).astype(np.float32)
    sx, sy = kx + kx[::-1], ky + ky[::-1]
    ax_, ay_ = kx - kx[::-1], ky - ky[::-1]
    for cn in (1, 3, 5):
        src = source((29, 43, cn) if cn > 1 else (29, 43), dtype, 17 + cn)
        for (a, b) in ((kx, ky), (sx, sy), (ax_, ay_), (sx, ay_), (kx[:4], ky[:2])):
            for border, delta in ((4, 0.0), (0, 1.5), (1, 0.0), (2, -0.75)):
                got = orc.orc_sepFilter2D(src, 6, a, b, (-1, -1), delta, border)
                want = orc.ref_sepFilter2D(src, 6, a, b, (-1, -1), delta, border)
                assert np.array_equal(bits(got), bits(want)), (dtype, cn, len(a), len(b), border, float(np.abs(got - want).max()))
        for (dx, dy, ks, scale) in ((1, 0, 3, 1.0), (0, 1, 3, 1.0), (2, 0, 5, 1.0), (1, 1, 5, 0.37), (0, 1, -1, 1.0), (1, 0, -1, 2.5), (0, 2, 7, 1.0)):
            got = orc.orc_Sobel(src, 6, dx, dy, ks, scale, 0.25, 4)
            want = orc.ref_Sobel(src, 6, dx, dy, ks, scale, 0.25, 4)
            assert np.array_equal(bits(got), bits(want)), (dtype, cn, dx, dy, ks, scale, float(np.abs(got - want).max()))
