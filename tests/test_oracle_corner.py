"""Pins oracle/corner.c against the real reference (CPU only): cornerHarris / cornerMinEigenVal with the global
relative norm the reference's own accelerated-backend test uses (test/ocl/test_imgproc.cpp:246-263), pyrDown bit-exact
for integers, goodFeaturesToTrack corner lists."""
import numpy as np
import pytest


def smooth_image(orc, h, w, seed):
    """corners need structure: blurred noise + a few rectangles (uniform noise has no stable maxima)"""
    src = orc.ref_rng_fill((h, w), np.uint8, seed, 0, 256)
    img = orc.ref_GaussianBlur(src, 9, 0, 0, 4)
    img = img.copy()
    img[h // 4:h // 2, w // 3:w // 3 + w // 5] = 220
    img[h // 2 + 5:h // 2 + 25, w // 8:w // 8 + 30] = 30
    return img


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_corner_response(orc, ref, dtype):
    for (w, h) in [(64, 48), (133, 77)]:
        img = smooth_image(orc, h, w, 5 + w)
        src = img if dtype == np.uint8 else (img.astype(np.float32) / 255.0)
        for bs, ks in [(2, 3), (3, 3), (5, 5), (2, 7), (4, -1), (1, 3)]:
            for border in (1, 2, 4):
                want = orc.ref_cornerHarris(src, bs, ks, 0.04, border)
                got = orc.orc_cornerHarris(src, bs, ks, 0.04, border)
                assert orc.rel_err(got, want) <= 1e-5, (bs, ks, border)
                if bs == 1:
                    continue      # rank-1 structure tensor: the smaller eigenvalue is pure rounding noise
                want = orc.ref_cornerMinEigenVal(src, bs, ks, border)
                got = orc.orc_cornerMinEigenVal(src, bs, ks, border)
                assert orc.rel_err(got, want) <= 1e-5, (bs, ks, border)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_pyrdown(orc, ref, dtype, cn):
    hi = {np.uint8: 256, np.uint16: 65536, np.int16: 32767, np.float32: 1.0}[dtype]
    for (w, h) in [(64, 48), (65, 49), (31, 7), (2, 2), (5, 1), (1, 9)]:
        src = orc.ref_rng_fill((h, w, cn) if cn > 1 else (h, w), dtype, 3 + w, -hi if dtype == np.int16 else 0, hi)
        for border in (1, 2, 3, 4):
            want = orc.ref_pyrDown(src, None, border)
            got = orc.orc_pyrDown(src, None, border)
            if dtype == np.float32:
                assert orc.rel_err(got, want) <= 1e-6
            else:
                assert np.array_equal(got, want), (w, h, border)
    src = orc.ref_rng_fill((48, 64, cn) if cn > 1 else (48, 64), dtype, 99, 0, hi)
    for dsize in [(33, 25), (31, 23)]:              # |dsize*2 - ssize| <= 2 is allowed (pyramids.cpp:899-901)
        want = orc.ref_pyrDown(src, dsize, 4)
        got = orc.orc_pyrDown(src, dsize, 4)
        assert (orc.rel_err(got, want) <= 1e-6) if dtype == np.float32 else np.array_equal(got, want)


def test_good_features_to_track(orc, ref):
    for (w, h) in [(160, 120), (97, 143)]:
        img = smooth_image(orc, h, w, 11 + w)
        for harris in (False, True):
            for maxc, q, md in [(50, 0.01, 5.0), (0, 0.05, 0.0), (25, 0.02, 12.3), (1000, 0.001, 1.0)]:
                want = orc.ref_goodFeaturesToTrack(img, maxc, q, md, 3, 3, harris, 0.04)
                got = orc.orc_goodFeaturesToTrack(img, maxc, q, md, None, 3, 3, harris, 0.04)
                assert len(want) > 0
                assert got.shape == want.shape and np.array_equal(got, want), (w, h, harris, maxc, q, md)
