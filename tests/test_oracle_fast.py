"""Pins oracle/fast.c (FAST corner detector in the form the features2d HAL consumes: dense score, 3x3 suppression, raster-order keypoints)
against cv::FAST of the real reference (CPU only): positions, order and responses equal for TYPE_9_16 (the type the HAL serves), every threshold and both
suppression settings, on noise and on structured images."""
import numpy as np
import pytest


def images(orc):
    noise = orc.ref_rng_fill((97, 131), np.uint8, 11, 0, 256)
    smooth = orc.ref_GaussianBlur(orc.ref_rng_fill((120, 160), np.uint8, 12, 0, 256), 5, 0, 0, 4)
    blocks = np.kron(orc.ref_rng_fill((12, 16), np.uint8, 13, 0, 256), np.ones((9, 9), np.uint8)).astype(np.uint8)
    tiny = orc.ref_rng_fill((7, 7), np.uint8, 14, 0, 256)
    return [noise, smooth, blocks, np.ascontiguousarray(blocks[:, :-3]), tiny, orc.ref_rng_fill((6, 40), np.uint8, 15, 0, 256)]


@pytest.mark.parametrize("nonmax", [True, False])
def test_fast_equals_the_reference(orc, ref, nonmax, ftype=2):
    for img in images(orc):
        for thr in (0, 1, 5, 10, 20, 40, 100):
            want = orc.ref_FAST(img, thr, nonmax, ftype)
            got = orc.orc_FAST(img, thr, nonmax, ftype)
            assert got.shape == want.shape and np.array_equal(got, want), (img.shape, thr, len(got), len(want))
