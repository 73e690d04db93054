"""Pinning of the medianBlur restatement (oracle/median.c) against the real reference: apertures 3, 5 (sort network), 7, 9
(histogram forms), 1/3/4 channels, degenerate one-pixel-wide / -high images, 16-bit and float depths for 3 / 5."""
import numpy as np
import pytest

import orc as O


@pytest.mark.ref
def test_median_matches_reference(ref):
    rng = np.random.default_rng(4)
    for shape in [(23, 40), (17, 29, 3), (12, 33, 4), (1, 9), (7, 1), (2, 2, 3), (64, 64)]:
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        for k in (3, 5, 7, 9):
            assert np.array_equal(O.orc_medianBlur(src, k), O.ref_medianBlur(src, k)), (shape, k)
    for dtype in (np.uint16, np.int16, np.float32):
        src = (rng.random((19, 27)) * 1000 - 300).astype(dtype) if dtype != np.uint16 else rng.integers(0, 65536, (19, 27), dtype=np.uint16)
        for k in (3, 5):
            assert np.array_equal(O.orc_medianBlur(src, k), O.ref_medianBlur(src, k)), (dtype, k)
        src3 = (rng.random((14, 21, 3)) * 2000 - 700).astype(dtype) if dtype != np.uint16 else rng.integers(0, 65536, (14, 21, 3), dtype=np.uint16)
        for k in (3, 5):
            assert np.array_equal(O.orc_medianBlur(src3, k), O.ref_medianBlur(src3, k)), (dtype, k, "3 channels")
    for shape, k in [((40, 52), 11), ((33, 47, 3), 13), ((70, 64, 4), 15), ((90, 100), 21), ((64, 80), 31)]:     # the histogram forms at larger apertures
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        assert np.array_equal(O.orc_medianBlur(src, k), O.ref_medianBlur(src, k)), (shape, k)


def test_median_known_answer():
    src = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.uint8)
    assert O.orc_medianBlur(src, 3).tolist() == [[2, 3, 3], [4, 5, 6], [7, 7, 8]]
