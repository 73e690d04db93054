"""Pinning of the erode / dilate restatement (oracle/morph.c) against the real reference: rectangular, cross and arbitrary
structuring elements, off-centre anchors, every depth on the path, default and explicit constant borders, extrapolating borders,
folded iterations of a rectangular element."""
import numpy as np
import pytest

import orc as O

CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
ODD = np.array([[1, 0, 0, 1, 0], [0, 0, 1, 0, 0], [1, 1, 0, 0, 1]], np.uint8)
KERNELS = [(None, (-1, -1)), (np.ones((3, 3), np.uint8), (-1, -1)), (np.ones((5, 5), np.uint8), (-1, -1)), (np.ones((2, 4), np.uint8), (3, 0)),
           (CROSS, (-1, -1)), (ODD, (1, 2)), (np.ones((1, 7), np.uint8), (-1, -1)), (np.ones((7, 1), np.uint8), (0, 5))]


def _src(dtype, shape, seed):
    rng = np.random.default_rng(seed)
    if dtype in (np.float32, np.float64):
        return (rng.random(shape) * 4 - 2).astype(dtype)
    info = np.iinfo(dtype)
    return rng.integers(info.min, int(info.max) + 1, shape, dtype=dtype)


@pytest.mark.ref
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32, np.float64])
def test_morph_matches_reference(ref, dtype):
    for shape in [(23, 40), (17, 29, 3), (1, 9), (6, 1, 4)]:
        src = _src(dtype, shape, 5 + len(shape))
        for op in (0, 1):
            for k, anchor in KERNELS:
                for border, bv in [(0, None), (0, 7.0), (1, None), (2, None), (4, None)]:
                    want = O.ref_morph(op, src, k, anchor, 1, border, bv)
                    got = O.orc_morph(op, src, k, anchor, border, bv)
                    assert np.array_equal(got, want), (dtype, shape, op, None if k is None else k.shape, anchor, border, bv)


@pytest.mark.ref
def test_folded_iterations_equal_bigger_rectangle(ref):
    """morphOp folds `iterations` of a rectangular element into one (:963-972): what reaches cv_hal_morph is iterations == 1"""
    src = _src(np.uint8, (31, 45, 3), 9)
    for op in (0, 1):
        want = O.ref_morph(op, src, None, (-1, -1), 3, 0, None)                  # empty kernel, 3 iterations -> 7x7
        assert np.array_equal(O.orc_morph(op, src, np.ones((7, 7), np.uint8)), want)
        want = O.ref_morph(op, src, np.ones((3, 5), np.uint8), (-1, -1), 2, 4, None)
        assert np.array_equal(O.orc_morph(op, src, np.ones((5, 9), np.uint8), (-1, -1), 4), want)


@pytest.mark.ref
@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32])
def test_iterated_irregular_elements(ref, dtype):
    """iterations > 1 of a cross / an arbitrary element reach cv_hal_morph as they are (only full rectangles are folded): repeated whole-image passes"""
    src = _src(dtype, (27, 38, 3), 12)
    ELL = np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], np.uint8)      # getStructuringElement(MORPH_ELLIPSE, (5, 5))
    for op in (0, 1):
        for k, anchor in [(CROSS, (-1, -1)), (ELL, (-1, -1)), (ODD, (1, 2))]:
            for it in (2, 3, 4):
                for border, bv in [(0, None), (0, 7.0), (1, None), (4, None)]:
                    assert np.array_equal(O.orc_morph(op, src, k, anchor, border, bv, iterations=it), O.ref_morph(op, src, k, anchor, it, border, bv)), (dtype, op, k.shape, it, border)


def test_morph_known_answer():
    src = np.array([[5, 1, 9], [7, 3, 2], [4, 8, 6]], np.uint8)
    assert O.orc_morph(1, src).tolist() == [[7, 9, 9], [8, 9, 9], [8, 8, 8]]          # dilate: outside ignored
    assert O.orc_morph(0, src).tolist() == [[1, 1, 1], [1, 1, 1], [3, 2, 2]]          # erode
    assert O.orc_morph(0, src, None, (-1, -1), 0, 0.0)[0].tolist() == [0, 0, 0]        # explicit constant 0 border


@pytest.mark.ref
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
def test_more_than_four_channels(ref, dtype):
    """5 / 6 / 9 channels (the reference's Imgproc_FilterSupportedFormats runs morphologyEx on 5): the default constant border and the extrapolating ones.  (A border
    Scalar that differs between its four entries is unrolled over the border ELEMENTS with period 4 there -- not a per-channel value -- and is outside the restatement.)"""
    for shape in [(23, 40, 5), (17, 29, 6), (6, 3, 9)]:
        src = _src(dtype, shape, 11 + shape[2])
        for op in (0, 1):
            for k, anchor in KERNELS[:6]:
                for border in (0, 1, 2, 4):
                    assert np.array_equal(O.orc_morph(op, src, k, anchor, border), O.ref_morph(op, src, k, anchor, 1, border)), (dtype, shape, op, anchor, border)
    o = O.oracle()
    src = _src(np.uint8, (5, 5, 5), 1); dst = np.empty_like(src); k = np.ones((3, 3), np.uint8)
    assert o.orc_morph(0, O.P(src), O.step(src), O.P(dst), O.step(dst), 5, 5, 0, 5, 5, 5, 0, 0, O.P(k), O.c_sz(3), 3, 3, 1, 1, 0, O._bvp((1.0, 2.0, 3.0, 4.0))) == 2
