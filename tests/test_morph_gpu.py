"""GPU parity for cv::erode / cv::dilate through cv_hal_morphInit / cv_hal_morph / cv_hal_morphFree: every
structuring element / anchor / depth / border of the oracle's own pinning matrix, the rolling fast path ((W*cn) % 16 == 0, full
3/5/7 rectangles), ROIs with real neighbours, folded iterations; bit-exact."""
import numpy as np
import pytest
import torch

from test_oracle_morph import KERNELS, _src

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def dev(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32, np.float64])
def test_morph_generic(cv, orc, dtype):
    n0 = cv.call_count("morph")
    for shape in [(23, 40), (17, 29, 3), (1, 9), (6, 1, 4)]:
        src = _src(dtype, shape, 5 + len(shape))
        for op, fn in ((0, cv.erode), (1, cv.dilate)):
            for k, anchor in KERNELS:
                for border, bv in [(0, None), (0, 7.0), (1, None), (2, None), (4, None)]:
                    want = orc.orc_morph(op, src, k, anchor, border, bv)
                    got = fn(dev(src), k, anchor, 1, border, bv).cpu().numpy()
                    assert np.array_equal(got, want), (dtype, shape, op, None if k is None else k.shape, anchor, border, bv)
    assert cv.call_count("morph") > n0


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_morph_rolling_path(cv, orc, cn):
    rng = np.random.default_rng(cn)
    for (w, h) in [(16, 1), (32, 2), (48, 5), (64, 23), (1040, 37), (2064, 70)]:
        if w * cn % 16:
            continue
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        for K in (3, 5, 7):
            k = np.ones((K, K), np.uint8)
            for op, fn in ((0, cv.erode), (1, cv.dilate)):
                for border in (0, 1, 2, 4):
                    assert np.array_equal(fn(dev(src), k, (-1, -1), 1, border).cpu().numpy(), orc.orc_morph(op, src, k, (-1, -1), border)), (w, h, K, op, border)
    # extremes stay extremes; iterations fold into a bigger rectangle; host pointers
    full = np.full((20, 64, cn) if cn > 1 else (20, 64), 255, np.uint8)
    assert (cv.erode(dev(full)).cpu().numpy() == 255).all() and (cv.dilate(dev(np.zeros_like(full))).cpu().numpy() == 0).all()
    src = rng.integers(0, 256, (40, 64, cn) if cn > 1 else (40, 64), dtype=np.uint8)
    assert np.array_equal(cv.dilate(dev(src), None, (-1, -1), 2).cpu().numpy(), orc.orc_morph(1, src, np.ones((5, 5), np.uint8)))
    assert np.array_equal(cv.erode(src, None, (-1, -1), 3), orc.orc_morph(0, src, np.ones((7, 7), np.uint8)))


def test_morph_roi_reads_real_neighbours(cv, orc):
    parent = _src(np.uint8, (40, 60, 3), 2)
    for roi in [(5, 4, 30, 20), (0, 0, 16, 16), (57, 38, 3, 2)]:
        for border in (0, 1, 4, 4 | 16):
            for op, fn in ((0, cv.erode), (1, cv.dilate)):
                want = orc.orc_morph(op, parent, np.ones((5, 3), np.uint8), (-1, -1), border, None, roi=roi if not border & 16 else None) if not border & 16 else \
                    orc.orc_morph(op, np.ascontiguousarray(parent[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]]), np.ones((5, 3), np.uint8), (-1, -1), border & ~16)
                got = fn(dev(parent), np.ones((5, 3), np.uint8), (-1, -1), 1, border, None, roi=roi).cpu().numpy()
                assert np.array_equal(got, want), (roi, border, op)


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32])
def test_iterated_irregular_elements_and_in_place(cv, orc, dtype):
    """iterations > 1 of a cross / ellipse / arbitrary element (declined before; morph.dispatch.cpp:455-460) as repeated passes on the GPU, and a call
    whose destination IS its source on the device (allowInplace): results identical to the restatement pinned in tests/test_oracle_morph.py"""
    CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    ELL = np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], np.uint8)
    ODD = np.array([[1, 0, 0, 1, 0], [0, 0, 1, 0, 0], [1, 1, 0, 0, 1]], np.uint8)
    n0 = cv.call_count("morph")
    for shape in [(27, 38, 3), (270, 480)]:
        src = _src(dtype, shape, 12)
        for op, fn in ((0, cv.erode), (1, cv.dilate)):
            for k, anchor in [(CROSS, (-1, -1)), (ELL, (-1, -1)), (ODD, (1, 2))]:
                for it in (2, 3, 4):
                    for border, bv in [(0, None), (0, 7.0), (1, None), (4, None)]:
                        got = fn(dev(src), k, anchor, it, border, bv).cpu().numpy()
                        assert np.array_equal(got, orc.orc_morph(op, src, k, anchor, border, bv, iterations=it)), (dtype, shape, op, k.shape, it, border)
            assert np.array_equal(fn(src, CROSS, (-1, -1), 3), orc.orc_morph(op, src, CROSS, iterations=3))            # host arrays
            for it in (1, 2, 3):                                                                                        # in place on the device
                d = dev(src)
                out = fn(d, ELL, (-1, -1), it, dst=d)
                assert out.data_ptr() == d.data_ptr() and np.array_equal(d.cpu().numpy(), orc.orc_morph(op, src, ELL, iterations=it)), (op, it)
                d = dev(src)
                fn(d, np.ones((3, 3), np.uint8), dst=d)
                assert np.array_equal(d.cpu().numpy(), orc.orc_morph(op, src, np.ones((3, 3), np.uint8)))
    assert cv.call_count("morph") > n0


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
def test_more_than_four_channels(cv, orc, dtype):
    """erode / dilate of 5 / 6 / 9 channel images (Imgproc_FilterSupportedFormats): default and extrapolating borders; a constant border whose Scalar differs
    between its entries is declined (the reference unrolls it over the border elements with period 4)"""
    for shape in [(23, 40, 5), (17, 29, 6), (6, 3, 9)]:
        src = _src(dtype, shape, 11 + shape[2])
        for op, fn in ((0, cv.erode), (1, cv.dilate)):
            for k, anchor in KERNELS[:6]:
                for border in (0, 1, 2, 4):
                    got = fn(dev(src), k, anchor, 1, border).cpu().numpy()
                    assert np.array_equal(got, orc.orc_morph(op, src, k, anchor, border)), (dtype, shape, op, anchor, border)
    with pytest.raises(NotImplementedError):
        cv.erode(dev(_src(dtype, (9, 9, 5), 3)), None, (-1, -1), 1, 0, (1.0, 2.0, 3.0, 4.0))


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_morph_large_rectangles_are_separable(cv, orc, cn):
    """cv::erode / cv::dilate on CV_8U with full rectangles beyond 7 x 7 (9 x 9 ... 129 x 1, odd anchors, folded iterations, custom border values, ROI windows): the
    LDS-ring kernel's minimum / maximum modes (k_seplong<4 / 5>: kw + kh comparisons per element) instead of one thread per output walking the whole element"""
    from opencv_amd import _lib
    rng = np.random.default_rng(90 + cn)
    for (w, h) in [(317, 70), (64, 33), (45, 130)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        for (kw, kh, anchor) in [(9, 9, (-1, -1)), (15, 15, (-1, -1)), (31, 5, (-1, -1)), (3, 25, (1, 20)), (21, 21, (3, 17)), (129, 1, (-1, -1)), (11, 11, (0, 0))]:
            if kw > w or kh > h:
                continue
            k = np.ones((kh, kw), np.uint8)
            for op, fn in ((0, cv.erode), (1, cv.dilate)):
                for border, bv in [(0, None), (0, 7.0), (1, None), (2, None), (4, None)]:
                    got = fn(dev(src), k, anchor, 1, border, bv).cpu().numpy()
                    kn = _lib.lib.mi355cv_lastKernel().decode()
                    assert ("k_seplong<%d," % (4 + op)) in kn, kn
                    assert np.array_equal(got, orc.orc_morph(op, src, k, anchor, border, bv)), (w, h, cn, kw, kh, anchor, op, border, bv)
    src = rng.integers(0, 256, (60, 200, cn) if cn > 1 else (60, 200), dtype=np.uint8)
    assert np.array_equal(cv.dilate(dev(src), None, (-1, -1), 6).cpu().numpy(), orc.orc_morph(1, src, np.ones((13, 13), np.uint8)))       # six 3 x 3 iterations fold into 13 x 13
    parent = rng.integers(0, 256, (80, 300, cn) if cn > 1 else (80, 300), dtype=np.uint8)
    for roi in [(5, 4, 200, 60), (0, 0, 128, 80), (290, 10, 10, 60)]:
        for op, fn in ((0, cv.erode), (1, cv.dilate)):
            got = fn(dev(parent), np.ones((11, 13), np.uint8), (-1, -1), 1, 1, None, roi=roi).cpu().numpy()
            assert np.array_equal(got, orc.orc_morph(op, parent, np.ones((11, 13), np.uint8), (-1, -1), 1, None, roi=roi)), (roi, op)


def _ellipse(kh, kw):
    """an elliptic mask like cv::getStructuringElement(MORPH_ELLIPSE) (any mask does: the element arrives as its non-zero taps)"""
    yy, xx = np.mgrid[0:kh, 0:kw]
    cy, cx = (kh - 1) / 2.0, (kw - 1) / 2.0
    return ((((yy - cy) / max(cy, 0.5)) ** 2 + ((xx - cx) / max(cx, 0.5)) ** 2) <= 1.0).astype(np.uint8)


def _last_kernel():
    from opencv_amd import _lib
    return _lib.lib.mi355cv_lastKernel().decode()


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
def test_irregular_elements_on_the_lds_tile(cv, orc, dtype):
    """crosses, ellipses and random masks up to 32 wide, every depth, 1-4 channels, ROI windows, custom border values, widths across the 64-column tile edge: k_morph_tile
    (the source box staged once, a sliding window per element row) against the restatement, bit for bit"""
    rng = np.random.default_rng(17)
    cross = np.zeros((9, 9), np.uint8); cross[4, :] = 1; cross[:, 4] = 1
    rnd31 = (rng.random((17, 31)) < 0.3).astype(np.uint8); rnd31[8, 15] = 1
    elements = [(_ellipse(5, 5), (-1, -1)), (_ellipse(15, 15), (-1, -1)), (_ellipse(7, 21), (3, 2)), (cross, (-1, -1)), (rnd31, (30, 0)), (_ellipse(32, 32), (-1, -1)),
                (np.ones((3, 3), np.uint8), (-1, -1))]
    for shape in [(37, 130), (20, 65, 3), (33, 64, 4), (5, 3)]:
        src = _src(dtype, shape, 11 + len(shape))
        for k, anchor in elements:
            if dtype == np.uint8 and k.all():
                continue                                      # (full rectangles on CV_8U have kernels of their own)
            for op, fn in ((0, cv.erode), (1, cv.dilate)):
                for border, bv in [(0, None), (0, 9.0), (1, None), (4, None)]:
                    want = orc.orc_morph(op, src, k, anchor, border, bv)
                    got = fn(dev(src), k, anchor, 1, border, bv).cpu().numpy()
                    assert "k_morph_tile" in _last_kernel(), _last_kernel()
                    assert np.array_equal(got, want), (dtype, shape, op, k.shape, anchor, border, bv)
    parent = _src(dtype, (40, 90, 3), 3)
    k = _ellipse(9, 13)
    for roi in [(7, 6, 70, 20), (0, 0, 64, 16), (86, 37, 4, 3)]:
        for border in (1, 4, 4 | 16):
            for op, fn in ((0, cv.erode), (1, cv.dilate)):
                if border & 16:                               # BORDER_ISOLATED: the window is the whole image
                    want = orc.orc_morph(op, np.ascontiguousarray(parent[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]]), k, (-1, -1), border & ~16)
                else:
                    want = orc.orc_morph(op, parent, k, (-1, -1), border, None, roi=roi)
                got = fn(dev(parent), k, (-1, -1), 1, border, None, roi=roi).cpu().numpy()
                assert "k_morph_tile" in _last_kernel(), _last_kernel()
                assert np.array_equal(got, want), (dtype, roi, border, op)
    # iterated irregular element: as many passes
    src = _src(dtype, (30, 70), 4)
    want = orc.orc_morph(0, orc.orc_morph(0, src, cross), cross)
    assert np.array_equal(cv.erode(dev(src), cross, (-1, -1), 2).cpu().numpy(), want)
