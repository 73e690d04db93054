"""GPU parity for rows a3-a6: filter2D, sepFilter2D, Sobel/Scharr, boxFilter, cvtColor -- through the C ABI,
against the oracle.  Integer outputs bit-exact, CV_32F within 1e-4 relative (ts/ocl_test.hpp:309 norm)."""
import os
import zlib

import numpy as np
import pytest
import torch

from test_oracle_filter import SHARPEN, BORDERS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def dev(a):
    return torch.from_numpy(a).cuda()


def last_kernel():
    from opencv_amd import _lib
    return _lib.lib.mi355cv_lastKernel().decode()


def rnd(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    if dtype == np.float32:
        return rng.random(shape, dtype=np.float32)
    info = np.iinfo(dtype)
    return rng.integers(info.min, int(info.max) + 1, shape, dtype=dtype)


def check(got, want, tol=1e-4):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.dtype == want.dtype and got.shape == want.shape
    if want.dtype == np.float32:
        import orc
        assert orc.rel_err(got, want) <= tol
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_filter2d_8u(cv, orc, cn):
    n0 = cv.call_count("filter")
    for (w, h) in [(64, 9), (100, 33), (1, 1), (3, 2), (257, 19)]:
        src = rnd((h, w, cn) if cn > 1 else (h, w), np.uint8, w)
        for k, anchor, delta in [(SHARPEN, (-1, -1), 0.0), (np.ones((3, 3), np.float32) * 0.125, (-1, -1), 3.0),
                                 (np.array([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], np.float32) * 0.25, (0, 2), 128.0),
                                 (np.array([[0.5, 0.25, 0.125, 0.0625, 0.0625]], np.float32), (1, 0), 0.0),
                                 (np.zeros((3, 3), np.float32), (-1, -1), 7.0)]:
            for border in BORDERS + [3]:
                want = orc.orc_filter2D(src, -1, k, anchor, delta, border)
                check(cv.filter2D(dev(src), -1, k, anchor, delta, border), want)
        check(cv.filter2D(src, -1, SHARPEN), orc.orc_filter2D(src, -1, SHARPEN))            # host pointers
    assert cv.call_count("filter") > n0


@pytest.mark.parametrize("cn,K", [(1, 3), (3, 3), (4, 3), (1, 5)])
def test_filter2d_rolling_path(cv, orc, cn, K):
    """Geometries the register-rolling kernel takes ((W*cn) % 16 == 0): several strips, short/odd heights so that
    segments walk both up and down, asymmetric float taps (catch a flipped row order), every border mode."""
    rng = np.random.default_rng(100 * K + cn)
    kern = (rng.random((K, K), dtype=np.float32) - 0.4).astype(np.float32) / K
    for (w, h) in [(16, 1), (32, 2), (48, 5), (1040, 37), (2064, 70), (16 * 67, 131)]:
        if w * cn % 16:
            continue
        src = rnd((h, w, cn) if cn > 1 else (h, w), np.uint8, w + h)
        for border in BORDERS:
            for k, delta in [(kern, 0.0), (kern * 3, 10.5)] + ([(SHARPEN, 0.0)] if K == 3 else []):
                want = orc.orc_filter2D(src, -1, k, (-1, -1), delta, border)
                check(cv.filter2D(dev(src), -1, k, (-1, -1), delta, border), want)
    # batch: frames are isolated images
    n = 5
    fr = rnd((n, 40, 64, cn) if cn > 1 else (n, 40, 64), np.uint8, 9)
    got = cv.filter2DBatch(dev(fr), -1, kern, (-1, -1), 2.0, 4).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], orc.orc_filter2D(fr[i], -1, kern, (-1, -1), 2.0, 4))
    # batch on a geometry the rolling kernel declines (falls to the generic kernel per frame)
    fr = rnd((3, 11, 50), np.uint8, 10)
    got = cv.filter2DBatch(dev(fr), -1, kern, (-1, -1), 0.0, 1).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], orc.orc_filter2D(fr[i], -1, kern, (-1, -1), 0.0, 1))


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_ragged_and_unaligned_rows_on_the_rolling_kernels(cv, orc, cn):
    """Row lengths that are not a multiple of 16 bytes, on tightly packed (so mostly unaligned) rows and on padded / offset
    views: the rolling kernels rebuild the last chunk from the row's last 16 bytes and store only its valid bytes."""
    rng = np.random.default_rng(40 + cn)
    kern = (rng.random((3, 3), dtype=np.float32) - 0.4).astype(np.float32) / 3
    for (w, h) in [(17, 3), (23, 9), (100, 33), (1000, 21), (1027, 18), (2049, 7), (65 * 16 // cn + 1, 11)]:
        if w * cn < 16:
            continue
        base = rng.integers(0, 256, (h, w + 5, cn) if cn > 1 else (h, w + 5), dtype=np.uint8)
        views = [np.ascontiguousarray(base[:, :w]), base[:, 3:3 + w]]                    # packed rows; offset view with a pitch of w+5 pixels
        for src in views:
            d = torch.from_numpy(np.ascontiguousarray(base)).cuda()[:, 3:3 + w] if src is views[1] else dev(src)
            srcc = np.ascontiguousarray(src)
            for border in (0, 1, 2, 4):
                check(cv.filter2D(d, -1, kern, (-1, -1), 1.5, border | 16), orc.orc_filter2D(srcc, -1, kern, (-1, -1), 1.5, border))
                check(cv.boxFilter(d, -1, (5, 5), (-1, -1), True, border | 16), orc.orc_boxFilter(srcc, -1, (5, 5), (-1, -1), True, border))
                check(cv.GaussianBlur(d, (5, 5), 1.2, 1.2, border | 16), orc.orc_sepSmoothFixedU8(srcc, [int(v) for v in orc.orc_getGaussianKernelQ(5, 1.2)],
                                                                                              [int(v) for v in orc.orc_getGaussianKernelQ(5, 1.2)], border))
                check(cv.GaussianBlur(d, (5, 5), 0, 0, border | 16), orc.orc_gaussianBlurBinomialU8(srcc, 5, border))
                check(cv.dilate(d, np.ones((3, 3), np.uint8), (-1, -1), 1, border | 16), orc.orc_morph(1, srcc, np.ones((3, 3), np.uint8), (-1, -1), border))
                check(cv.erode(d, np.ones((5, 5), np.uint8), (-1, -1), 1, border | 16), orc.orc_morph(0, srcc, np.ones((5, 5), np.uint8), (-1, -1), border))
                if cn == 1:
                    check(cv.Sobel(d, cv.CV_16S, 1, 0, 3, 1.0, 0.0, border | 16), orc.orc_Sobel(srcc, 3, 1, 0, 3, 1.0, 0.0, border))
                    check(cv.Sobel(d, cv.CV_32F, 0, 1, 3, 0.5, 0.0, border | 16), orc.orc_Sobel(srcc, 5, 0, 1, 3, 0.5, 0.0, border))
                    want = orc.orc_cornerHarris(srcc, 2, 3, 0.04, border)
                    got = cv.cornerHarris(d, 2, 3, 0.04, border | 16).cpu().numpy()
                    import orc as O
                    assert O.rel_err(got, want) <= 1e-4, (w, h, border)


def test_filter2d_depths_and_roi(cv, orc):
    rng = np.random.default_rng(5)
    k5 = (rng.uniform(-3, 10, (5, 5)) / 37.0).astype(np.float32)
    for dtype, ddepth in [(np.uint8, 3), (np.uint8, 5), (np.uint16, -1), (np.int16, -1), (np.float32, -1), (np.uint8, -1), (np.uint8, 2)]:
        src = rnd((37, 64, 3), dtype, 77)
        for border in BORDERS:
            check(cv.filter2D(dev(src), ddepth, k5, (-1, -1), 0.5, border), orc.orc_filter2D(src, ddepth, k5, (-1, -1), 0.5, border))
    parent = rnd((40, 60, 3), np.uint8, 1)
    for roi in [(5, 4, 30, 20), (0, 0, 16, 16), (58, 38, 2, 2)]:
        for border in (1, 2, 4, 4 | 16):
            want = orc.orc_filter2D(parent, -1, k5, (-1, -1), 0.0, border, roi=roi)
            check(cv.filter2D(dev(parent), -1, k5, (-1, -1), 0.0, border, roi=roi), want)
            check(cv.filter2D(parent, -1, k5, (-1, -1), 0.0, border, roi=roi), want)


def test_filter2d_4k_config2(cv, orc):
    """BASELINE config 2: cvtColor(BGR2GRAY) + filter2D 3x3 on 3840x2160 CV_8U."""
    bgr = rnd((2160, 3840, 3), np.uint8, 809564)
    gray = cv.cvtColor(dev(bgr), cv.COLOR_BGR2GRAY)
    want_gray = orc.orc_cvtColor(bgr, 6)
    check(gray, want_gray)
    out = cv.filter2D(gray, -1, SHARPEN)
    check(out, orc.orc_filter2D(want_gray, -1, SHARPEN))


def test_sepfilter_modes(cv, orc):
    src = rnd((41, 80, 3), np.uint8, 9)
    s3, s5 = [0.25, 0.5, 0.25], [0.0625, 0.25, 0.375, 0.25, 0.0625]
    for srcv in (src, np.ascontiguousarray(src[:, :77])):        # 77*3 = 231: 7 tail elements take the integer form
        for kx, ky in [(s3, s3), (s5, s3), (s5, s5)]:
            for border in BORDERS:
                check(cv.sepFilter2D(dev(srcv), -1, kx, ky, (-1, -1), 0.0, border), orc.orc_sepFilter2D(srcv, -1, kx, ky, (-1, -1), 0.0, border))
    for kx, ky in [([-1, 0, 1], [1, 2, 1]), ([1, 2, 1], [-1, 0, 1]), ([1, -2, 1], [3, 10, 3])]:
        for border in BORDERS:
            check(cv.sepFilter2D(dev(src), 3, kx, ky, (-1, -1), 2.0, border), orc.orc_sepFilter2D(src, 3, kx, ky, (-1, -1), 2.0, border))
    rng = np.random.default_rng(1)
    srcf = rnd((29, 64, 3), np.float32, 8)
    kx = rng.uniform(-1, 1, 7).astype(np.float32); ky = rng.uniform(-1, 1, 5).astype(np.float32)
    for border in BORDERS:
        check(cv.sepFilter2D(dev(srcf), -1, kx, ky, (2, 1), 0.1, border), orc.orc_sepFilter2D(srcf, -1, kx, ky, (2, 1), 0.1, border))


def test_sepfilter_and_sobel_roi(cv, orc):
    """a ROI inside a larger image with a non-isolated border: the taps left of / above the ROI are real pixels of the parent
    (the case the reference's Imgproc_Sobel.borderTypes checks)"""
    parent = rnd((40, 60), np.uint8, 77)
    pf = rnd((40, 60, 3), np.float32, 78)
    for roi in [(5, 4, 30, 20), (1, 1, 1, 1), (58, 38, 2, 2), (0, 0, 16, 16)]:
        for border in BORDERS:
            for dx, dy, k in [(1, 0, 3), (0, 1, 3), (1, 1, 5)]:
                want = orc.orc_Sobel(parent, 3, dx, dy, k, 1.0, 0.0, border, roi=roi)
                check(cv.Sobel(dev(parent), cv.CV_16S, dx, dy, k, 1.0, 0.0, border, roi=roi), want)
                check(cv.Sobel(parent, cv.CV_16S, dx, dy, k, 1.0, 0.0, border, roi=roi), want)
            s3 = [0.25, 0.5, 0.25]
            check(cv.sepFilter2D(dev(parent), -1, s3, s3, (-1, -1), 0.0, border, roi=roi), orc.orc_sepFilter2D(parent, -1, s3, s3, (-1, -1), 0.0, border, roi=roi))
            kx, ky = [0.1, 0.5, 0.2], [0.7, -0.1, 0.2]
            check(cv.sepFilter2D(dev(pf), -1, kx, ky, (-1, -1), 0.5, border, roi=roi), orc.orc_sepFilter2D(pf, -1, kx, ky, (-1, -1), 0.5, border, roi=roi))


def test_sepfilter_fixed_point_rolling(cv, orc):
    """8U -> 8U with taps that are multiples of 1/256 (the reference's integer row pass + float column pass): rows of a multiple of 16
    elements run on the rolling kernel, the others (scalar tail in the reference) on the generic one; both bit-exact"""
    taps = [[0.25, 0.5, 0.25], [0.125, 0.75, 0.125], [0.0625, 0.25, 0.375, 0.25, 0.0625], [0.0, 0.5, 0.0, 0.5, 0.0][:5]]
    for cn in (1, 3, 4):
        for (w, h) in [(64, 23), (1040, 37), (336, 19), (330, 11)]:
            src = rnd((h, w, cn) if cn > 1 else (h, w), np.uint8, w + cn)
            for k in taps:
                for dl in (0.0, 3.0):
                    for border in (0, 1, 2, 4):
                        check(cv.sepFilter2D(dev(src), -1, k, k, (-1, -1), dl, border), orc.orc_sepFilter2D(src, -1, k, k, (-1, -1), dl, border))
    big = rnd((2160, 3840), np.uint8, 5)
    check(cv.sepFilter2D(dev(big), -1, taps[0], taps[0]), orc.orc_sepFilter2D(big, -1, taps[0], taps[0]))


@pytest.mark.parametrize("ksize", [1, 3, 5, 7, -1])
def test_sobel_scharr(cv, orc, ksize):
    src8 = rnd((33, 70), np.uint8, 3)
    srcf = rnd((33, 70, 3), np.float32, 4)
    for dx, dy in [(1, 0), (0, 1)] + ([(1, 1), (2, 0)] if ksize >= 3 else []):
        for border in (0, 1, 2, 4):
            check(cv.Sobel(dev(src8), cv.CV_16S, dx, dy, ksize, 1.0, 0.0, border), orc.orc_Sobel(src8, 3, dx, dy, ksize, 1.0, 0.0, border))
            sc = 1.0 / (255.0 * 2 * 4)
            check(cv.Sobel(dev(src8), cv.CV_32F, dx, dy, ksize, sc, 0.0, border), orc.orc_Sobel(src8, 5, dx, dy, ksize, sc, 0.0, border))
            check(cv.Sobel(dev(srcf), -1, dx, dy, ksize, 1.0, 0.25, border), orc.orc_Sobel(srcf, -1, dx, dy, ksize, 1.0, 0.25, border))
    if ksize == -1:
        check(cv.Scharr(dev(src8), cv.CV_16S, 1, 0), orc.orc_Sobel(src8, 3, 1, 0, -1))


def test_sobel_box_rolling_path(cv, orc):
    """Geometries the rolling separable kernels take: Sobel/Scharr 8U->16S (ksize 3, 5, Scharr; every dx/dy order the
    int16 range check admits) and the normalised 8U box filter (3, 5, 7; cn 1, 3, 4); bit-exact."""
    for (w, h) in [(16, 1), (32, 2), (64, 23), (1040, 37), (2064, 70)]:
        src = rnd((h, w), np.uint8, w + h)
        for border in (0, 1, 2, 4):
            for ksize, orders in [(3, [(1, 0), (0, 1), (1, 1), (2, 0), (0, 2)]), (5, [(1, 0), (0, 1), (1, 1), (2, 0)]), (-1, [(1, 0), (0, 1)])]:
                for dx, dy in orders:
                    check(cv.Sobel(dev(src), cv.CV_16S, dx, dy, ksize, 1.0, 0.0, border), orc.orc_Sobel(src, 3, dx, dy, ksize, 1.0, 0.0, border))
        for cn in (1, 3, 4):
            if w * cn % 16:
                continue
            srcc = rnd((h, w, cn) if cn > 1 else (h, w), np.uint8, w + cn)
            for k in (3, 5, 7):
                for border in (0, 1, 2, 4):
                    check(cv.boxFilter(dev(srcc), -1, (k, k), (-1, -1), True, border), orc.orc_boxFilter(srcc, -1, (k, k), (-1, -1), True, border))
        # float path: Sobel 8U -> 32F with a scale (what cornerHarris asks for), sepFilter2D with non-smooth / unsymmetric taps
        sc = 1.0 / (255.0 * 2 * 4)
        for border in (0, 1, 2, 4):
            for ksize, orders in [(3, [(1, 0), (0, 1), (1, 1)]), (5, [(1, 0), (0, 1), (2, 0)]), (-1, [(1, 0), (0, 1)])]:
                for dx, dy in orders:
                    check(cv.Sobel(dev(src), cv.CV_32F, dx, dy, ksize, sc, 0.0, border), orc.orc_Sobel(src, 5, dx, dy, ksize, sc, 0.0, border))
            for kx, ky, dl in [([-1, 0, 1], [0.3, 0.4, 0.3], 0.0), ([0.1, 0.5, 0.2], [0.7, -0.1, 0.2], 3.5), ([0.1, 0.2, 0.4, 0.2, 0.1], [-2, -1, 0, 1, 2], 128.0),
                               ([0.05, 0.1, 0.4, 0.3, 0.15], [0.3, 0.3, 0.2, 0.1, 0.1], 0.0)]:
                check(cv.sepFilter2D(dev(src), -1, kx, ky, (-1, -1), dl, border), orc.orc_sepFilter2D(src, -1, kx, ky, (-1, -1), dl, border))
                check(cv.sepFilter2D(dev(src), cv.CV_32F, kx, ky, (-1, -1), dl, border), orc.orc_sepFilter2D(src, 5, kx, ky, (-1, -1), dl, border))
    # seven symmetric float taps, 8U -> 8U: cv::GaussianBlur(7 x 7, sigma 2) as the reference runs it on a submatrix (sepFilter2D with float taps; ORB's blur)
    g7 = orc.orc_getGaussianKernel(7, 2.0).astype(np.float32)
    for (w, h) in [(64, 23), (1040, 37), (333, 19), (97, 61), (17, 9)]:
        src = rnd((h, w), np.uint8, w + 7)
        for border in (4, 1, 2, 0):
            check(cv.sepFilter2D(dev(src), -1, g7, g7, (-1, -1), 0.0, border), orc.orc_sepFilter2D(src, -1, g7, g7, (-1, -1), 0.0, border))
        win = dev(np.pad(src, ((9, 5), (13, 6)), mode="edge"))[9:9 + h, 13:13 + w]             # unaligned rows inside a larger buffer, isolated border
        check(cv.sepFilter2D(win, -1, g7, g7, (-1, -1), 0.0, 4 | 16), orc.orc_sepFilter2D(src, -1, g7, g7, (-1, -1), 0.0, 4))
    # multi-channel derivative / float-tap filters on the rolling kernels
    for cn in (3, 4):
        for (w, h) in [(32, 5), (64, 23), (1040, 37), (333, 19)]:
            srcc = rnd((h, w, cn), np.uint8, w + cn)
            for border in (0, 1, 4):
                for ksize, dx, dy in [(3, 1, 0), (3, 0, 1), (5, 1, 1), (-1, 0, 1)]:
                    check(cv.Sobel(dev(srcc), cv.CV_16S, dx, dy, ksize, 1.0, 0.0, border), orc.orc_Sobel(srcc, 3, dx, dy, ksize, 1.0, 0.0, border))
                    check(cv.Sobel(dev(srcc), cv.CV_32F, dx, dy, ksize, 0.25, 0.0, border), orc.orc_Sobel(srcc, 5, dx, dy, ksize, 0.25, 0.0, border))
                for kx, ky, dl in [([0.1, 0.5, 0.2], [0.7, -0.1, 0.2], 3.5), ([0.05, 0.1, 0.4, 0.3, 0.15], [0.3, 0.3, 0.2, 0.1, 0.1], 0.0)]:
                    check(cv.sepFilter2D(dev(srcc), -1, kx, ky, (-1, -1), dl, border), orc.orc_sepFilter2D(srcc, -1, kx, ky, (-1, -1), dl, border))
                    check(cv.sepFilter2D(dev(srcc), cv.CV_32F, kx, ky, (-1, -1), dl, border), orc.orc_sepFilter2D(srcc, 5, kx, ky, (-1, -1), dl, border))
    full = np.full((40, 64), 255, np.uint8)
    for k in (3, 5, 7):
        assert (cv.blur(dev(full), (k, k)).cpu().numpy() == 255).all()


def test_boxfilter(cv, orc):
    # (integer sources are bit-exact into every destination depth, CV_32F included: ColumnSum<int, float>'s float / double split is reproduced)
    for dtype, ddepth in [(np.uint8, -1), (np.uint8, 5), (np.float32, -1), (np.uint16, -1), (np.int16, -1),
                          (np.uint8, 3), (np.uint8, 2), (np.uint16, 0), (np.uint16, 3), (np.uint16, 5), (np.int16, 5)]:
        src = rnd((31, 66 + (ddepth == 5), 3), dtype, 21)
        for ksize, anchor in [((3, 3), (-1, -1)), ((5, 5), (-1, -1)), ((2, 2), (-1, -1)), ((7, 3), (1, 2)), ((16, 16), (-1, -1)), ((17, 17), (-1, -1))]:
            for normalize in (True, False):
                for border in (0, 1, 4):
                    check(cv.boxFilter(dev(src), ddepth, ksize, anchor, normalize, border),
                          orc.orc_boxFilter(src, ddepth, ksize, anchor, normalize, border), tol=1e-6 if dtype == np.float32 else 0.0)
    src = rnd((20, 33), np.uint8, 2)
    check(cv.blur(src, (3, 3)), orc.orc_boxFilter(src, -1, (3, 3)))
    for dtype, ddepth in [(np.uint8, -1), (np.uint8, 5), (np.float32, -1), (np.uint16, -1), (np.int16, 5)]:       # 5 / 9 channels (Imgproc_FilterSupportedFormats blurs 5)
        for cn in (5, 9):
            src = rnd((31, 47, cn), dtype, 30 + cn)
            for ksize in [(3, 3), (11, 11), (4, 7)]:
                for normalize in (True, False):
                    for border in (0, 1, 4):
                        check(cv.boxFilter(dev(src), ddepth, ksize, (-1, -1), normalize, border),
                              orc.orc_boxFilter(src, ddepth, ksize, (-1, -1), normalize, border), tol=1e-6 if dtype == np.float32 else 0.0)


def test_boxfilter_two_pass(cv, orc):
    """windows the rolling kernels and k_sepmx do not take (CV_32F beyond 7 x 7, CV_8U into 16S / 32F, 16-bit sources, more than 4 channels) run as RowSum + ColumnSum
    (k_box_rows / k_box_cols) instead of kw * kh gathers per output: integer sums bit for bit into every depth, double sums within 1e-6 of the restatement's running sums;
    large windows (a guided filter's 61 x 61, 121 x 121), windows larger than the image, ROI windows, rows longer than a staging block"""
    for dtype, ddepth, cn in [(np.float32, -1, 1), (np.float32, -1, 3), (np.uint8, 5, 1), (np.uint8, 3, 3), (np.uint16, -1, 1), (np.int16, 5, 2), (np.uint8, -1, 6), (np.uint16, 5, 1)]:
        for shape, ksize, anchor in [((70, 1100), (61, 61), (-1, -1)), ((45, 200), (121, 121), (-1, -1)), ((33, 90), (5, 31), (2, 30)), ((20, 40), (41, 9), (0, 0)), ((9, 12), (25, 25), (-1, -1))]:
            src = rnd(shape + (cn,) if cn > 1 else shape, dtype, ksize[0] + cn)
            for normalize in (True, False):
                if normalize and dtype == np.uint16 and ksize[0] * ksize[1] > (1 << 15):
                    continue                                   # (the reference switches to double sums there: declined)
                for border in (0, 1, 2, 4):
                    got = cv.boxFilter(dev(src), ddepth, ksize, anchor, normalize, border)
                    assert "k_box_rows" in last_kernel(), last_kernel()
                    check(got, orc.orc_boxFilter(src, ddepth, ksize, anchor, normalize, border), tol=1e-6 if dtype == np.float32 else 0.0)
    parent = rnd((60, 120, 3), np.float32, 5)
    for roi in [(10, 8, 90, 40), (0, 0, 64, 16), (116, 57, 4, 3)]:
        for border in (1, 4):
            check(cv.boxFilter(dev(parent), -1, (15, 11), (-1, -1), True, border, roi=roi), orc.orc_boxFilter(parent, -1, (15, 11), (-1, -1), True, border, roi=roi), tol=1e-6)
            assert "k_box_rows" in last_kernel(), last_kernel()
    frames = rnd((3, 50, 70), np.float32, 6)
    got = cv.boxFilterBatch(dev(frames), -1, (21, 21)).cpu().numpy()
    for f in range(3):
        check(got[f], orc.orc_boxFilter(frames[f], -1, (21, 21)), tol=1e-6)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_cvtcolor(cv, orc, dtype):
    for (w, h) in [(1, 1), (7, 3), (64, 5), (263, 31), (1024, 4)]:
        for code in range(12):
            scn = {0: 3, 1: 4, 2: 3, 3: 4, 4: 3, 5: 4, 6: 3, 7: 3, 8: 1, 9: 1, 10: 4, 11: 4}[code]
            src = rnd((h, w, scn) if scn > 1 else (h, w), dtype, 1000 + code + w)
            want = orc.orc_cvtColor(src, code)
            check(cv.cvtColor(dev(src), code), want, tol=1e-6)
            if w == 263:
                check(cv.cvtColor(src, code), want, tol=1e-6)


def test_cvtcolor_known_answer_hash(cv):
    """Imgproc_cvtColor_BE (test_color.cpp:2847-2849): adler32 of the GPU result on the RNG(0) image."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "color_rng0.npz"))
    assert zlib.adler32(cv.cvtColor(dev(g["src"]), cv.COLOR_BGR2GRAY).cpu().numpy().tobytes()) == 0x3008c6b8
    assert zlib.adler32(cv.cvtColor(dev(g["src"]), cv.COLOR_RGB2GRAY).cpu().numpy().tobytes()) == 0x416bd44a
    frames = torch.from_numpy(np.stack([g["src"]] * 3)).cuda()
    out = cv.cvtColorBatch(frames, cv.COLOR_BGR2GRAY)
    for f in range(3):
        assert zlib.adler32(out[f].cpu().numpy().tobytes()) == 0x3008c6b8


def _kernel(cv):
    from opencv_amd import _lib
    return _lib.lib.mi355cv_lastKernel().decode()


def test_f32_rolling_filters(cv, orc):
    """CV_32FC1 -> CV_32FC1 on the rolling kernels (north_star's second parity class): sepFilter2D with symmetric / antisymmetric / general taps of 3, 5
    and 7, Sobel, GaussianBlur (which the reference runs as sepFilter2D with float taps), boxFilter with its double sums; ragged and unaligned rows;
    <= 1e-6 relative against the restatement (the contract is 1e-4)."""
    rng = np.random.default_rng(3)
    for (w, h) in [(64, 23), (1040, 37), (333, 19), (4, 5), (2064, 70)]:
        src = rnd((h, w), np.float32, w + h) * 255 - 100
        for border in (0, 1, 2, 4):
            for kx, ky, dl in [([0.25, 0.5, 0.25], [0.25, 0.5, 0.25], 0.0), ([-1, 0, 1], [1, 2, 1], 0.5), ([0.1, 0.5, 0.2], [0.7, -0.1, 0.2], 3.5),
                               ([0.0625, 0.25, 0.375, 0.25, 0.0625], [0.0625, 0.25, 0.375, 0.25, 0.0625], 0.0), ([-1, -2, 0, 2, 1], [1, 4, 6, 4, 1], 0.0),
                               (rng.uniform(-1, 1, 7).tolist(), rng.uniform(-1, 1, 7).tolist(), 0.25), ([0.05, 0.1, 0.4, 0.3, 0.15], [0.3, 0.3, 0.2, 0.1, 0.1], 0.0)]:
                check(cv.sepFilter2D(dev(src), -1, kx, ky, (-1, -1), dl, border), orc.orc_sepFilter2D(src, -1, kx, ky, (-1, -1), dl, border), tol=1e-6)
                if w >= 64:
                    assert "k_sep_roll<SepF32F" in _kernel(cv), _kernel(cv)
            for ksize, dx, dy in [(3, 1, 0), (3, 0, 1), (5, 1, 1), (7, 2, 0), (-1, 0, 1)]:
                check(cv.Sobel(dev(src), -1, dx, dy, ksize, 0.5, 0.25, border), orc.orc_Sobel(src, -1, dx, dy, ksize, 0.5, 0.25, border), tol=1e-6)
            for k in (3, 5, 7):
                for normalize in (True, False):
                    check(cv.boxFilter(dev(src), -1, (k, k), (-1, -1), normalize, border), orc.orc_boxFilter(src, -1, (k, k), (-1, -1), normalize, border), tol=1e-6)
        for ks, sg in [(3, 0.0), (5, 1.2), (7, 0.0), (5, 0.0)]:
            want = orc.ref_GaussianBlur(src, ks, sg, sg, 4) if orc.load_ref() is not None else None
            got = cv.GaussianBlur(dev(src), (ks, ks), sg)
            if want is not None:
                check(got, want, tol=1e-6)
    # a whole 4K frame, and the same through the batch entries
    big = rnd((2160, 3840), np.float32, 11)
    g5 = cv.getGaussianKernel(5, 1.2, cv.CV_32F)
    want = orc.orc_sepFilter2D(big, -1, g5, g5)
    check(cv.sepFilter2D(dev(big), -1, g5, g5), want, tol=1e-6)
    fb = dev(np.stack([big, big[::-1].copy()]))
    check(cv.sepFilter2DBatch(fb, -1, g5, g5)[0], want, tol=1e-6)
    check(cv.boxFilterBatch(fb, -1, (5, 5))[0], orc.orc_boxFilter(big, -1, (5, 5)), tol=1e-6)
    check(cv.SobelBatch(fb, cv.CV_32F, 1, 0, 3)[0], orc.orc_Sobel(big, -1, 1, 0, 3), tol=1e-6)


@pytest.mark.parametrize("dtype", [np.uint16, np.int16])
def test_16bit_sources_on_the_rolling_kernels(cv, orc, dtype):
    """CV_16UC1 / CV_16SC1 sepFilter2D and Sobel / Scharr with float taps (the reference's RowFilter<ushort|short, float> + column filter with cvRound + saturate),
    to the same depth and to CV_32F, on k_sep_roll<SepF16...>: a 16-bit element is a 2-byte pixel of the rolling skeleton.  Ragged and unaligned rows, every border
    rule, values that saturate; bit-exact for the 16-bit outputs, <= 1e-6 relative for CV_32F."""
    rng = np.random.default_rng(9)
    lo, hi = (0, 65536) if dtype == np.uint16 else (-32768, 32768)
    for (w, h) in [(64, 23), (1040, 37), (333, 19), (8, 5), (2064, 70)]:
        src = rng.integers(lo, hi, (h, w)).astype(dtype)
        for border in (0, 1, 2, 4):
            for kx, ky, dl in [([0.25, 0.5, 0.25], [0.25, 0.5, 0.25], 0.0), ([-1, 0, 1], [1, 2, 1], 0.5), ([0.1, 0.5, 0.2], [0.7, -0.1, 0.2], 3.5),
                               ([0.0625, 0.25, 0.375, 0.25, 0.0625], [0.0625, 0.25, 0.375, 0.25, 0.0625], 0.0), ([-1, -2, 0, 2, 1], [1, 4, 6, 4, 1], 0.0)]:
                for ddepth in (-1, 5):
                    got = cv.sepFilter2D(dev(src), ddepth, kx, ky, (-1, -1), dl, border)
                    check(got, orc.orc_sepFilter2D(src, ddepth, kx, ky, (-1, -1), dl, border), tol=1e-6)
                    if w >= 64 and w % 2 == 0:                               # rows of a multiple of 4 bytes
                        assert "k_sep_roll<SepF16" in _kernel(cv), _kernel(cv)
            if dtype == np.int16:
                for ksize, dx, dy in [(3, 1, 0), (3, 0, 1), (5, 1, 1), (-1, 0, 1)]:
                    check(cv.Sobel(dev(src // 64), -1, dx, dy, ksize, 1.0, 0.0, border), orc.orc_Sobel(src // 64, -1, dx, dy, ksize, 1.0, 0.0, border))
    big = rng.integers(lo, hi, (2160, 3840)).astype(dtype)
    g5 = cv.getGaussianKernel(5, 1.2, cv.CV_32F)
    want = orc.orc_sepFilter2D(big, -1, g5, g5)
    check(cv.sepFilter2D(dev(big), -1, g5, g5), want)
    assert "k_sep_roll<SepF16" in _kernel(cv), _kernel(cv)
    fb = dev(np.stack([big, big[::-1].copy()]))
    check(cv.sepFilter2DBatch(fb, -1, g5, g5)[0], want)


def _np_binom16(src, k, border):
    """cv::GaussianBlur on CV_16U with sigma 0: the plain integer binomial sums with ONE rounding, (S + 8) >> 4 for 3x3 and (S + 128) >> 8 for 5x5 (what the
    reference's Q16.16 hline / vline pair evaluates to, smooth.simd.hpp:1425-1452, :1596-1632; pinned against the reference in tests/test_oracle_smooth16.py)"""
    mode = {0: "constant", 1: "edge", 2: "symmetric", 3: "wrap", 4: "reflect"}[border]
    r = k // 2
    p = np.pad(src.astype(np.int64), r, mode=mode)
    t = np.array([1, 2, 1] if k == 3 else [1, 4, 6, 4, 1], np.int64)
    h = sum(t[i] * p[:, i:i + src.shape[1]] for i in range(k))
    v = sum(t[i] * h[i:i + src.shape[0]] for i in range(k))
    return ((v + (8 if k == 3 else 128)) >> (4 if k == 3 else 8)).astype(np.uint16)


def test_gaussian_16u_binomial_on_the_rolling_kernel(cv, orc):
    """cv_hal_gaussianBlurBinomial on CV_16UC1 (the one hook the reference's Q16.16 GaussianBlur path has, smooth.dispatch.cpp:726-760): k_sep_roll<Binom16>,
    bit-exact against the integer restatement and, where the real reference travelled with the tree, against cv::GaussianBlur itself"""
    rng = np.random.default_rng(4)
    have_ref = orc.load_ref() is not None
    for (w, h) in [(64, 23), (1040, 37), (334, 19), (8, 5), (2064, 70), (3840, 2160)]:
        src = rng.integers(0, 65536, (h, w)).astype(np.uint16)
        src[0, :8] = 65535; src[-1, -8:] = 65535
        for k in (3, 5):
            for border in ((0, 1, 2, 4) if w < 3000 else (4,)):
                got = cv.GaussianBlur(dev(src), (k, k), 0, borderType=border)
                assert "k_sep_roll<Binom16" in _kernel(cv), _kernel(cv)
                check(got, _np_binom16(src, k, border))
                if have_ref and w < 3000:
                    check(got, orc.ref_GaussianBlur(src, k, 0.0, 0.0, border))
    with pytest.raises(NotImplementedError):
        cv.GaussianBlur(dev(np.zeros((16, 16), np.uint16)), (7, 7), 1.5)               # sigma != 0: the reference's Q16.16 path has no hook for it


def test_gaussian_16u_binomial_beyond_the_rolling_kernel(cv, orc):
    """round 5 (80 of the 88 hook calls of GaussianBlur_Bitexact.Linear16U and overflow_20121 were declined): 7 / 9 taps, 2-4 channels, BORDER_WRAP, images smaller
    than the kernel -- k_binom16_direct, bit for bit against the integer restatement tests/test_oracle_smooth16.py pins to the reference, and against the reference itself"""
    from test_oracle_smooth16 import np_binom16
    rng = np.random.default_rng(14)
    have_ref = orc.load_ref() is not None
    for (w, h) in [(1, 3), (3, 1), (2, 2), (3, 3), (5, 5), (7, 7), (37, 23), (256, 128), (1283, 70)]:
        for cn in (1, 2, 3, 4):
            src = rng.integers(0, 65536, (h, w, cn) if cn > 1 else (h, w)).astype(np.uint16)
            src.flat[:: max(1, src.size // 5)] = 65535
            for k in (3, 5, 7, 9):
                for border in (0, 1, 2, 3, 4):
                    if (w == 1 or h == 1) and border != 0:
                        continue                                     # cv::GaussianBlur clamps the kernel in a one-pixel dimension (no square kernel, no hook call)
                    got = cv.GaussianBlur(dev(src), (k, k), 0, borderType=border | 16)
                    check(got, np_binom16(src, k, border))
                    if have_ref and w * h < 40000:
                        check(got, orc.ref_GaussianBlur(src, k, 0.0, 0.0, border | 16))
    assert "k_binom16_direct" in _kernel(cv)
    full = np.full((100, 100), 65535, np.uint16)                     # GaussianBlur_Bitexact.overflow_20121
    assert int(cv.GaussianBlur(dev(full), (9, 9), 0).cpu().numpy().min()) == 65535
    # a window of a larger image with real pixels around it (the hook's margins): equal to the same window of the filtered parent away from the parent's border
    parent = rng.integers(0, 65536, (60, 90, 3)).astype(np.uint16)
    whole = np_binom16(parent, 7, 4)
    got = cv.GaussianBlur(dev(parent), (7, 7), 0, borderType=4, roi=(10, 8, 50, 30)) if "roi" in cv.GaussianBlur.__code__.co_varnames else None
    if got is not None:
        check(got, whole[8:38, 10:60])


def test_submatrix_calls_stay_on_the_rolling_kernels(cv, orc):
    """A cv::Mat ROI with real pixels around it (the HAL's offset / full-size and margin contracts): the rolling kernels run on the parent's geometry and
    store the window only -- every window position relative to the 16-byte chunk grid, windows touching the parent's edges, one-pixel windows;
    8U fixed-point sepFilter2D, Sobel 8U->16S, float-tap sepFilter2D 8U->8U / 32F, CV_32F sepFilter2D and box, 8U box; results equal to the
    restatement's ROI call; the kernel that ran is a rolling one."""
    parent = rnd((96, 400), np.uint8, 77)
    pf = rnd((96, 400), np.float32, 78)
    s3 = [0.25, 0.5, 0.25]
    rolled = 0
    for roi in [(16, 8, 256, 40), (5, 4, 130, 20), (33, 1, 64, 94), (0, 0, 400, 96), (0, 3, 17, 5), (383, 90, 17, 6), (100, 50, 1, 1), (17, 17, 335, 3), (1, 0, 398, 96)]:
        for border in (0, 1, 2, 4):
            check(cv.Sobel(dev(parent), cv.CV_16S, 1, 0, 3, 1.0, 0.0, border, roi=roi), orc.orc_Sobel(parent, 3, 1, 0, 3, 1.0, 0.0, border, roi=roi))
            rolled += "k_sep_roll<Deriv16" in _kernel(cv) and "window" in _kernel(cv)
            check(cv.Sobel(dev(parent), cv.CV_16S, 1, 1, 5, 1.0, 0.0, border, roi=roi), orc.orc_Sobel(parent, 3, 1, 1, 5, 1.0, 0.0, border, roi=roi))
            check(cv.sepFilter2D(dev(parent), -1, s3, s3, (-1, -1), 0.0, border, roi=roi), orc.orc_sepFilter2D(parent, -1, s3, s3, (-1, -1), 0.0, border, roi=roi))
            kx, ky = [0.1, 0.5, 0.2], [0.7, -0.1, 0.2]
            check(cv.sepFilter2D(dev(parent), -1, kx, ky, (-1, -1), 3.5, border, roi=roi), orc.orc_sepFilter2D(parent, -1, kx, ky, (-1, -1), 3.5, border, roi=roi))
            check(cv.sepFilter2D(dev(parent), cv.CV_32F, kx, ky, (-1, -1), 3.5, border, roi=roi), orc.orc_sepFilter2D(parent, 5, kx, ky, (-1, -1), 3.5, border, roi=roi), tol=1e-6)
            check(cv.sepFilter2D(dev(pf), -1, kx, ky, (-1, -1), 0.5, border, roi=roi), orc.orc_sepFilter2D(pf, -1, kx, ky, (-1, -1), 0.5, border, roi=roi), tol=1e-6)
            g5 = [0.0625, 0.25, 0.375, 0.25, 0.0625]
            check(cv.sepFilter2D(dev(pf), -1, g5, g5, (-1, -1), 0.0, border, roi=roi), orc.orc_sepFilter2D(pf, -1, g5, g5, (-1, -1), 0.0, border, roi=roi), tol=1e-6)
            check(cv.boxFilter(dev(parent), -1, (5, 5), (-1, -1), True, border, roi=roi), orc.orc_boxFilter(parent, -1, (5, 5), (-1, -1), True, border, roi=roi))
            check(cv.boxFilter(dev(pf), -1, (3, 3), (-1, -1), True, border, roi=roi), orc.orc_boxFilter(pf, -1, (3, 3), (-1, -1), True, border, roi=roi), tol=1e-6)
    assert rolled >= 30, rolled


def test_filter2d_f32_rolling(cv, orc):
    """cv::filter2D CV_32FC1 -> CV_32FC1, 3x3 and 5x5 with a centred anchor, on the rolling kernel (delta + the taps in raster order, one FMA each): <= 1e-6 of the
    restatement; other anchors / sizes / channel counts keep the generic kernel and the same bound"""
    rng = np.random.default_rng(12)
    for (w, h) in [(64, 23), (1040, 37), (333, 19), (4, 5), (2064, 70)]:
        src = rnd((h, w), np.float32, w + h) * 200 - 50
        for K in (3, 5):
            k = rng.uniform(-1, 1, (K, K)).astype(np.float32)
            for border in (0, 1, 2, 4):
                check(cv.filter2D(dev(src), -1, k, (-1, -1), 0.75, border), orc.orc_filter2D(src, -1, k, (-1, -1), 0.75, border), tol=1e-6)
                if w >= 64:
                    assert "k_filter2d_roll_f32" in _kernel(cv), _kernel(cv)
        check(cv.filter2D(dev(src), -1, SHARPEN, (-1, -1), 0.0, 4), orc.orc_filter2D(src, -1, SHARPEN, (-1, -1), 0.0, 4), tol=1e-6)
    big = rnd((2160, 3840), np.float32, 3)
    k = rng.uniform(-1, 1, (3, 3)).astype(np.float32)
    check(cv.filter2D(dev(big), -1, k), orc.orc_filter2D(big, -1, k), tol=1e-6)
    src3 = rnd((40, 64, 3), np.float32, 5)
    check(cv.filter2D(dev(src3), -1, k), orc.orc_filter2D(src3, -1, k), tol=1e-6)


def test_large_filter2d_opt_in(orc):
    """filter2D with >= 130 taps on a whole image is the reference's DFT case and is declined by default (the FFTs' float error cannot be reproduced by a sum);
    MI355CV_FILTER_LARGE=1 serves it with the direct sum = the reference's own non-DFT engine: bit for bit against the restatement of that engine, and within 1 grey level
    (CV_8U) / 1e-5 relative (CV_32F) of what the reference's DFT path returns.  A process of its own: the switch is read once."""
    import subprocess, sys, textwrap
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
        import opencv_amd as cv, orc
        rng = np.random.default_rng(9)
        k = (rng.random((13, 11), dtype=np.float32) - 0.4).astype(np.float32); k /= np.abs(k).sum()
        for dtype in (np.uint8, np.float32):
            src = rng.integers(0, 256, (240, 320)).astype(dtype) if dtype == np.uint8 else rng.random((240, 320), dtype=np.float32)
            try:
                got = cv.filter2D(torch.from_numpy(src).cuda(), -1, k).cpu().numpy()
            except NotImplementedError:
                print("DECLINED", np.dtype(dtype).name); continue
            want = orc.orc_filter2D(src, -1, k)
            exact = bool(np.array_equal(got, want)) if dtype == np.uint8 else float(np.abs(got - want).max()) <= 1e-5
            dev = -1.0
            if orc.load_ref() is not None:
                ref = orc.ref_filter2D(src, -1, k)
                dev = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) if dtype == np.uint8 else float(np.abs(got - ref).max() / np.abs(ref).max())
            print("SERVED", np.dtype(dtype).name, "direct-engine-exact", exact, "max-deviation-from-the-DFT-path", dev)
    """ % (ROOT, ROOT))
    for flag in ("", "0", "1"):
        env = dict(os.environ); env.pop("MI355CV_FILTER_LARGE", None)
        if flag: env["MI355CV_FILTER_LARGE"] = flag
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        if flag == "":
            # the default since round 6: CV_32F destinations are served (the float bar is 1e-4; the direct sum is ~1e-6 from the DFT result), CV_8U stays with the CPU
            assert "DECLINED uint8" in p.stdout and "DECLINED float32" not in p.stdout, p.stdout
            l = [l.split() for l in p.stdout.splitlines() if l.startswith("SERVED float32")]
            assert len(l) == 1 and l[0][3] == "True" and float(l[0][5]) <= 1e-5, p.stdout
        elif flag == "0":
            assert p.stdout.count("DECLINED") == 2, p.stdout
        else:
            lines = [l.split() for l in p.stdout.splitlines() if l.startswith("SERVED")]
            assert len(lines) == 2 and all(l[3] == "True" for l in lines), p.stdout
            d8, d32 = float(lines[0][5]), float(lines[1][5])
            assert d8 <= 1.0 and d32 <= 1e-5, p.stdout                       # (-1: the reference did not travel with the tree)



def test_filter2d_tile_kernel(cv, orc):
    """filter2D kernels the rolling path does not take run on k_filter2d_tile (LDS-staged source box, a sliding float window per kernel row): the chain per output is the
    reference's non-DFT engine's (delta, then fma over the non-zero taps in raster order, one rounding), so integer results are bit for bit and float results equal the
    restatement's to the last bit.  Kernel shapes with ragged chunks, sparse rows (zero taps stay out of the chain), every depth pair / border / anchor, 1-4 channels, ROI
    windows (the parent's pixels as border), images smaller than a tile and smaller than the kernel."""
    rng = np.random.default_rng(21)
    def kern(kh, kw, sparse=False):
        k = (rng.uniform(-1, 1, (kh, kw)) / (0.3 * kh * kw)).astype(np.float32)
        if sparse: k[rng.random((kh, kw)) < 0.6] = 0
        return k
    n0 = cv.call_count("filter")
    # depth pairs on a 3-channel image, 7 x 7 (whole image: below the DFT bound)
    k7 = kern(7, 7)
    for dtype, ddepth in [(np.uint8, -1), (np.uint8, 3), (np.uint8, 5), (np.uint8, 2), (np.uint16, -1), (np.uint16, 5), (np.int16, -1), (np.int16, 5), (np.float32, -1)]:
        src = rnd((45, 83, 3), dtype, 3)
        for border in BORDERS + [3]:
            want = orc.orc_filter2D(src, ddepth, k7, (-1, -1), 0.25, border)
            got = cv.filter2D(dev(src), ddepth, k7, (-1, -1), 0.25, border)
            assert "k_filter2d_tile" in last_kernel(), last_kernel()
            check(got, want, tol=1e-6)
    # shapes: ragged chunks, one row / one column, sparse, odd anchors; 1, 2 and 4 channels; widths across the 64-column tile edge
    for (kh, kw), anchor, sparse in [((3, 3), (0, 2), False), ((9, 5), (-1, -1), False), ((5, 9), (8, 0), True), ((11, 11), (-1, -1), False), ((1, 9), (3, 0), False),
                                     ((9, 1), (0, 7), False), ((4, 6), (5, 3), True), ((10, 12), (-1, -1), True)]:
        k = kern(kh, kw, sparse)
        for shape in [(19, 130), (16, 64, 2), (33, 65, 4), (3, 5), (1, 1)]:
            src = rnd(shape, np.uint8, kh * 100 + kw)
            for border in (0, 1, 4):
                want = orc.orc_filter2D(src, -1, k, anchor, 1.5, border)
                check(cv.filter2D(dev(src), -1, k, anchor, 1.5, border), want)
                assert "k_filter2d_tile" in last_kernel(), last_kernel()
    # large kernels: CV_32F whole images are served by default (the reference's DFT case, float bar), CV_8U through a ROI window (a submatrix is never the DFT case)
    for (kh, kw) in [(13, 10), (21, 21), (31, 31), (32, 32)]:
        k = kern(kh, kw)
        srcf = rnd((70, 150), np.float32, kh)
        check(cv.filter2D(dev(srcf), -1, k, (-1, -1), 0.0, 4), orc.orc_filter2D(srcf, -1, k, (-1, -1), 0.0, 4), tol=1e-6)
        assert "k_filter2d_tile" in last_kernel(), last_kernel()
        parent = rnd((90, 140, 3), np.uint8, kw)
        for roi in [(20, 25, 100, 40), (0, 0, 64, 16), (130, 80, 10, 10)]:
            for border in (1, 4):
                want = orc.orc_filter2D(parent, -1, k, (-1, -1), 0.0, border, roi=roi)
                check(cv.filter2D(dev(parent), -1, k, (-1, -1), 0.0, border, roi=roi), want)
                assert "k_filter2d_tile" in last_kernel(), last_kernel()
            with pytest.raises(NotImplementedError):         # an ISOLATED window is a whole image to cv::filter2D (filter.dispatch.cpp:1536-1538): CV_8U, the DFT case
                cv.filter2D(dev(parent), -1, k, (-1, -1), 0.0, 4 | 16, roi=roi)
    # the batch entry: frames along grid z
    frames = rnd((5, 40, 70), np.uint8, 8)
    k = kern(9, 9)
    got = cv.filter2DBatch(dev(frames), -1, k).cpu().numpy()
    assert "k_filter2d_tile" in last_kernel(), last_kernel()
    for f in range(5):
        assert np.array_equal(got[f], orc.orc_filter2D(frames[f], -1, k))
    assert cv.call_count("filter") > n0


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32, np.float64])
def test_filters_into_64f(cv, orc, dtype):
    """filter2D / sepFilter2D / Sobel / Scharr with a CV_64F destination (the reference's Imgproc_FilterSupportedFormats pairs 8U / 16U / 16S / 64F -> 64F, 5 channels):
    double kernels, double rows, every multiply-add fused like the reference's AVX2 + FMA copy -- bit for bit against oracle/filter64.c (tests/test_oracle_filter64.py:
    == the reference).  CV_32F -> CV_64F exists for the separable engine only (getLinearFilter has no such pair): filter2D declines it."""
    from test_oracle_filter64 import source, bits
    rng = np.random.default_rng(4)
    kx, ky = rng.uniform(-1, 1, 11).astype(np.float32), rng.uniform(-1, 1, 7).astype(np.float32)
    for cn in (1, 3, 5):
        src = source((31, 47, cn) if cn > 1 else (31, 47), dtype, 7 + cn)
        d = torch.from_numpy(src).cuda()
        for k in (rng.uniform(-10, 10, (5, 5)).astype(np.float32), rng.uniform(-1, 1, (3, 7)).astype(np.float32), rng.uniform(-1, 1, (4, 2))):
            for border, delta, anchor in ((4, 0.0, (-1, -1)), (0, 0.5, (-1, -1)), (1, -3.25, (0, 1)), (2, 0.0, (-1, -1))):
                if dtype == np.float32:
                    with pytest.raises(NotImplementedError):
                        cv.filter2D(d, 6, k, anchor, delta, border)
                    continue
                got = cv.filter2D(d, 6, k, anchor, delta, border).cpu().numpy()
                assert got.dtype == np.float64 and np.array_equal(bits(got), bits(orc.orc_filter2D(src, 6, k, anchor, delta, border))), (cn, k.shape, border)
        for (a, b) in ((kx, ky), (kx + kx[::-1], ky + ky[::-1]), (kx - kx[::-1], ky - ky[::-1]), (kx[:4], ky[:2])):
            for border, delta in ((4, 0.0), (0, 1.5), (1, 0.0), (2, -0.75)):
                got = cv.sepFilter2D(d, 6, a, b, (-1, -1), delta, border).cpu().numpy()
                assert np.array_equal(bits(got), bits(orc.orc_sepFilter2D(src, 6, a, b, (-1, -1), delta, border))), (cn, len(a), border)
        for (dx, dy, ks, scale) in ((1, 0, 3, 1.0), (2, 0, 5, 1.0), (1, 1, 5, 0.37), (0, 2, 7, 1.0)):
            got = cv.Sobel(d, 6, dx, dy, ks, scale, 0.25, 4).cpu().numpy()
            assert np.array_equal(bits(got), bits(orc.orc_Sobel(src, 6, dx, dy, ks, scale, 0.25, 4))), (cn, dx, dy, ks)
        for (dx, dy, scale) in ((0, 1, 1.0), (1, 0, 2.5)):
            got = cv.Scharr(d, 6, dx, dy, scale, 0.25, 4).cpu().numpy()
            assert np.array_equal(bits(got), bits(orc.orc_Sobel(src, 6, dx, dy, -1, scale, 0.25, 4))), (cn, dx, dy)
    if dtype != np.float32:
        from opencv_amd import _lib
        cv.filter2D(torch.from_numpy(source((31, 47), dtype, 3)).cuda(), 6, np.ones((3, 3), np.float32))
        assert "k_filter2d_generic64" in _lib.lib.mi355cv_lastKernel().decode()


def test_boxfilter_into_64f(cv, orc):
    """boxFilter / blur with a CV_64F destination: exact for integer sources (int sum, one double multiply), 1e-13 relative for float / double sources (double sums)"""
    rng = np.random.default_rng(1)
    for dtype, dd in [(np.uint8, 6), (np.uint16, 6), (np.int16, 6), (np.float32, 6), (np.float64, 6), (np.float64, -1)]:
        for cn in (1, 3, 5):
            src = (rng.random((31, 47, cn)) * 200 - 50).astype(dtype) if dtype in (np.float32, np.float64) else rng.integers(0, 200, (31, 47, cn)).astype(dtype)
            for ks in ((3, 3), (11, 11), (4, 7)):
                for norm in (True, False):
                    for border in (0, 1, 4):
                        got = cv.boxFilter(torch.from_numpy(src).cuda(), dd, ks, (-1, -1), norm, border).cpu().numpy()
                        want = orc.orc_boxFilter(src, dd, ks, normalize=norm, border=border)
                        assert got.dtype == np.float64
                        if dtype in (np.float32, np.float64):
                            assert np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max()), (dtype, cn, ks, norm, border)
                        else:
                            assert np.array_equal(got, want), (dtype, cn, ks, norm, border)



def test_sepfilter_long_kernels(cv, orc):
    """separable kernels of 34-129 taps per axis (Imgproc_GaussianBlur.regression_11303: a 71-tap Gaussian on CV_32F) on the LDS-ring kernel (seplong.hip): BIT for bit
    against the restatement, floats included -- the kernel keeps the order of every multiply-add chain (tests/test_seplong_emu.py replays the same lines on the CPU)"""
    rng = np.random.default_rng(9)
    g71 = np.exp(-0.5 * ((np.arange(71) - 35) / 8.64421) ** 2); g71 = (g71 / g71.sum()).astype(np.float32)
    g41 = np.exp(-0.5 * ((np.arange(41) - 20) / 6.5) ** 2); g41 = (g41 / g41.sum()).astype(np.float32)
    kx, ky = rng.uniform(-1, 1, 41).astype(np.float32) / 8, rng.uniform(-1, 1, 37).astype(np.float32) / 8
    for dtype, ddepth in [(np.float32, -1), (np.uint8, -1), (np.uint8, 5), (np.uint16, 5), (np.int16, -1)]:
        for cn in (1, 3):
            src = rnd((53, 90, cn) if cn > 1 else (53, 90), dtype, 60 + cn)
            for (a, b) in ((g71, g71), (g41, g71), (kx, ky), (g41, ky[:5])):
                for border in (4, 0, 1):
                    got = cv.sepFilter2D(dev(src), ddepth, a, b, (-1, -1), 0.0, border).cpu().numpy()
                    want = orc.orc_sepFilter2D(src, ddepth, a, b, (-1, -1), 0.0, border)
                    assert np.array_equal(got.view(np.uint32) if got.dtype == np.float32 else got, want.view(np.uint32) if want.dtype == np.float32 else want), (dtype, ddepth, cn, len(a), len(b), border)
    from opencv_amd import _lib
    assert "k_seplong<0," in _lib.lib.mi355cv_lastKernel().decode()
    big = rnd((211, 2115), np.float32, 3)                                              # the reference test's geometry
    got = cv.GaussianBlur(dev(big), (0, 0), 8.64421).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), orc.orc_sepFilter2D(big, -1, g71_exact(orc), g71_exact(orc)).view(np.uint32))
    # the tap bound itself, from the library (tests/test_declines_cpu.py pins its value): served at the bound, refused above it
    top = cv.limit("sep_max_taps")
    kt = np.full(top, 1.0 / top, np.float32); src = rnd((140, 300), np.float32, 8)
    got = cv.sepFilter2D(dev(src), -1, kt, kt, (-1, -1), 0.0, 4).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), orc.orc_sepFilter2D(src, -1, kt, kt, (-1, -1), 0.0, 4).view(np.uint32))
    with pytest.raises(NotImplementedError):
        cv.sepFilter2D(dev(src), -1, np.full(top + 2, 1.0 / (top + 2), np.float32), kt, (-1, -1), 0.0, 4)


def test_seplong_every_engine_and_geometry(cv, orc):
    """what the LDS-ring kernel takes over from the one-thread-per-output kernels (10-33 taps, odd anchors, even lengths, 2 channels, ragged widths, ROI windows with real
    pixels around them, batches): every engine of createSeparableLinearFilter bit for bit, on shapes that span several strips and segments"""
    rng = np.random.default_rng(4)
    g19 = np.exp(-0.5 * ((np.arange(19) - 9) / 3.0) ** 2); g19 = (g19 / g19.sum()).astype(np.float32)
    q11 = np.round(np.exp(-0.5 * ((np.arange(11) - 5) / 2.0) ** 2) * 40); q11[5] += 256 - q11.sum(); q11 = (q11 / 256).astype(np.float32)
    b11 = np.array([1.0]);
    for _ in range(8): b11 = np.convolve(b11, [1, 1])
    d11 = np.convolve(b11, [-1, 0, 1]).astype(np.float32); b11 = np.convolve(b11, [1, 2, 1]).astype(np.float32)
    r12, r10 = (rng.uniform(-1, 1, 12) / 6).astype(np.float32), (rng.uniform(-1, 1, 10) / 6).astype(np.float32)
    bits = lambda a: a.view(np.uint32) if a.dtype == np.float32 else a
    from opencv_amd import _lib
    for dtype, ddepth, kx, ky, anchor, delta, tag in [(np.float32, -1, g19, g19, (-1, -1), 0.0, "k_seplong<0,"), (np.float32, -1, r12, r10, (3, 7), 0.5, "k_seplong<0,"),
                                                      (np.uint8, -1, g19, g19, (-1, -1), 0.0, "k_seplong<0,"), (np.uint8, -1, q11, q11, (-1, -1), 1.0, "k_seplong<1,"),
                                                      (np.uint8, 3, b11, d11, (-1, -1), 0.0, "k_seplong<2,"), (np.uint16, 5, g19, r10, (-1, 2), 0.0, "k_seplong<0,")]:
        for cn in (1, 2, 3, 4):
            for (h, w) in [(300, 1000), (37, 53), (1, 40), (40, 1)]:
                if cn in (2, 4) and h == 300:
                    continue
                src = rnd((h, w, cn) if cn > 1 else (h, w), dtype, 11 * cn + h)
                for border in (4, 0, 1, 2):
                    got = cv.sepFilter2D(dev(src), ddepth, kx, ky, anchor, delta, border).cpu().numpy()
                    assert tag in _lib.lib.mi355cv_lastKernel().decode(), _lib.lib.mi355cv_lastKernel().decode()
                    assert np.array_equal(bits(got), bits(orc.orc_sepFilter2D(src, ddepth, kx, ky, anchor, delta, border))), (dtype, ddepth, len(kx), cn, (h, w), border)
        parent = rnd((96, 400), dtype, 77)
        for roi in [(16, 8, 256, 40), (5, 4, 130, 20), (383, 90, 17, 6), (0, 0, 400, 96)]:
            x, y, w, h = roi
            for border in (4, 0, 1):
                got = cv.sepFilter2D(dev(parent), ddepth, kx, ky, anchor, delta, border, roi=roi).cpu().numpy()
                assert np.array_equal(bits(got), bits(orc.orc_sepFilter2D(parent, ddepth, kx, ky, anchor, delta, border, roi=roi))), (dtype, roi, border)
    # a batch: frames along grid z
    fr = rnd((5, 130, 270), np.float32, 2)
    got = cv.sepFilter2DBatch(dev(fr), -1, g19, g19).cpu().numpy()
    for i in range(5):
        assert np.array_equal(bits(got[i]), bits(orc.orc_sepFilter2D(fr[i], -1, g19, g19)))


def g71_exact(orc):
    """cv::getGaussianKernel(71, 8.64421, CV_32F) as the reference computes it (bit-exact taps through the library's own entry point)"""
    import opencv_amd
    return np.asarray(opencv_amd.getGaussianKernel(71, 8.64421, opencv_amd.CV_32F)).ravel()
