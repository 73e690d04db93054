"""GPU parity for k_sepmx (opencv_amd/csrc/sepmx.hip): cv::GaussianBlur on CV_8U beyond 9 taps with both passes on the matrix cores (v_mfma_i32_32x32x32_i8: the row
pass against a Toeplitz matrix of kx, the column pass against one of ky on the two byte planes of the 16-bit row sums).  Integer arithmetic, so the bar is bit for bit
against the restatement of fixedSmoothInvoker (smooth.simd.hpp:1926) that tests/test_oracle_smooth.py pins to the reference at these lengths: every border rule, 1-4
channels, ragged widths around the 256-byte strips and the 32-row tiles, segment seams of tall images, ROI windows with real pixels around them (unaligned row starts: the
staging shift delta), unequal kernels per axis, batches, and the hand-over to the vector kernel where the matrix form does not apply (a tap above 127, more than five K steps)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def _dev(a):
    return torch.from_numpy(a).cuda()


def last_kernel():
    from opencv_amd import _lib
    return _lib.lib.mi355cv_lastKernel().decode()


def taps(orc, n, sigma):
    return [int(v) for v in orc.orc_getGaussianKernelQ(n, sigma)]


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_sepmx_shapes_and_borders(cv, orc, cn):
    rng = np.random.default_rng(40 + cn)
    for (w, h) in [(700, 75), (256 // cn, 32), (257, 33), (31, 5), (1, 40), (40, 1), (513, 64), (90, 300)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        for (kw, kh, sigma) in [(19, 19, 3.0), (11, 33, 2.5), (33, 13, 5.0)]:
            kx, ky = taps(orc, kw, sigma), taps(orc, kh, sigma)
            for border in (0, 1, 2, 3, 4):
                got = cv.sepSmoothFixedU8(_dev(src), kx, ky, border).cpu().numpy()
                k = last_kernel()
                if (kw - 1) * cn <= 128 - 15 and border != 3 and w >= kw:
                    assert "k_sepmx<" in k, k                                     # (BORDER_WRAP on a wide image, reflections in an image narrower than the kernel: k_seplong)
                assert np.array_equal(got, orc.orc_sepSmoothFixedU8(src, kx, ky, border)), (w, h, cn, kw, kh, border, k)


def test_sepmx_long_kernels_and_extremes(cv, orc):
    rng = np.random.default_rng(50)
    src = rng.integers(0, 256, (200, 600), dtype=np.uint8)
    src[:40] = 255; src[40:80] = 0                                  # saturated and empty bands: the row sums' extremes (65280, 0) cross both byte planes
    for (kw, kh, sigma) in [(65, 65, 11.0), (97, 41, 16.0), (129, 129, 21.0), (129, 11, 30.0), (15, 129, 40.0)]:
        kx, ky = taps(orc, kw, sigma), taps(orc, kh, sigma)
        for border in (0, 1, 4):
            got = cv.sepSmoothFixedU8(_dev(src), kx, ky, border).cpu().numpy()
            assert "k_sepmx<" in last_kernel(), last_kernel()
            assert np.array_equal(got, orc.orc_sepSmoothFixedU8(src, kx, ky, border)), (kw, kh, border, last_kernel())
    # every byte value against the largest taps the matrix form takes (127 + 2 + 127 = 256), and the hand-over above them (an identity row of taps: 256)
    ramp = np.tile(np.arange(256, dtype=np.uint8), (64, 3))
    ky = [3] + [10] * 25 + [3]
    for kx, mx in (([0] * 4 + [127, 2, 127] + [0] * 4, True), ([0] * 5 + [256] + [0] * 5, False), ([0] * 4 + [64, 128, 64] + [0] * 4, False)):
        for border in (1, 4):
            got = cv.sepSmoothFixedU8(_dev(ramp), kx, ky, border).cpu().numpy()
            assert ("k_sepmx<" in last_kernel()) == mx, last_kernel()
            assert np.array_equal(got, orc.orc_sepSmoothFixedU8(ramp, kx, ky, border)), (kx, border, last_kernel())


def test_sepmx_tall_images_cross_segment_seams(cv, orc):
    rng = np.random.default_rng(51)
    for (w, h, cn) in [(300, 1000, 1), (100, 777, 3)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        for (kw, kh, sigma) in [(19, 19, 3.0), (21, 67, 11.0)]:
            kx, ky = taps(orc, kw, sigma), taps(orc, kh, sigma)
            got = cv.sepSmoothFixedU8(_dev(src), kx, ky, 4).cpu().numpy()
            assert "k_sepmx<" in last_kernel() and "seg=" in last_kernel()
            assert int(last_kernel().split("seg=")[1]) < h, last_kernel()
            assert np.array_equal(got, orc.orc_sepSmoothFixedU8(src, kx, ky, 4)), (w, h, cn, kw, kh)


def test_sepmx_roi_with_margins_and_unaligned_rows(cv, orc):
    """windows into a larger image: borders read the real neighbours, the row start is at an arbitrary byte (the staged block shifts by delta so that its loads stay aligned)"""
    rng = np.random.default_rng(52)
    for cn in (1, 3):
        parent = rng.integers(0, 256, (120, 640, cn) if cn > 1 else (120, 640), dtype=np.uint8)
        kx, ky = taps(orc, 19, 3.0), taps(orc, 25, 4.0)
        deltas = set()
        for (x0, y0, w, h) in [(5, 4, 400, 90), (0, 0, 320, 20), (1, 100, 600, 20), (630, 0, 10, 120), (37, 11, 513, 66), (16, 0, 512, 120)]:
            margins = (x0, y0, 640 - x0 - w, 120 - y0 - h)
            roi = parent[y0:y0 + h, x0:x0 + w]
            for border in (0, 1, 2, 4):
                want = orc.orc_sepSmoothFixedU8(roi, kx, ky, border, margins)
                got = cv.sepSmoothFixedU8(_dev(parent)[y0:y0 + h, x0:x0 + w], kx, ky, border, margins=margins)
                assert "k_sepmx<" in last_kernel(), last_kernel()
                deltas.add(int(last_kernel().split("delta=")[1].split()[0]))
                assert np.array_equal(got.cpu().numpy(), want), (cn, x0, y0, w, h, border, last_kernel())
        assert len(deltas) > 2, deltas


@pytest.mark.parametrize("cn", [2, 3, 4])
def test_sepmx_colour_images_with_long_kernels(cv, orc, cn):
    """channels are interleaved elements, so a row of taps spans (nx - 1) * cn + 32 bytes: up to thirteen K steps of the row pass (three channels x 129 taps, four x 97)"""
    rng = np.random.default_rng(80 + cn)
    src = rng.integers(0, 256, (150, 700, cn), dtype=np.uint8)
    seen = set()
    for (kw, kh, sigma) in [(45, 45, 7.5), (65, 33, 11.0), (97, 19, 16.0), (129, 129, 21.0)]:
        if (kw - 1) * cn + 32 > 416:
            continue
        kx, ky = taps(orc, kw, sigma), taps(orc, kh, sigma)
        for border in (0, 1, 4):
            got = cv.sepSmoothFixedU8(_dev(src), kx, ky, border).cpu().numpy()
            assert "k_sepmx<" in last_kernel(), last_kernel()
            seen.add(last_kernel().split("<")[1].split(",")[0])
            assert np.array_equal(got, orc.orc_sepSmoothFixedU8(src, kx, ky, border)), (cn, kw, kh, border, last_kernel())
    assert seen & {"7", "9", "13"}, seen
    got = cv.boxFilter(_dev(src), -1, (51, 51)).cpu().numpy()
    assert "k_sepmx<" in last_kernel(), last_kernel()
    assert np.array_equal(got, orc.orc_boxFilter(src, -1, (51, 51)))


def test_sepmx_batch_equals_frames(cv, orc):
    rng = np.random.default_rng(53)
    frames = rng.integers(0, 256, (5, 130, 517), dtype=np.uint8)
    got = cv.GaussianBlurBatch(_dev(frames), (19, 19), sigmaX=3.0).cpu().numpy()
    assert "k_sepmx<" in last_kernel(), last_kernel()
    kx = taps(orc, 19, 3.0)
    for i in range(5):
        assert np.array_equal(got[i], orc.orc_sepSmoothFixedU8(frames[i], kx, kx, 4)), i


def test_sepmx_handover(cv, orc, monkeypatch):
    """what the matrix form does not take stays on the vector kernel (k_seplong's Q8.8 mode) with the same bits: 4 channels x 129 taps (17 K steps: more than thirteen)"""
    rng = np.random.default_rng(54)
    src = rng.integers(0, 256, (70, 300, 4), dtype=np.uint8)
    kx = taps(orc, 129, 21.0)
    got = cv.sepSmoothFixedU8(_dev(src), kx, kx, 4).cpu().numpy()
    assert "k_seplong<3," in last_kernel(), last_kernel()
    assert np.array_equal(got, orc.orc_sepSmoothFixedU8(src, kx, kx, 4))


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_sepmx_box_filter(cv, orc, cn):
    """cv::boxFilter / cv::blur on CV_8U beyond the 7 x 7 of the rolling kernels: the window sum on the matrix cores (taps of 1), finished with the reference's own
    normalisations -- ColumnSum<ushort, uchar>'s reciprocal pair for areas <= 256, ColumnSum<int, uchar>'s float multiply with its double row tail beyond, plain
    saturation un-normalised (box_filter.simd.hpp:340-455) -- bit for bit against the restatement, with odd anchors, every border rule, ragged widths and ROI windows"""
    rng = np.random.default_rng(60 + cn)
    for (w, h) in [(317, 70), (256 // cn, 33), (45, 130)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        for (ks, anchor, norm) in [((9, 9), (-1, -1), True), ((15, 15), (-1, -1), True), ((16, 16), (-1, -1), True), ((31, 31), (-1, -1), True), ((11, 5), (2, 4), True),
                                   ((3, 25), (-1, 20), True), ((21, 21), (-1, -1), False), ((129, 129), (-1, -1), True), ((9, 2), (8, 0), False)]:
            if ks[0] > w or ks[1] > h:
                continue
            for border in (0, 1, 2, 4):
                got = cv.boxFilter(_dev(src), -1, ks, anchor, norm, border).cpu().numpy()
                k = last_kernel()
                # (what the matrix-core kernel declines -- a border rule that piles more than 254 unit weights on a rim pixel of a narrow image -- is the two-pass box filter's)
                assert ("k_sepmx<" in k and ",box>" in k) or "k_box_rows" in k, (k, ks)
                assert np.array_equal(got, orc.orc_boxFilter(src, -1, ks, anchor, norm, border)), (w, h, cn, ks, anchor, norm, border, k)
    # up to cv::boxFilter's largest window on one channel (nine K steps per pass); more channels x 255 taps do not fit the row pass and stay where they were
    if cn in (1, 3):                                                   # (the restatement walks kw * kh taps per output: one big image is enough)
        big = rng.integers(0, 256, (262, 270, cn) if cn > 1 else (262, 270), dtype=np.uint8)
        for ks, border in (((255, 255), 1), ((201, 31), 4), ((131, 255), 4)):
            got = cv.boxFilter(_dev(big), -1, ks, (-1, -1), True, border).cpu().numpy()
            if cn == 1:
                assert "k_sepmx<" in last_kernel(), (last_kernel(), ks)
            assert np.array_equal(got, orc.orc_boxFilter(big, -1, ks, (-1, -1), True, border)), (cn, ks, border, last_kernel())
    # a window into a larger image, and batches
    parent = rng.integers(0, 256, (90, 400, cn) if cn > 1 else (90, 400), dtype=np.uint8)
    for roi in [(5, 4, 300, 60), (0, 0, 128, 90), (390, 10, 10, 70)]:
        for border in (1, 4):
            got = cv.boxFilter(_dev(parent), -1, (13, 13), borderType=border, roi=roi).cpu().numpy()
            assert "k_sepmx<" in last_kernel(), last_kernel()
            assert np.array_equal(got, orc.orc_boxFilter(parent, -1, (13, 13), border=border, roi=roi)), (roi, border)
    frames = rng.integers(0, 256, (3, 70, 300, cn) if cn > 1 else (3, 70, 300), dtype=np.uint8)
    got = cv.boxFilterBatch(_dev(frames), -1, (25, 25)).cpu().numpy()
    assert "k_sepmx<" in last_kernel(), last_kernel()
    for i in range(3):
        assert np.array_equal(got[i], orc.orc_boxFilter(frames[i], -1, (25, 25))), i


def test_sepmx_adaptive_threshold_mean_rides_on_it(cv, orc):
    rng = np.random.default_rng(70)
    src = rng.integers(0, 256, (200, 333), dtype=np.uint8)
    for bs in (15, 51, 129, 201, 255):
        got = cv.adaptiveThreshold(_dev(src), 255, 0, 0, bs, 5).cpu().numpy()
        assert np.array_equal(got, orc.orc_adaptiveThreshold(src, 255, 0, bs, 5, 0)), bs


def test_sepmx_random_geometry_and_repeatability(cv, orc):
    """the walk keeps two steps of rows in flight behind counted waits: (1) random sizes / channels / taps / borders / ROI windows against the restatement, (2) a batch of
    full-HD frames filtered 12 times over -- every pass must give the same bytes as the first, and the first the restatement's (a race would show up as an odd pass)"""
    rng = np.random.default_rng(1234)
    for it in range(60):
        cn = int(rng.integers(1, 5))
        w, h = int(rng.integers(8, 700)), int(rng.integers(1, 260))
        kw = int(rng.choice([7, 9, 11, 15, 19, 25, 33])); kh = int(rng.choice([7, 9, 13, 19, 31, 65]))
        if (kw - 1) * cn > 113 or kw > w or kh > 2 * h + 1:
            continue
        sx, sy = max(0.9, kw / 6.0), max(0.9, kh / 6.0)
        kx, ky = taps(orc, kw, sx), taps(orc, kh, sy)
        border = int(rng.choice([0, 1, 2, 4]))
        pw, ph = w + int(rng.integers(0, 40)), h + int(rng.integers(0, 20))
        parent = rng.integers(0, 256, (ph, pw, cn) if cn > 1 else (ph, pw), dtype=np.uint8)
        x0, y0 = int(rng.integers(0, pw - w + 1)), int(rng.integers(0, ph - h + 1))
        margins = (x0, y0, pw - x0 - w, ph - y0 - h)
        roi = parent[y0:y0 + h, x0:x0 + w]
        got = cv.sepSmoothFixedU8(_dev(parent)[y0:y0 + h, x0:x0 + w], kx, ky, border, margins=margins).cpu().numpy()
        assert np.array_equal(got, orc.orc_sepSmoothFixedU8(roi, kx, ky, border, margins)), (it, cn, w, h, kw, kh, border, margins, last_kernel())
    frames = _dev(rng.integers(0, 256, (24, 1080, 1920), dtype=np.uint8))
    first = cv.GaussianBlurBatch(frames, (19, 19), sigmaX=3.0).clone()
    assert "k_sepmx<" in last_kernel()
    kx = taps(orc, 19, 3.0)
    host = frames.cpu().numpy()
    for i in (0, 11, 23):
        assert np.array_equal(first[i].cpu().numpy(), orc.orc_sepSmoothFixedU8(host[i], kx, kx, 4)), i
    out = torch.empty_like(frames)
    for rep in range(12):
        cv.GaussianBlurBatch(frames, (19, 19), sigmaX=3.0, dst=out)
        assert torch.equal(out, first), rep
