"""GPU parity for row a1 (cv::GaussianBlur CV_8U) -- through the C ABI, against the oracle.

Mirrors GaussianBlur_Bitexact.Linear8U (test_smooth_bitexact.cpp:139-173): same sizes, channel
counts, kernels and the five isolated border modes; bit-exact (max |diff| == 0) is the bar.
"""
import numpy as np
import pytest
import torch

import refpatterns as rp
from test_oracle_smooth import LINEAR8U, V_U8

pytestmark = pytest.mark.gpu

BORDERS = [0, 1, 2, 3, 4]


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def _dev(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("border", BORDERS)
@pytest.mark.parametrize("case", range(len(LINEAR8U)))
def test_linear8u_matrix_device_and_host(cv, orc, case, border):
    cn, (w, h), kx, ky = LINEAR8U[case]
    big = rp.smooth_bitexact_pattern(h + 20, w + 20, cn)
    roi = np.ascontiguousarray(big[10:10 + h, 10:10 + w])
    want = orc.orc_sepSmoothFixedU8(roi, kx, ky, border)
    got_dev = cv.sepSmoothFixedU8(_dev(roi), kx, ky, border | cv.BORDER_ISOLATED).cpu().numpy()
    assert np.array_equal(got_dev, want)
    got_host = cv.sepSmoothFixedU8(roi, kx, ky, border | cv.BORDER_ISOLATED)
    assert isinstance(got_host, np.ndarray) and np.array_equal(got_host, want)


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
@pytest.mark.parametrize("ksize", [3, 5])
def test_gaussianblur_api_fast_and_generic_paths(cv, orc, cn, ksize):
    rng = np.random.default_rng(1234 + cn * 10 + ksize)
    n0 = cv.call_count("gaussianBlurBinomial")
    ncalls = 0
    # widths with w*cn % 16 == 0 take the rolling kernel, the others the generic one
    for (w, h) in [(16, 1), (16, 2), (32, 3), (64, 7), (1024, 33), (1040, 50), (2048, 16), (4112, 9),
                   (3, 3), (5, 4), (17, 31), (100, 37), (1, 9), (9, 1), (333, 5)]:
        shape = (h, w, cn) if cn > 1 else (h, w)
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        for border in BORDERS:
            kw = ksize if (w > 1 or border == 0) else 1
            kh = ksize if (h > 1 or border == 0) else 1
            want = orc.orc_sepSmoothFixedU8(src, V_U8[kw], V_U8[kh], border)
            got = cv.GaussianBlur(_dev(src), (ksize, ksize), 0, 0, border).cpu().numpy()
            assert np.array_equal(got, want), (w, h, cn, ksize, border)
            if kw == kh == ksize:
                ncalls += 1
    assert cv.call_count("gaussianBlurBinomial") - n0 == ncalls      # the GPU hook really ran


@pytest.mark.parametrize("ksize", [7, 9])
def test_gaussianblur_wide_binomial(cv, orc, ksize):
    rng = np.random.default_rng(ksize)
    src = rng.integers(0, 256, (45, 77, 3), dtype=np.uint8)
    for border in BORDERS:
        want = orc.orc_gaussianBlurBinomialU8(src, ksize, border)
        got = cv.GaussianBlur(_dev(src), ksize, 0, 0, border).cpu().numpy()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("cn", [1, 3, 4])
@pytest.mark.parametrize("ksize", [3, 5, 7, 9])
def test_gaussianblur_any_sigma_rolling_path(cv, orc, cn, ksize):
    """Q8.8 taps from sigma (and the 7/9 binomial tables) on geometries the rolling separable kernel takes
    ((W*cn) % 16 == 0): taps from the oracle's bit-exact generator, result from the oracle's fixed-point filter."""
    rng = np.random.default_rng(77 + ksize * 10 + cn)
    for (w, h) in [(16, 1), (32, 2), (48, 5), (64, 23), (1040, 37), (2064, 70)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        for sigma in (0.0, 0.8, 1.5, 3.1):
            if sigma == 0.0 and ksize <= 5:
                continue                                   # the binomial kernel of smooth.hip: covered above
            k = [int(v) for v in orc.orc_getGaussianKernelQ(ksize, sigma)]
            for border in (0, 1, 2, 4):
                want = orc.orc_sepSmoothFixedU8(src, k, k, border)
                got = cv.GaussianBlur(_dev(src), (ksize, ksize), sigma, sigma, border).cpu().numpy()
                assert np.array_equal(got, want), (w, h, cn, ksize, sigma, border)
    # extremes: all-255 input must stay 255 (no overflow of the 16-bit lanes), asymmetric taps through the explicit entry
    src = np.full((20, 64, cn) if cn > 1 else (20, 64), 255, np.uint8)
    assert (cv.GaussianBlur(_dev(src), (ksize, ksize), 1.1).cpu().numpy() == 255).all()
    kx = list(range(1, ksize + 1)); kx[-1] += 256 - sum(kx)
    ky = kx[::-1]
    src = rng.integers(0, 256, (33, 96, cn) if cn > 1 else (33, 96), dtype=np.uint8)
    for border in (0, 1, 2, 4):
        assert np.array_equal(cv.sepSmoothFixedU8(_dev(src), kx, ky, border).cpu().numpy(), orc.orc_sepSmoothFixedU8(src, kx, ky, border)), border


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_gaussian_long_kernels(cv, orc, cn):
    """cv::GaussianBlur on CV_8U with 11 .. 129 taps (sigma 1.7 .. 21: beyond the register-rolling kernels' 9 taps) on the matrix-core kernel (sepmx.hip; the LDS-ring kernel's
    Q8.8 mode, seplong.hip, where a row of taps spans more than five 32-byte K steps), up to mi355cv_limit("gauss8u_max_ksize") and refused above it; the restatement
    is pinned to the reference at these lengths in tests/test_oracle_smooth.py"""
    from opencv_amd import _lib
    rng = np.random.default_rng(5 + cn)
    top = cv.limit("gauss8u_max_ksize")
    for (w, h) in [(1000, 300), (53, 37), (64, 1)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        for (kw, kh, sigma) in [(19, 19, 3.0), (11, 27, 2.0), (33, 33, 5.5), (61, 25, 10.0), (top, 65, 21.0)]:
            if kw > 33 and w == 1000 and cn > 1:
                continue
            kx = [int(v) for v in orc.orc_getGaussianKernelQ(kw, sigma)]; ky = [int(v) for v in orc.orc_getGaussianKernelQ(kh, sigma)]
            for border in (0, 1, 2, 4):
                got = cv.GaussianBlur(_dev(src), (kw, kh), sigma, sigma, border).cpu().numpy()
                k = _lib.lib.mi355cv_lastKernel().decode()
                assert ("k_sepmx<" in k) if (kw - 1) * cn <= 113 and h > 1 else ("k_sepmx<" in k or "k_seplong<3," in k), k      # (one row: ky = [256], not an int8 tap)
                assert np.array_equal(got, orc.orc_sepSmoothFixedU8(src, kx, ky, border)), (w, h, cn, kw, kh, sigma, border)
    src = np.full((40, 200, cn) if cn > 1 else (40, 200), 255, np.uint8)
    assert (cv.GaussianBlur(_dev(src), (65, 65), 11.0).cpu().numpy() == 255).all()
    with pytest.raises(NotImplementedError):
        cv.GaussianBlur(_dev(src), (top + 2, top + 2), 25.0)
    # CV_32F / CV_16U at the float bound (the separable hook's): the hook serves what it declined beyond 33 taps until round 6
    if cn == 1:
        f = rng.uniform(0, 1, (200, 333)).astype(np.float32)
        n = cv.limit("gauss_float_max_ksize")
        k = np.asarray(cv.getGaussianKernel(n, 21.0, cv.CV_32F)).ravel()
        got = cv.GaussianBlur(_dev(f), (n, n), 21.0).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), orc.orc_sepFilter2D(f, -1, k, k).view(np.uint32))


def test_non_isolated_margins(cv, orc):
    """ROI inside a larger image: borders read the real neighbours (hal margins contract)."""
    rng = np.random.default_rng(7)
    parent = rng.integers(0, 256, (60, 90, 3), dtype=np.uint8)
    for (x0, y0, w, h) in [(5, 4, 40, 30), (0, 0, 32, 20), (1, 58, 80, 2), (88, 0, 2, 60)]:
        margins = (x0, y0, 90 - x0 - w, 60 - y0 - h)
        roi = parent[y0:y0 + h, x0:x0 + w]
        for border in (1, 2, 4):
            want = orc.orc_sepSmoothFixedU8(roi, V_U8[5], V_U8[5], border, margins)
            got = cv.sepSmoothFixedU8(_dev(parent)[y0:y0 + h, x0:x0 + w], V_U8[5], V_U8[5], border, margins=margins)
            assert np.array_equal(got.cpu().numpy(), want), (x0, y0, border)
            got_h = cv.sepSmoothFixedU8(roi, V_U8[5], V_U8[5], border, margins=margins)
            assert np.array_equal(got_h, want)


@pytest.mark.parametrize("shape,ksize", [((1080, 1920, 3), 5), ((2160, 3840), 5), ((2160, 3840), 3), ((1080, 1920, 4), 3)])
def test_full_size_frames_bit_exact(cv, orc, shape, ksize):
    """BASELINE configs: 1080p 8UC3 (cfg1) and 4K 8UC1 (headline), default border REFLECT_101."""
    rng = np.random.default_rng(809564)
    src = rng.integers(0, 256, shape, dtype=np.uint8)
    want = orc.orc_gaussianBlurBinomialU8(src, ksize, 4)
    got = cv.GaussianBlur(_dev(src), ksize, 0).cpu().numpy()
    assert np.array_equal(got, want)
    # in-place call clones the source first (smooth.dispatch.cpp:685-686)
    d = _dev(src)
    cv.GaussianBlur(d, ksize, 0, dst=d)
    assert np.array_equal(d.cpu().numpy(), want)


def test_batch_entry_equals_per_frame(cv, orc):
    rng = np.random.default_rng(99)
    frames = rng.integers(0, 256, (5, 270, 480, 3), dtype=np.uint8)
    d = _dev(frames)
    out = cv.GaussianBlurBatch(d, 5).cpu().numpy()
    for f in range(frames.shape[0]):
        assert np.array_equal(out[f], orc.orc_gaussianBlurBinomialU8(frames[f], 5, 4)), f
    # 8K frame through the batch entry with 2 frames: size-independent property -- a constant image
    # stays constant, and identical frames give identical results
    big = torch.full((2, 4320, 7680), 77, dtype=torch.uint8, device="cuda")
    big[:, 1000:1100, 2000:2100] = 200
    o = cv.GaussianBlurBatch(big, 5)
    assert torch.equal(o[0], o[1])
    assert int(o[0, 0, 0]) == 77 and int(o[0, 4319, 7679]) == 77
    patch = big[0, 990:1110, 1990:2110].cpu().numpy()
    assert np.array_equal(o[0, 992:1108, 1992:2108].cpu().numpy(), orc.orc_gaussianBlurBinomialU8(patch, 5, 4)[2:-2, 2:-2])
