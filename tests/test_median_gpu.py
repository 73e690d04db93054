"""GPU parity for cv::medianBlur through cv_hal_medianBlur: CV_8U, apertures 3 and 5, 1/3/4 channels, aligned /
ragged / unaligned rows, one to several strips, degenerate heights, host pointers; bit-exact against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


@pytest.mark.parametrize("cn", [1, 3, 4])
@pytest.mark.parametrize("ksize", [3, 5])
def test_median(cv, orc, cn, ksize):
    rng = np.random.default_rng(ksize * 10 + cn)
    n0 = cv.call_count("medianBlur")
    for (w, h) in [(1, 1), (3, 5), (16, 1), (32, 2), (23, 9), (64, 23), (100, 33), (1040, 37), (1027, 18), (2064, 41)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        got = cv.medianBlur(torch.from_numpy(src).cuda(), ksize).cpu().numpy()
        assert np.array_equal(got, orc.orc_medianBlur(src, ksize)), (w, h, cn, ksize)
        if w >= 64:                                   # wide enough for the rolling path whatever the row raggedness (roll.h eligible)
            from opencv_amd import _lib
            assert ("k_median_roll<%d,%d," % (ksize, cn)) in _lib.lib.mi355cv_lastKernel().decode(), (w, h, cn, ksize, _lib.lib.mi355cv_lastKernel().decode())
    # salt-and-pepper on a ramp: the textbook use; constant images stay constant
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 4, (40, 1))
    noisy = ramp.copy(); noisy[rng.random(ramp.shape) < 0.05] = 255; noisy[rng.random(ramp.shape) < 0.05] = 0
    noisy = np.repeat(noisy[:, :, None], cn, axis=2) if cn > 1 else noisy
    assert np.array_equal(cv.medianBlur(noisy, ksize), orc.orc_medianBlur(noisy, ksize))          # host pointers
    flat = np.full((20, 48, cn) if cn > 1 else (20, 48), 77, np.uint8)
    assert (cv.medianBlur(torch.from_numpy(flat).cuda(), ksize).cpu().numpy() == 77).all()
    assert cv.call_count("medianBlur") > n0


def test_median_declines_what_it_does_not_cover(cv):
    from opencv_amd import _lib
    with pytest.raises(NotImplementedError) as e:
        cv.medianBlur(torch.zeros((20, 40), dtype=torch.float32, device="cuda"), 7)          # the reference asserts CV_8U for apertures above 5 (median_blur.simd.hpp:878)
    assert "aperture 7" in str(e.value)
    with pytest.raises(NotImplementedError):
        cv.medianBlur(torch.zeros((20, 40, 2), dtype=torch.uint8, device="cuda"), 7)         # the reference asserts cn 1 / 3 / 4 for its CV_8U histogram forms
    with pytest.raises(NotImplementedError):
        cv.medianBlur(torch.zeros((40, 40), dtype=torch.uint8, device="cuda"), cv.limit("median8u_max_ksize") + 2)


@pytest.mark.parametrize("dtype", [np.uint16, np.int16, np.float32])
@pytest.mark.parametrize("ksize", [3, 5])
def test_median_16bit_and_float(cv, orc, dtype, ksize):
    """medianBlur_SortNet on CV_16U / CV_16S / CV_32F (median_blur.simd.hpp:862-868): declined in round 3, now k_median_typed"""
    from opencv_amd import _lib
    rng = np.random.default_rng(ksize + 7)
    for shape in [(1, 1), (5, 3), (16, 1), (1, 23), (23, 9, 3), (64, 23, 4), (37, 1040), (270, 480, 3), (33, 47, 2)]:
        src = (rng.random(shape) * 3000 - 1000).astype(dtype) if dtype != np.uint16 else rng.integers(0, 65536, shape, dtype=np.uint16)
        got = cv.medianBlur(torch.from_numpy(src).cuda(), ksize).cpu().numpy()
        assert np.array_equal(got, orc.orc_medianBlur(src, ksize)), (dtype, shape, ksize)
        assert "k_median_typed" in _lib.lib.mi355cv_lastKernel().decode()
    src = (rng.random((40, 64)) * 100).astype(dtype)
    assert np.array_equal(cv.medianBlur(src, ksize), orc.orc_medianBlur(src, ksize))          # host pointers


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_median_large_apertures_u8(cv, orc, cn):
    """CV_8U apertures 7 .. 31 (the reference's histogram forms medianBlur_8u_Om / _O1, median_blur.simd.hpp:84, :348; Imgproc_MedianBlur.hires_regression_13409 uses 9):
    k_median_bits_u8, bit-exact -- incl. images smaller than the aperture, ragged tile edges and a constant image"""
    from opencv_amd import _lib
    rng = np.random.default_rng(cn)
    for (w, h, k) in [(40, 23, 7), (130, 37, 9), (65, 9, 11), (5, 4, 7), (1, 30, 9), (200, 70, 15), (97, 50, 21), (70, 66, 31)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        got = cv.medianBlur(torch.from_numpy(src).cuda(), k).cpu().numpy()
        assert np.array_equal(got, orc.orc_medianBlur(src, k)), (w, h, cn, k)
        assert "k_median_bits_u8" in _lib.lib.mi355cv_lastKernel().decode()
    flat = np.full((30, 50, cn) if cn > 1 else (30, 50), 201, np.uint8)
    assert (cv.medianBlur(torch.from_numpy(flat).cuda(), 9).cpu().numpy() == 201).all()
    two = rng.integers(0, 256, (33, 47, 2), dtype=np.uint8)                                   # 2 channels: the sort-network apertures take any channel count
    for k in (3, 5):
        assert np.array_equal(cv.medianBlur(torch.from_numpy(two).cuda(), k).cpu().numpy(), orc.orc_medianBlur(two, k)), k
    big = rng.integers(0, 256, (540, 960, cn) if cn > 1 else (540, 960), dtype=np.uint8)
    assert np.array_equal(cv.medianBlur(big, 9), orc.orc_medianBlur(big, 9))                   # host pointers, many tiles
