"""GPU parity for cv::bilateralFilter (§8 f1, cv_hal_bilateralFilter) through the C ABI against the oracle, which is pinned bit for bit to the real
reference (tests/test_oracle_thresh.py::test_bilateral_filter_matches_reference)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


@pytest.mark.parametrize("cn", [1, 3])
def test_bilateral_filter(cv, orc, cn):
    rng = np.random.default_rng(11 + cn)
    for (w, h) in [(97, 33), (64, 20), (41, 17), (130, 9), (333, 200)]:
        shape = (h, w, cn) if cn > 1 else (h, w)
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        sm = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256).astype(np.uint8)
        noisy = np.clip((np.repeat(sm[..., None], cn, -1) if cn > 1 else sm).astype(int) + rng.integers(-6, 7, shape), 0, 255).astype(np.uint8)
        for img in (src, noisy):
            for d, sc, ss in [(5, 25.0, 3.0), (9, 75.0, 75.0), (0, 30.0, 2.0), (3, 10.0, 1.0), (15, 40.0, 4.0)]:
                for border in (4, 1, 0, 2):
                    got = cv.bilateralFilter(torch.from_numpy(img).cuda(), d, sc, ss, border).cpu().numpy()
                    assert np.array_equal(got, orc.orc_bilateralFilter(img, d, sc, ss, border)), (w, h, d, sc, ss, border)
    img = rng.integers(0, 256, (50, 70, cn) if cn > 1 else (50, 70), dtype=np.uint8)
    assert np.array_equal(cv.bilateralFilter(img, 7, 50.0, 3.0), orc.orc_bilateralFilter(img, 7, 50.0, 3.0))            # host arrays
    with pytest.raises(NotImplementedError):
        cv.bilateralFilter(torch.from_numpy(img).cuda(), cv.limit("bilateral_max_d") + 2, 50.0, 3.0)                    # radius beyond the LDS tile (mi355cv_limit): declined


@pytest.mark.parametrize("cn", [1, 3])
def test_bilateral_filter_32f(cv, orc, cn):
    """CV_32FC1 / CV_32FC3 (declined): k_minmax_f32 + host-built table + k_bilateral_f32, within 1e-6 of the restatement (itself 1e-6 from the reference),
    every border, radii up to 16, NaN pixels, a constant image, host arrays"""
    from opencv_amd import _lib
    rng = np.random.default_rng(21 + cn)
    for (w, h) in [(97, 33), (64, 20), (41, 17), (333, 100)]:
        src = (rng.random((h, w, cn) if cn > 1 else (h, w), dtype=np.float32) * 3 - 1).astype(np.float32)
        for d, sc, ss in [(5, 0.3, 2.0), (9, 1.5, 7.0), (0, 0.2, 2.0), (3, 10.0, 1.0), (33, 0.5, 6.0)]:
            for border in (4, 1, 0, 2):
                got = cv.bilateralFilter(torch.from_numpy(src).cuda(), d, sc, ss, border).cpu().numpy()
                want = orc.orc_bilateralFilter(src, d, sc, ss, border)
                assert orc.rel_err(got, want) <= 1e-6 and np.abs(got - want).max() <= 3e-6, (cn, w, h, d, sc, ss, border, orc.rel_err(got, want))
        assert "k_bilateral_f32" in _lib.lib.mi355cv_lastKernel().decode()
    src = rng.random((30, 40, cn) if cn > 1 else (30, 40), dtype=np.float32); src[5, 7] = np.nan; src[10:12, 20] = np.nan
    got = cv.bilateralFilter(torch.from_numpy(src).cuda(), 5, 0.3, 2.0).cpu().numpy()
    assert not np.isnan(got).any() and np.abs(got - orc.orc_bilateralFilter(src, 5, 0.3, 2.0)).max() <= 3e-6
    flat = np.full((20, 30, cn) if cn > 1 else (20, 30), -2.5, np.float32)
    assert np.array_equal(cv.bilateralFilter(torch.from_numpy(flat).cuda(), 5, 0.3, 2.0).cpu().numpy(), flat)
    img = rng.random((50, 70, cn) if cn > 1 else (50, 70), dtype=np.float32)
    assert np.abs(cv.bilateralFilter(img, 7, 0.4, 3.0) - orc.orc_bilateralFilter(img, 7, 0.4, 3.0)).max() <= 3e-6       # host arrays


def test_bilateral_32f_values_the_colour_table_cannot_span_are_declined(cv, orc):
    """the CV_32F colour table covers the image's own [min, max] only and is indexed unchecked (bilateral_filter.simd.hpp:679).  The zeros of a BORDER_CONSTANT halo
    with 0 outside that range, and +-Inf pixels, would index far beyond it (a fault on the GPU): those calls are declined, everything in range is still served."""
    rng = np.random.default_rng(5)
    src = (1000 + 10 * rng.random((40, 60), dtype=np.float32)).astype(np.float32)
    dev = torch.from_numpy(src).cuda()
    with pytest.raises(NotImplementedError):
        cv.bilateralFilter(dev, 5, 0.3, 2.0, 0)                                   # BORDER_CONSTANT, 0 not in [1000, 1010]
    got = cv.bilateralFilter(dev, 5, 0.3, 2.0, 4).cpu().numpy()                   # any replicated border: served
    assert np.abs(got - orc.orc_bilateralFilter(src, 5, 0.3, 2.0, 4)).max() <= 3e-3
    neg = (src - 1005).astype(np.float32)                                        # 0 inside the range: BORDER_CONSTANT served
    got = cv.bilateralFilter(torch.from_numpy(neg).cuda(), 5, 0.3, 2.0, 0).cpu().numpy()
    assert np.abs(got - orc.orc_bilateralFilter(neg, 5, 0.3, 2.0, 0)).max() <= 3e-5
    inf = neg.copy(); inf[7, 9] = np.inf
    with pytest.raises(NotImplementedError):
        cv.bilateralFilter(torch.from_numpy(inf).cuda(), 5, 0.3, 2.0, 4)


def test_bilateral_row_range_is_filtered_as_an_image_of_its_own(cv, orc):
    """cv_hal_bilateralFilter carries no margins (hal_replacement.hpp:1068): a full-width row range of a larger image is dense, so the hook cannot tell it
    from a whole image and filters it with the border rule at its first and last rows, where cv::bilateralFilter's own code path pads with the
    parent's real rows (copyMakeBorder without BORDER_ISOLATED).  The reference's bundled ndsrvp HAL does the same.  This test pins the divergence:
    the range equals the filtered copy of the range, and differs from the range of the filtered parent in its first / last `radius` rows only."""
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (200, 160), dtype=np.uint8)
    d, sc, ss, r = 9, 60.0, 4.0, 4
    dev = torch.from_numpy(img).cuda()
    got = cv.bilateralFilter(dev[60:140], d, sc, ss).cpu().numpy()
    assert np.array_equal(got, orc.orc_bilateralFilter(np.ascontiguousarray(img[60:140]), d, sc, ss))
    whole = orc.orc_bilateralFilter(img, d, sc, ss)[60:140]
    assert np.array_equal(got[r:-r], whole[r:-r]) and not np.array_equal(got[:r], whole[:r])


def test_bilateral_filter_4k_timing_shape(cv, orc):
    """a 4K frame: the result of a crop that depends only on the crop's neighbourhood equals the oracle on that neighbourhood"""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (2160, 3840), dtype=np.uint8)
    got = cv.bilateralFilter(torch.from_numpy(img).cuda(), 9, 75.0, 75.0)
    crop = np.ascontiguousarray(img[:120, :264])
    want = orc.orc_bilateralFilter(crop, 9, 75.0, 75.0)
    assert np.array_equal(got[:100, :256].cpu().numpy(), want[:100, :256])       # (columns < 256 are vector-body columns in both)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16])
def test_image_moments(cv, orc, dtype):
    """cv_hal_imageMoments: the ten spatial moments equal the oracle's (pinned to cv::moments) as doubles, bit for bit"""
    rng = np.random.default_rng(8)
    info = np.iinfo(dtype)
    for (w, h) in [(1, 1), (31, 5), (32, 32), (33, 65), (200, 97), (3840, 2160), (1000, 37)]:
        for src in (rng.integers(info.min, int(info.max) + 1, (h, w), dtype=dtype), np.full((h, w), info.max, dtype=dtype)):
            for binary in (False, True):
                got = cv.moments(torch.from_numpy(src).cuda(), binary)
                want = orc.orc_moments(src, binary)
                assert [got[k] for k in ("m00", "m10", "m01", "m20", "m11", "m02", "m30", "m21", "m12", "m03")] == want.tolist(), (w, h, binary)
    src = rng.integers(0, 256, (70, 90)).astype(dtype)
    assert list(cv.moments(src).values()) == orc.orc_moments(src).tolist()                      # host arrays


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_image_moments_float(cv, orc, dtype):
    """CV_32F / CV_64F: every sum inside a 32 x 32 tile is a chain of double additions in raster order in the reference (momentsInTile<float / double, double,
    double>); the kernel walks the chains in that order (a lane per tile row, then the rows one after the other), so the ten doubles are equal bit for bit"""
    rng = np.random.default_rng(9)
    for (w, h) in [(1, 1), (31, 5), (32, 32), (33, 65), (200, 97), (3840, 2160), (1000, 37), (64, 1)]:
        for src in ((rng.random((h, w)) * 255 - 40).astype(dtype), np.full((h, w), 0.1, dtype=dtype), (rng.random((h, w)) < 0.3).astype(dtype)):
            for binary in (False, True):
                got = cv.moments(torch.from_numpy(src).cuda(), binary)
                want = orc.orc_moments(src, binary)
                assert [got[k] for k in ("m00", "m10", "m01", "m20", "m11", "m02", "m30", "m21", "m12", "m03")] == want.tolist(), (w, h, binary)
    src = (rng.random((70, 90)) * 3).astype(dtype)
    assert list(cv.moments(src).values()) == orc.orc_moments(src).tolist()                      # host arrays
