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
    with pytest.raises(NotImplementedError):
        cv.medianBlur(torch.zeros((20, 40), dtype=torch.uint8, device="cuda"), 7)
    with pytest.raises(NotImplementedError):
        cv.medianBlur(torch.zeros((20, 40), dtype=torch.float32, device="cuda"), 3)
