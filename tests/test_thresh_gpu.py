"""GPU parity for cv::threshold through cv_hal_threshold: every depth, every fixed-level type, thresholds inside /
at / beyond the range, vectorised (16-byte) and ragged rows, in place, host pointers; bit-exact."""
import numpy as np
import pytest
import torch

from test_oracle_thresh import CASES, _src

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


@pytest.mark.parametrize("dtype,threshes,maxvals", CASES)
def test_threshold(cv, orc, dtype, threshes, maxvals):
    n0 = cv.call_count("threshold")
    for shape in [(37, 61, 3), (16, 64), (5, 1), (1, 3), (33, 1024, 4)]:
        src = _src(dtype, 3, shape)
        d = torch.from_numpy(src).cuda()
        for t in threshes:
            for m in maxvals[:2]:
                for ttype in range(5):
                    rv_want, want = orc.orc_threshold(src, t, m, ttype)
                    rv, got = cv.threshold(d, t, m, ttype)
                    assert rv == rv_want and np.array_equal(got.cpu().numpy(), want), (dtype, shape, t, m, ttype)
    src = _src(dtype, 4, (40, 80))
    rv_want, want = orc.orc_threshold(src, threshes[2], maxvals[0], 3)
    rv, got = cv.threshold(src, threshes[2], maxvals[0], 3)                      # host pointers
    assert isinstance(got, np.ndarray) and np.array_equal(got, want)
    d = torch.from_numpy(src).cuda()
    cv.threshold(d, threshes[2], maxvals[0], 3, dst=d)                           # in place, as gftt uses it
    assert np.array_equal(d.cpu().numpy(), want)
    assert cv.call_count("threshold") > n0


def test_threshold_4k_bandwidth_shape(cv, orc):
    src = np.random.default_rng(5).integers(0, 256, (2160, 3840), dtype=np.uint8)
    rv, got = cv.threshold(torch.from_numpy(src).cuda(), 127, 255, 0)
    assert np.array_equal(got.cpu().numpy(), orc.orc_threshold(src, 127, 255, 0)[1])


def test_adaptive_threshold(cv, orc):
    rng = np.random.default_rng(21)
    for shape in [(37, 61), (64, 64), (5, 9), (1, 20), (480, 640)]:
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        for bs in (3, 5, 7, 11, 15):
            for ttype in (0, 1):
                for C in (0.0, 2.0, -3.5):
                    want = orc.orc_adaptiveThreshold(src, 255.0, ttype, bs, C)
                    got = cv.adaptiveThreshold(torch.from_numpy(src).cuda(), 255.0, cv.ADAPTIVE_THRESH_MEAN_C, ttype, bs, C).cpu().numpy()
                    assert np.array_equal(got, want), (shape, bs, ttype, C)
    src = rng.integers(0, 256, (50, 70), dtype=np.uint8)
    assert np.array_equal(cv.adaptiveThreshold(src, 200.0, 0, 0, 5, 3.0), orc.orc_adaptiveThreshold(src, 200.0, 0, 5, 3.0))      # host pointers
    d = torch.from_numpy(src).cuda()
    cv.adaptiveThreshold(d, 200.0, 0, 1, 7, 1.0, dst=d)                                                                       # in place
    assert np.array_equal(d.cpu().numpy(), orc.orc_adaptiveThreshold(src, 200.0, 1, 7, 1.0))
    # ADAPTIVE_THRESH_GAUSSIAN_C (float blur + cvRound, thresh.cpp:1720-1727) and MEAN_C blocks beyond 15 x 15 (int32 box sums)
    for shape in [(37, 61), (5, 9), (1, 20), (480, 640)]:
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        for bs in (3, 5, 7, 11, 21, 33):
            for ttype in (0, 1):
                for C in (0.0, -3.5):
                    want = orc.orc_adaptiveThreshold(src, 255.0, ttype, bs, C, method=1)
                    got = cv.adaptiveThreshold(torch.from_numpy(src).cuda(), 255.0, cv.ADAPTIVE_THRESH_GAUSSIAN_C, ttype, bs, C).cpu().numpy()
                    assert np.array_equal(got, want), ("gaussian", shape, bs, ttype, C)
        for bs in (17, 31, 51):
            want = orc.orc_adaptiveThreshold(src, 200.0, 0, bs, 1.5)
            assert np.array_equal(cv.adaptiveThreshold(torch.from_numpy(src).cuda(), 200.0, cv.ADAPTIVE_THRESH_MEAN_C, 0, bs, 1.5).cpu().numpy(), want), ("mean", shape, bs)
    assert np.array_equal(cv.adaptiveThreshold(src, 255.0, cv.ADAPTIVE_THRESH_GAUSSIAN_C, 0, 7, 2.0), orc.orc_adaptiveThreshold(src, 255.0, 0, 7, 2.0, method=1))   # host pointers
    # blocks of 35 .. 129: the float blur runs on the LDS-ring separable kernel (seplong.hip); restatement pinned to thresh.cpp:1692-1727 in tests/test_oracle_thresh.py
    top = cv.limit("adaptive_gaussian_max_block")
    assert top == 129                                             # a wider bound needs parity cases up to it (tests/test_declines_cpu.py holds the same number)
    for shape in [(150, 333), (37, 61), (5, 9)]:
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        for bs in (35, 65, 101, top):
            for ttype, C in ((0, 0.0), (1, -3.5)):
                want = orc.orc_adaptiveThreshold(src, 255.0, ttype, bs, C, method=1)
                got = cv.adaptiveThreshold(torch.from_numpy(src).cuda(), 255.0, cv.ADAPTIVE_THRESH_GAUSSIAN_C, ttype, bs, C).cpu().numpy()
                assert np.array_equal(got, want), ("gaussian", shape, bs, ttype, C)
    with pytest.raises(NotImplementedError):                      # beyond the separable hook's tap bound: refused, never a CPU path
        cv.adaptiveThreshold(torch.from_numpy(src).cuda(), 255.0, cv.ADAPTIVE_THRESH_GAUSSIAN_C, 0, top + 2, 0.0)
    for bs in (101, cv.limit("adaptive_mean_max_block")):         # MEAN_C up to its own bound
        src = rng.integers(0, 256, (150, 333), dtype=np.uint8)
        assert np.array_equal(cv.adaptiveThreshold(torch.from_numpy(src).cuda(), 200.0, cv.ADAPTIVE_THRESH_MEAN_C, 0, bs, 1.5).cpu().numpy(), orc.orc_adaptiveThreshold(src, 200.0, 0, bs, 1.5)), bs
    with pytest.raises(NotImplementedError):
        cv.adaptiveThreshold(torch.from_numpy(src).cuda(), 200.0, cv.ADAPTIVE_THRESH_MEAN_C, 0, cv.limit("adaptive_mean_max_block") + 2, 1.5)


def test_gaussian_c_float_blur_bits(cv, orc):
    """the CV_32F separable blur ADAPTIVE_THRESH_GAUSSIAN_C runs on the GPU gives the oracle's floats bit for bit on 8-bit valued images (the oracle gives
    the reference's, tests/test_oracle_thresh.py): the rounded mean cannot differ by a tie"""
    rng = np.random.default_rng(7)
    src = rng.integers(0, 256, (257, 333), dtype=np.uint8).astype(np.float32)
    for bs in (3, 5, 7, 11, 21, 33, 65, 129):
        k = cv.getGaussianKernel(bs, 0.0, 5)
        got = cv.sepFilter2D(torch.from_numpy(src).cuda(), -1, k, k, borderType=1 | 16).cpu().numpy()
        want = orc.orc_sepFilter2D(src, 5, k, k, border=1)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), bs
