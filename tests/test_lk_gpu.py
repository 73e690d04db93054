"""GPU parity for the sparse pyramidal Lucas-Kanade tracker: cv_hal_ScharrDeriv and cv_hal_LKOpticalFlowLevel through the
C ABI, BIT-EXACT against the oracle (next points compared as bit patterns, status, error) -- at the hooks' own granularity with host
arrays (the padded rectangles are staged), and as the fully device-resident mirror opencv_amd.calcOpticalFlowPyrLK."""
import numpy as np
import pytest
import torch

from test_oracle_lk import frames, points, same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def dev(a):
    return torch.from_numpy(a).cuda()


def test_scharr_deriv(cv, orc):
    rng = np.random.default_rng(1)
    for (w, h) in [(1, 1), (2, 5), (7, 1), (33, 17), (640, 480)]:
        for cn in (1, 3, 4):
            img = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
            want = orc.orc_ScharrDeriv(img)
            assert np.array_equal(cv.ScharrDeriv(dev(img)).cpu().numpy(), want), (w, h, cn)
            assert np.array_equal(cv.ScharrDeriv(img), want), (w, h, cn, "host")


def test_copy_make_border(cv):
    rng = np.random.default_rng(2)
    for shape, dtype in [((37, 53), np.uint8), ((20, 31, 3), np.uint8), ((16, 9, 6), np.int16)]:
        a = rng.integers(0, 200, shape).astype(dtype)
        pad = ((5, 7), (9, 4)) + (((0, 0),) if a.ndim == 3 else ())
        for bt, mode in ((4, "reflect"), (1, "edge"), (2, "symmetric"), (3, "wrap"), (0, "constant")):
            got = cv.copyMakeBorder(dev(a), 5, 7, 9, 4, bt).cpu().numpy()
            assert np.array_equal(got, np.pad(a, pad, mode=mode)), (shape, bt)
        whole = torch.zeros((shape[0] + 12, shape[1] + 13) + shape[2:], dtype=dev(a).dtype, device="cuda")
        inner = whole[5:5 + shape[0], 9:9 + shape[1]]
        inner[...] = dev(a)
        cv.copyMakeBorder(inner, 5, 7, 9, 4, 4 | 16, dst=whole)                        # in place, as buildOpticalFlowPyramid does
        assert np.array_equal(whole.cpu().numpy(), np.pad(a, pad, mode="reflect"))


@pytest.mark.parametrize("cn", [1, 3])
def test_hooks_with_host_arrays(cv, orc, cn):
    """the oracle's driver with the three kernels replaced by the hooks, all arrays on the host"""
    A, B = frames(240, 320, cn, 3 + cn)
    p = points(240, 320, 300, 5)

    def tracker(I, dI, J, prevPts, nextPts, status, err, winW, winH, maxCount, eps2, getMinEig, minEig):
        cv.LKOpticalFlowLevel(I, dI, J, prevPts, nextPts, status, err, (winW, winH), maxCount, eps2, getMinEig, minEig)

    n0 = cv.call_count("LKOpticalFlowLevel")
    for win in [(21, 21), (9, 11), (5, 7)]:
        got = orc.lk_drive(A, B, p, win, 3, (3, 30, 0.01), 0, 1e-4, None, lambda a: cv.pyrDown(a), lambda a: cv.ScharrDeriv(a), tracker)
        same(got, orc.orc_calcOpticalFlowPyrLK(A, B, p, win, 3), (win, cn))
    assert cv.call_count("LKOpticalFlowLevel") > n0


@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("win", [(21, 21), (15, 15), (8, 8), (31, 13)])
def test_device_resident(cv, orc, cn, win):
    A, B = frames(360, 480, cn, 7 + cn)
    p = points(360, 480, 2000, 9)
    got = cv.calcOpticalFlowPyrLK(dev(A), dev(B), dev(p), None, win, 3)
    got = tuple(g.cpu().numpy() for g in got)
    want = orc.orc_calcOpticalFlowPyrLK(A, B, p, win, 3)
    same(got, want, (win, cn))
    assert 0 < want[1].sum() < len(p)


def test_hook_level_loop_equals_one_call(cv, orc):
    """the mirror assembled from the video module's hooks (padded pyramids built with copyMakeBorder on the device) against the one-call
    entry point and the oracle"""
    A, B = frames(300, 420, 1, 31)
    p = points(300, 420, 500, 3)
    a = tuple(g.cpu().numpy() for g in cv.calcOpticalFlowPyrLK_hooks(dev(A), dev(B), dev(p), None, (15, 15), 3))
    b = tuple(g.cpu().numpy() for g in cv.calcOpticalFlowPyrLK(dev(A), dev(B), dev(p), None, (15, 15), 3))
    want = orc.orc_calcOpticalFlowPyrLK(A, B, p, (15, 15), 3)
    same(a, want, "hooks"); same(b, want, "one call")


def test_large_window_takes_the_thread_per_point_kernel(cv, orc):
    """61x61x3 windows need more LDS than the wave-per-point kernel may use: the one-thread-per-point kernel serves them"""
    A, B = frames(200, 260, 3, 21)
    p = points(200, 260, 150, 4)
    got = cv.calcOpticalFlowPyrLK(dev(A), dev(B), dev(p), None, (61, 61), 1)
    same(tuple(g.cpu().numpy() for g in got), orc.orc_calcOpticalFlowPyrLK(A, B, p, (61, 61), 1), "61x61x3")


def test_flags_and_criteria(cv, orc):
    A, B = frames(200, 260, 1, 9, shift=(5.2, 3.1))
    p = points(200, 260, 250, 6)
    guess = p + np.float32([4.0, 2.5])
    for flags, nextPts in ((0, None), (8, None), (4, guess), (12, guess)):
        for crit in ((3, 30, 0.01), (1, 5, 0.0), (2, 0, 0.3), (3, 0, 0.01), (0, 0, 0.0), (3, 100, 1e-4)):
            got = cv.calcOpticalFlowPyrLK(dev(A), dev(B), dev(p), dev(nextPts) if nextPts is not None else None, (21, 21), 2, crit, flags, 1e-3)
            want = orc.orc_calcOpticalFlowPyrLK(A, B, p, (21, 21), 2, crit, flags, 1e-3, nextPts)
            same(tuple(g.cpu().numpy() for g in got), want, (flags, crit))
    host = cv.calcOpticalFlowPyrLK(A, B, p, None, (15, 15), 2)                          # numpy in, numpy out
    same(host, orc.orc_calcOpticalFlowPyrLK(A, B, p, (15, 15), 2), "host")
