"""GPU parity for SURVEY §8 f3 "features2d detectors": FAST 9-of-16 through the C ABI -- the two features2d HAL hooks (dense scores, 3x3
suppression) and the one-call detector -- against the restatement pinned to cv::FAST (tests/test_oracle_fast.py).  Integer work: bit-exact,
keypoint lists equal including their order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def dev(a):
    return torch.from_numpy(a).cuda()


def images():
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    blocks = np.kron(rng.integers(0, 256, (12, 16), dtype=np.uint8), np.ones((9, 9), np.uint8)).astype(np.uint8)
    yy, xx = np.mgrid[0:240, 0:322]
    smooth = ((np.sin(xx / 7.0) + np.cos(yy / 5.0)) * 60 + 128 + rng.integers(-8, 8, xx.shape)).clip(0, 255).astype(np.uint8)
    return [noise, blocks, np.ascontiguousarray(blocks[:, :-3]), smooth, rng.integers(0, 256, (7, 7), dtype=np.uint8), rng.integers(0, 256, (6, 40), dtype=np.uint8),
            rng.integers(0, 256, (5, 300), dtype=np.uint8)]


def test_dense_and_nms_hooks(cv, orc):
    for img in images():
        want = orc.orc_FAST_dense(img, 2)
        got = cv.FAST_dense(dev(img))
        assert np.array_equal(got.cpu().numpy(), want), img.shape
        assert np.array_equal(cv.FAST_dense(img), want)                                     # host image: staged
        wn = orc.orc_FAST_nms(want)
        assert np.array_equal(cv.FAST_NMS(got).cpu().numpy(), wn), img.shape
    with pytest.raises(NotImplementedError):
        cv.FAST_dense(dev(images()[0]), cv.FAST_TYPE_7_12)


@pytest.mark.parametrize("nonmax", [True, False])
def test_fast_keypoints(cv, orc, nonmax):
    n0 = cv.call_count("FAST")
    for img in images():
        for thr in (0, 1, 5, 10, 20, 40, 100):
            want = orc.orc_FAST(img, thr, nonmax, 2)
            got = cv.FAST(dev(img), thr, nonmax)
            assert got.shape == want.shape and np.array_equal(got, want), (img.shape, thr, len(got), len(want))
            if thr <= 20:
                assert np.array_equal(cv.FAST_hooks(dev(img), thr, nonmax), want), (img.shape, thr)
    assert cv.call_count("FAST") > n0


def test_fast_full_frame(cv, orc):
    """1080p frame: tens of thousands of keypoints, order included; the capacity-doubling path of the binding"""
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:1080, 0:1920]
    img = ((np.sin(xx / 9.0) * np.cos(yy / 6.0)) * 90 + 128 + rng.integers(-20, 20, xx.shape)).clip(0, 255).astype(np.uint8)
    for thr, nonmax in ((10, True), (25, False)):
        want = orc.orc_FAST(img, thr, nonmax, 2, cap=2000000)
        got = cv.FAST(dev(img), thr, nonmax)
        assert len(want) > 1000 and got.shape == want.shape and np.array_equal(got, want), (thr, len(got), len(want))
