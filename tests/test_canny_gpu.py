"""GPU parity for cv::Canny through cv_hal_canny: structured scenes with long connected edges (hysteresis across
many tiles), noise, 1 and 3 channels, L1 / L2, apertures 3 / 5, swapped and degenerate thresholds; bit-exact against the oracle."""
import numpy as np
import pytest
import torch

from test_oracle_canny import scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


@pytest.mark.parametrize("cn", [1, 3])
def test_canny(cv, orc, cn):
    n0 = cv.call_count("canny")
    for (w, h) in [(160, 120), (97, 143), (33, 9), (640, 200), (1000, 700)]:
        img = scene(h, w, cn, w + cn)
        for t1, t2, ap, L2 in [(50, 150, 3, False), (50, 150, 3, True), (20, 60, 3, False), (150, 50, 3, False), (400, 900, 5, False), (300, 700, 5, True),
                               (0, 0, 3, False), (1000, 2000, 3, False)]:
            got = cv.Canny(torch.from_numpy(img).cuda(), t1, t2, ap, L2).cpu().numpy()
            want = orc.orc_Canny(img, t1, t2, ap, L2)
            assert np.array_equal(got, want), (w, h, cn, t1, t2, ap, L2, int((got != want).sum()))
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (60, 80, cn) if cn > 1 else (60, 80), dtype=np.uint8)
    assert np.array_equal(cv.Canny(noise, 100, 200), orc.orc_Canny(noise, 100, 200))                 # host pointers
    assert cv.call_count("canny") > n0


def test_canny_long_spiral(cv, orc):
    """one spiral whose brightness fades outwards, so it is a strong edge only near its centre: hysteresis has to carry the edge label through hundreds of tiles"""
    h, w = 600, 800
    img = np.zeros((h, w), np.uint8)
    t = np.linspace(0, 40 * np.pi, 200000)
    r = 8 + t * 2.2
    val = np.clip(255 - t * 6, 30, 255).astype(np.uint8)
    for d in (-1, 0, 1):
        x = (w / 2 + (r + d) * np.cos(t)).round().astype(int); y = (h / 2 + (r + d) * np.sin(t)).round().astype(int)
        ok = (x >= 2) & (x < w - 2) & (y >= 2) & (y < h - 2)
        img[y[ok], x[ok]] = val[ok]
    got = cv.Canny(torch.from_numpy(img).cuda(), 40, 600).cpu().numpy()
    want = orc.orc_Canny(img, 40, 600)
    strong = orc.orc_Canny(img, 600, 600)
    assert np.array_equal(got, want) and (want > 0).sum() > 20 * (strong > 0).sum() > 0
