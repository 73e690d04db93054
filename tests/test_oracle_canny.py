"""Pinning of the cv::Canny restatement (oracle/canny.c) against the real reference (which runs its striped, multi-threaded
implementation): structured and noisy images, 1 and 3 channels, L1 / L2 magnitude, apertures 3 / 5, swapped thresholds."""
import numpy as np
import pytest

import orc as O


def scene(h, w, cn, seed):
    """rectangles, a disc and a ramp on noise, blurred a little: long connected edges (hysteresis) plus texture"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = (xx * 255 // max(w - 1, 1)).astype(np.float64) * 0.3
    img[h // 5: h // 2, w // 6: w // 2] += 120
    img[(yy - h * 0.65) ** 2 + (xx - w * 0.7) ** 2 < (min(h, w) * 0.2) ** 2] += 90
    img += rng.normal(0, 6, (h, w))
    img = np.clip(img, 0, 255).astype(np.uint8)
    if cn > 1:
        img = np.stack([np.roll(img, 3 * c, axis=1) if c else img for c in range(cn)], axis=2)
        img = np.ascontiguousarray(img)
    return O.ref_GaussianBlur(img, 3, 0, 0, 4)


@pytest.mark.ref
@pytest.mark.parametrize("cn", [1, 3])
def test_canny_matches_reference(ref, cn):
    for (w, h) in [(160, 120), (97, 143), (33, 9), (640, 200)]:
        img = scene(h, w, cn, w + cn)
        for t1, t2, ap, L2 in [(50, 150, 3, False), (50, 150, 3, True), (20, 60, 3, False), (150, 50, 3, False), (400, 900, 5, False), (300, 700, 5, True),
                               (0, 0, 3, False), (1000, 2000, 3, False)]:
            want = O.ref_Canny(img, t1, t2, ap, L2)
            got = O.orc_Canny(img, t1, t2, ap, L2)
            assert np.array_equal(got, want), (w, h, cn, t1, t2, ap, L2, int((got != want).sum()))
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (60, 80, cn) if cn > 1 else (60, 80), dtype=np.uint8)
    assert np.array_equal(O.orc_Canny(noise, 100, 200), O.ref_Canny(noise, 100, 200))
