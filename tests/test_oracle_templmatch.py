"""Pins oracle/templmatch.c against the real reference (CPU only).  The reference's correlation comes from float FFTs;
its own accuracy test (test_templmatch.cpp:310-345, matchTemplate_Modes) allows 1e-3 relative against a brute-force
double evaluation -- the oracle IS that brute-force evaluation, so the same bound applies (observed: ~1e-6)."""
import numpy as np
import pytest


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
def test_matchtemplate_modes(orc, ref, dtype, cn, method):
    hi = 256 if dtype == np.uint8 else 1.0
    worst = 0.0
    for (iw, ih, tw, th) in [(64, 48, 8, 8), (97, 61, 17, 9), (40, 40, 40, 40), (130, 33, 5, 30)]:
        img = orc.ref_rng_fill((ih, iw, cn) if cn > 1 else (ih, iw), dtype, 100 + iw, 0, hi)
        tpl = orc.ref_rng_fill((th, tw, cn) if cn > 1 else (th, tw), dtype, 200 + tw, 0, hi)
        want = orc.ref_matchTemplate(img, tpl, method)
        got = orc.orc_matchTemplate(img, tpl, method)
        worst = max(worst, orc.rel_err(got, want))
    assert worst <= 1e-4, worst          # parity contract for CV_32F results (the reference's own bound is 1e-3)


def test_matchtemplate_exact_hit(orc, ref):
    """a template cut out of the image: CCORR_NORMED / CCOEFF_NORMED = 1 and SQDIFF = 0 at its location"""
    img = orc.ref_rng_fill((80, 100), np.uint8, 7, 0, 256)
    tpl = np.ascontiguousarray(img[20:36, 30:62])
    for method, val in [(3, 1.0), (5, 1.0), (0, 0.0), (1, 0.0)]:
        r = orc.orc_matchTemplate(img, tpl, method)
        assert abs(float(r[20, 30]) - val) <= 1e-6
        assert (np.unravel_index(np.argmax(r) if val == 1.0 else np.argmin(r), r.shape)) == (20, 30)
