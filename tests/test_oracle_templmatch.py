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


def _masks(orc, th, tw, cn, kind, seed):
    """kind: 'u8' binary-after-threshold (values 0 / 1 / 7 / 255), 'u8cn' one mask per channel, 'f32' weights in [0, 1), 'f32cn'"""
    shape = (th, tw, cn) if (kind.endswith("cn") and cn > 1) else (th, tw)
    if kind.startswith("u8"):
        m = orc.ref_rng_fill(shape, np.uint8, seed, 0, 4)
        return np.ascontiguousarray(np.array([0, 1, 7, 255], np.uint8)[m])
    return orc.ref_rng_fill(shape, np.float32, seed, 0, 1.0)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("kind", ["u8", "u8cn", "f32", "f32cn"])
def test_matchtemplate_mask_modes(orc, ref, dtype, cn, kind):
    """matchTemplateMask (templmatch.cpp:762): the restatement (direct double correlations, the float expressions in the reference's order) against the
    reference (float FFTs).  The error is measured against the largest term of the expression, as the reference's own masked test does
    (test_templmatch2.cpp: a naive double evaluation, 'eps' relative to the maximum of the result)"""
    if kind.endswith("cn") and cn == 1:
        pytest.skip("same as the one-channel mask")
    hi = 256 if dtype == np.uint8 else 1.0
    for method in range(6):
        worst = 0.0
        for (iw, ih, tw, th) in [(64, 48, 8, 8), (97, 61, 17, 9), (130, 33, 5, 30)]:
            img = orc.ref_rng_fill((ih, iw, cn) if cn > 1 else (ih, iw), dtype, 300 + iw, 0, hi)
            tpl = orc.ref_rng_fill((th, tw, cn) if cn > 1 else (th, tw), dtype, 400 + tw, 0, hi)
            mask = _masks(orc, th, tw, cn, kind, 500 + tw)
            want = orc.ref_matchTemplateMask(img, tpl, method, mask)
            got = orc.orc_matchTemplateMask(img, tpl, method, mask)
            assert np.isfinite(want).all()
            worst = max(worst, orc.rel_err(got, want))
        assert worst <= 1e-4, (method, worst)


def test_matchtemplate_mask_exact_hit(orc, ref):
    """a template cut out of the image, half of it masked away and the masked half of the TEMPLATE overwritten: the masked scores still peak at its location"""
    img = orc.ref_rng_fill((80, 100), np.uint8, 7, 0, 256)
    tpl = np.ascontiguousarray(img[20:36, 30:62]).copy()
    mask = np.zeros(tpl.shape, np.uint8); mask[:, :16] = 255
    tpl[:, 16:] = 99
    for method, val in [(3, 1.0), (5, 1.0), (0, 0.0), (1, 0.0)]:
        r = orc.orc_matchTemplateMask(img, tpl, method, mask)
        assert abs(float(r[20, 30]) - val) <= 2e-6, (method, r[20, 30])
        assert (np.unravel_index(np.argmax(r) if val == 1.0 else np.argmin(r), r.shape)) == (20, 30)
