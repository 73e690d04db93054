"""Pins oracle/color_misc.c and oracle/hist.c against the real reference (CPU only): 4:2:0 / 4:2:2 encoders, the 4:2:2 decoder, XYZ
(8U / 16U), the 16-bit packed formats, premultiplied alpha, equalizeHist and THRESH_OTSU.  Widths cover the reference's vector bodies
and their scalar tails."""
import numpy as np
import pytest

import orc as o

pytestmark = pytest.mark.skipif(o.load_ref() is None, reason="oracle/_ref/libocvref.so not built")

SIZES = [(2, 2), (6, 4), (34, 6), (70, 10), (130, 4), (642, 482)]


def src_for(code, w, h, rng):
    kind = o.MISC_CODES[code][0]
    k = o.MISC_CODES[code]
    if kind == "to_xyz" or kind == "from_xyz": return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind in ("to_5x5", "enc422", "enc420p"): return rng.integers(0, 256, (h, w, k[1]), dtype=np.uint8)
    if kind in ("from_5x5", "5x5_to_gray", "dec422"): return rng.integers(0, 256, (h, w, 2), dtype=np.uint8)
    if kind == "gray_to_5x5": return rng.integers(0, 256, (h, w), dtype=np.uint8)
    return rng.integers(0, 256, (h, w, 4), dtype=np.uint8)


@pytest.mark.parametrize("code", sorted(o.MISC_CODES))
def test_misc_codes(code):
    rng = np.random.default_rng(code)
    for (w, h) in SIZES:
        src = src_for(code, w, h, rng)
        got, want = o.orc_cvtColorMisc(src, code), o.ref_cvtColorMisc(src, code)
        assert np.array_equal(got, want), (code, w, h, int((got != want).sum()))


@pytest.mark.parametrize("code", [32, 33, 34, 35])
def test_xyz_16u_and_4ch(code):
    rng = np.random.default_rng(code)
    for (w, h) in [(5, 3), (37, 9), (400, 20)]:
        src = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
        assert np.array_equal(o.orc_cvtColorMisc(src, code), o.ref_cvtColorMisc(src, code))
    # extreme values (saturation both ways)
    ex = np.array([[[0, 0, 255], [255, 0, 0], [0, 255, 0], [255, 255, 255], [0, 0, 0], [255, 255, 0], [3, 250, 7]]], np.uint8)
    ex = np.tile(ex, (2, 7, 1))
    assert np.array_equal(o.orc_cvtColorMisc(ex, code), o.ref_cvtColorMisc(ex, code))


@pytest.mark.parametrize("scn", [3, 4])
def test_two_plane_encode(scn):
    rng = np.random.default_rng(scn)
    for (w, h) in [(2, 2), (34, 6), (130, 4), (642, 482)]:
        src = rng.integers(0, 256, (h, w, scn), dtype=np.uint8)
        for swap in (0, 1):
            for uidx in (1, 2):                                  # the reference swaps U and V for uIdx == 2 only (color_yuv.simd.hpp:2120)
                assert np.array_equal(o.orc_cvtBGRtoTwoPlaneYUV(src, swap, uidx), o.ref_cvtBGRtoTwoPlaneYUV(src, swap, uidx)), (w, h, swap, uidx)


@pytest.mark.parametrize("code", [54, 55, 70, 71])
def test_hsv_to_bgr_follows_the_vector_width(code):
    """cv_hal_cvtHSVtoBGR: the reference truncates inside its vector loop and rounds in the scalar tail, so the
    8-bit result depends on the lanes of the build that runs (the hook follows the 8-lane AVX2 build, the widest color_hsv is dispatched for).  The restatement takes that width as a parameter; with 8 lanes (AVX2) it
    equals the oracle/ref build bit for bit, and a 4-lane evaluation differs wherever the two loops split a row differently."""
    rng = np.random.default_rng(code)
    differs = 0
    for (w, h) in [(1, 1), (31, 3), (32, 2), (70, 5), (130, 7), (641, 9)]:
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for dcn in (3, 4):
            want = np.empty((h, w, dcn), np.uint8)
            r = o.load_ref()
            assert r.ref_cvtColorSz(o.P(src), o.step(src), w, h, o.cvtype(src), o.P(want), o.step(want), w, h, o.cvtype(want), code) == 0
            assert np.array_equal(o.orc_cvtHSVtoBGR(src, code, dcn, 8), want), (w, h, dcn)
            differs += int((o.orc_cvtHSVtoBGR(src, code, dcn, 4) != want).sum())
    assert differs > 0


def test_alpha_all_pairs():
    """every (value, alpha) pair, placed in vector bodies and in scalar tails"""
    v, a = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    for w in (256, 251):
        src = np.zeros((256, w, 4), np.uint8)
        src[..., 0] = v[:, :w]; src[..., 1] = v[:, ::-1][:, :w]; src[..., 2] = 255 - v[:, :w]; src[..., 3] = a[:, :w]
        for code in (125, 126):
            assert np.array_equal(o.orc_cvtColorMisc(src, code), o.ref_cvtColorMisc(src, code)), (code, w)


def test_equalize_hist():
    rng = np.random.default_rng(3)
    for (w, h) in [(1, 1), (7, 5), (64, 48), (641, 481)]:
        for lo, hi in [(0, 256), (100, 140), (17, 18), (250, 256)]:
            src = rng.integers(lo, hi, (h, w), dtype=np.uint8)
            assert np.array_equal(o.orc_equalizeHist(src), o.ref_equalizeHist(src)), (w, h, lo, hi)
    grad = (np.add.outer(np.arange(300), np.arange(500)) % 256).astype(np.uint8)
    assert np.array_equal(o.orc_equalizeHist(grad), o.ref_equalizeHist(grad))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_threshold_otsu(dtype):
    rng = np.random.default_rng(11)
    top = 256 if dtype == np.uint8 else 65536
    for (w, h) in [(1, 1), (9, 7), (320, 240), (643, 481)]:
        for mode in range(4):
            if mode == 0: src = rng.integers(0, top, (h, w)).astype(dtype)
            elif mode == 1: src = np.where(rng.random((h, w)) < 0.3, rng.integers(top // 8, top // 4, (h, w)), rng.integers(top // 2, top - 1, (h, w))).astype(dtype)
            elif mode == 2: src = np.full((h, w), top // 3, dtype)
            else: src = np.where(rng.random((h, w)) < 0.5, 0, top - 1).astype(dtype)
            for ttype in range(5):
                for maxval in (255.0, 200.4):
                    tv, td = o.orc_thresholdOtsu(src, maxval, ttype)
                    rv, rd = o.ref_threshold(src, 0.0, maxval, ttype | 8)
                    assert tv == rv and np.array_equal(td, rd), (dtype, w, h, mode, ttype, maxval, tv, rv)


@pytest.mark.parametrize("code", [32, 33, 34, 35])
def test_xyz_32f(code):
    """CV_32F XYZ (155 + 155 declined calls of Imgproc_ColorXYZ.accuracy): the restatement follows the body / tail split of the reference's row loop in the
    SSE baseline build (4 pixels per vector, products and sums rounded one by one; the scalar tail associates the other way) -- bit for bit against oracle/_ref"""
    rng = np.random.default_rng(100 + code)
    for (w, h) in [(1, 1), (3, 2), (4, 3), (5, 3), (37, 9), (400, 20), (1023, 4)]:
        for cn in ((3, 4) if code in (32, 33) else (3,)):
            src = (rng.random((h, w, cn), dtype=np.float32) * 1.5 - 0.25).astype(np.float32)
            got, want = o.orc_cvtColorMisc(src, code), o.ref_cvtColorMisc(src, code)
            assert got.dtype == np.float32 and np.array_equal(got, want), (code, w, h, cn, float(np.abs(got - want).max()))
