"""Pinning of the CV_8U YUV / YCrCb / NV12 / NV21 restatement (oracle/color_yuv.c) against the real reference: widths that
exercise the reference's SIMD body and its scalar tail, 3- and 4-channel sources / destinations, every code of the family, and
the saturating corners of the colour cube."""
import numpy as np
import pytest

import orc as O


def _img(h, w, cn, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
    corners = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [16, 128, 128], [235, 240, 16], [1, 254, 3]], np.uint8)
    n = min(len(corners), w)
    a[0, :n, :3] = corners[:n]
    return a


@pytest.mark.ref
def test_yuv_family_matches_reference(ref):
    for (w, h) in [(1, 1), (7, 3), (64, 5), (263, 9), (1030, 2)]:
        for code, (scn, _, _) in O._YUV_FWD.items():
            for cn in (3, 4):
                src = _img(h, w, cn, code + cn)
                assert np.array_equal(O.orc_cvtColorYUV(src, code), O.ref_cvtColorYUV(src, code)), (w, h, code, cn)
        for code in O._YUV_INV:
            src = _img(h, w, 3, code)
            assert np.array_equal(O.orc_cvtColorYUV(src, code), O.ref_cvtColorYUV(src, code)), (w, h, code)
        for code in O._HSV:                                      # BGR/RGB -> HSV, 180 and 256 hue ranges
            for cn in (3, 4):
                src = _img(h, w, cn, code + cn)
                assert np.array_equal(O.orc_cvtColorYUV(src, code), O.ref_cvtColorYUV(src, code)), (w, h, code, cn)
    for (w, h) in [(2, 2), (6, 4), (64, 8), (262, 6), (1030, 4)]:
        rng = np.random.default_rng(w)
        src = rng.integers(0, 256, (h * 3 // 2, w), dtype=np.uint8)
        src[0, :2] = [0, 255]; src[h, :2] = [255, 0]
        for code in O._YUV_NV:
            assert np.array_equal(O.orc_cvtColorYUV(src, code), O.ref_cvtColorYUV(src, code)), (w, h, code)
        for code in O._YUV_3P:                                   # I420 / YV12; heights with h % 4 == 2 shift the second chroma plane by half a row
            assert np.array_equal(O.orc_cvtColorYUV(src, code), O.ref_cvtColorYUV(src, code)), (w, h, code)


def test_yuv_known_answers():
    """grey stays grey: BGR (g,g,g) -> Y = g, U = V = 128 and back; NV12 (16,128,128) is black, (235,128,128) is white"""
    g = np.full((2, 4, 3), 100, np.uint8)
    yuv = O.orc_cvtColorYUV(g, 82)
    assert (yuv[..., 0] == 100).all() and (yuv[..., 1:] == 128).all()
    assert np.array_equal(O.orc_cvtColorYUV(yuv, 84), g)
    nv = np.zeros((3, 2), np.uint8); nv[:2] = 16; nv[2] = 128
    assert (O.orc_cvtColorYUV(nv, 91) == 0).all()
    nv[:2] = 235
    assert (O.orc_cvtColorYUV(nv, 91) == 255).all()


def test_yuv_16u_matches_reference():
    """CV_16U members of the YUV / YCrCb family (integer, delta 32768): restated for the next round's kernels, pinned here"""
    import orc as o
    if o.load_ref() is None:
        pytest.skip("oracle/_ref/libocvref.so not built")
    rng = np.random.default_rng(3)
    r = o.load_ref()
    for (w, h) in [(1, 1), (7, 3), (33, 5), (640, 48)]:
        for scn in (3, 4):
            src = rng.integers(0, 65536, (h, w, scn), dtype=np.uint16)
            for code, (swap, cb) in {82: (0, 0), 83: (1, 0), 36: (0, 1), 37: (1, 1)}.items():
                want = np.empty((h, w, 3), np.uint16); got = np.empty((h, w, 3), np.uint16)
                assert r.ref_cvtColorSz(o.P(src), o.step(src), w, h, o.cvtype(src), o.P(want), o.step(want), w, h, o.cvtype(want), code) == 0
                o.oracle().orc_cvtBGRtoYUV16u(o.P(src), o.step(src), o.P(got), o.step(got), w, h, scn, swap, cb)
                assert np.array_equal(got, want), (w, h, scn, code)
        src = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
        for code, (swap, cb) in {84: (0, 0), 85: (1, 0), 38: (0, 1), 39: (1, 1)}.items():
            for dcn in (3, 4):
                want = np.empty((h, w, dcn), np.uint16); got = np.empty((h, w, dcn), np.uint16)
                assert r.ref_cvtColorSz(o.P(src), o.step(src), w, h, o.cvtype(src), o.P(want), o.step(want), w, h, o.cvtype(want), code) == 0
                o.oracle().orc_cvtYUVtoBGR16u(o.P(src), o.step(src), o.P(got), o.step(got), w, h, dcn, swap, cb)
                assert np.array_equal(got, want), (w, h, dcn, code)


def test_yuv_32f_forward_matches_reference():
    """CV_32F BGR -> YUV / YCrCb: the vector body and the (compiler-contracted) scalar tail of the reference build, bit for bit"""
    import orc as o
    if o.load_ref() is None:
        pytest.skip("oracle/_ref/libocvref.so not built")
    rng = np.random.default_rng(3)
    r = o.load_ref()
    for (w, h) in [(1, 1), (7, 3), (33, 5), (643, 48)]:
        for scn in (3, 4):
            src = rng.random((h, w, scn), dtype=np.float32)
            for code, (swap, cb) in {82: (0, 0), 83: (1, 0), 36: (0, 1), 37: (1, 1)}.items():
                want = np.empty((h, w, 3), np.float32); got = np.empty((h, w, 3), np.float32)
                assert r.ref_cvtColorSz(o.P(src), o.step(src), w, h, o.cvtype(src), o.P(want), o.step(want), w, h, o.cvtype(want), code) == 0
                o.oracle().orc_cvtBGRtoYUV32f(o.P(src), o.step(src), o.P(got), o.step(got), w, h, scn, swap, cb)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, scn, code)


def test_yuv_32f_inverse_matches_reference():
    """CV_32F YUV / YCrCb -> BGR(A): bit for bit"""
    import orc as o
    if o.load_ref() is None:
        pytest.skip("oracle/_ref/libocvref.so not built")
    rng = np.random.default_rng(4)
    r = o.load_ref()
    for (w, h) in [(1, 1), (7, 3), (33, 5), (643, 48)]:
        src = rng.random((h, w, 3), dtype=np.float32)
        for code in (84, 85, 38, 39):
            for dcn in (3, 4):
                want = np.empty((h, w, dcn), np.float32)
                assert r.ref_cvtColorSz(o.P(src), o.step(src), w, h, o.cvtype(src), o.P(want), o.step(want), w, h, o.cvtype(want), code) == 0
                got = o.orc_cvtColorYUVwide(src, code, dcn)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, dcn, code)
