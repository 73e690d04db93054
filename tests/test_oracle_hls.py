"""Pinning of oracle/color_hls.c against the real reference: BGR/RGB(A) <-> HLS for CV_8U -- EXHAUSTIVELY, all 2^24 inputs through the reference's vector body and
through its scalar tail (the two differ in a few thousand ties; which pixels of a row take which is part of the restatement) -- and for CV_32F, BGR/RGB(A) <-> HSV for
CV_32F (north_star's 1e-4 for CV_32F; the reference's own vector body and scalar tail differ from each other by the fusing of one multiply-add)."""
import numpy as np
import pytest

import orc as O

pytestmark = pytest.mark.ref


def all_colours():
    c = np.arange(1 << 24, dtype=np.uint32)
    return np.stack([c & 255, (c >> 8) & 255, (c >> 16) & 255], axis=-1).astype(np.uint8)


@pytest.mark.parametrize("code", [52, 69, 60, 73])
def test_hls_8u_exhaustive(ref, code):
    allc = all_colours()
    body = allc.reshape(4096, 4096, 3)                       # rows of 16 blocks of 256 pixels: every pixel in the vector body
    assert np.array_equal(O.orc_cvtColorHxx(body, code), O.ref_cvtColor(body, code, 3))
    n = (1 << 24) // 7 * 7
    tail = allc[:n].reshape(-1, 7, 3)                        # rows of 7 pixels: every pixel in the scalar tail
    assert np.array_equal(O.orc_cvtColorHxx(tail, code), O.ref_cvtColor(tail, code, 3))
    # the split is real: modelling body pixels with the tail's arithmetic is NOT the reference (forward direction)
    if code in (52, 69):
        assert (O.orc_cvtColorHxx(tail, code).reshape(-1, 3) != O.ref_cvtColor(body, code, 3).reshape(-1, 3)[:n]).sum() > 1000


@pytest.mark.parametrize("code", [52, 53, 68, 69, 60, 61, 72, 73])
def test_hls_8u_geometries(ref, code):
    rng = np.random.default_rng(code)
    fwd = code in O._HLS_FWD
    for (w, h, cn) in [(256, 9, 3), (263, 7, 3), (1000, 5, 4), (7, 5, 3), (40, 3, 4), (519, 4, 4), (1, 1, 3), (255, 3, 3)]:
        src = rng.integers(0, 256, (h, w, cn if fwd else 3), dtype=np.uint8)
        dcn = 3 if fwd else cn
        assert np.array_equal(O.orc_cvtColorHxx(src, code, dcn), O.ref_cvtColor(src, code, dcn)), (w, h, cn)


@pytest.mark.parametrize("code", [52, 53, 40, 41, 60, 61, 54, 55])
def test_hls_hsv_32f(ref, code):
    """CV_32F: BGR/RGB(A) -> HLS / HSV on colours in [0, 1] incl. grays and saturated primaries, and back from (h in [0, 360), l / s / v in [0, 1])"""
    rng = np.random.default_rng(code)
    fwd = code in O._HLS_FWD or code in O._HSV
    for (w, h, cn) in [(263, 31, 3), (64, 5, 4), (7, 3, 3)]:
        if fwd:
            src = rng.random((h, w, cn), dtype=np.float32)
            src[0, :5] = 0.25; src[1, :3] = (1, 0, 0) + ((0.5,) if cn == 4 else ()); src[2, :2] = 0
            dcn = 3
        else:
            src = rng.random((h, w, 3), dtype=np.float32); src[..., 0] *= 359.9
            src[0, :4, 2 if code in (60, 61) else 1] = 0                                          # zero saturation
            dcn = cn
        want = O.ref_cvtColor(src, code, dcn)
        got = O.orc_cvtColorHxx(src, code, dcn)
        assert got.shape == want.shape and O.rel_err(got, want) <= 1e-5, (code, w, h, cn, O.rel_err(got, want))
        assert np.abs(got - want).max() <= 2e-4 * max(1.0, float(np.abs(want).max())), (code, np.abs(got - want).max())
