"""Pins oracle/color_lab.c (CV_8U L*a*b*, both directions) to the reference build in oracle/_ref: EVERY 8-bit colour -- all 2^24 BGR triples through
BGR2Lab / LRGB2Lab and all 2^24 Lab triples through Lab2BGR / Lab2LRGB -- plus the remaining codes and the 4-channel forms on random images.
The tables of the restatement (softfloat arithmetic restated as IEEE float / double, Turkowski's cube root, the C library's pow) are thereby
checked entry by entry wherever an entry is reachable."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.skipif(orc.load_ref() is None, reason="oracle/_ref not built")


def all_colours():
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.empty((4096, 4096, 3), np.uint8)
    img[..., 0] = (v & 255).reshape(4096, 4096)
    img[..., 1] = ((v >> 8) & 255).reshape(4096, 4096)
    img[..., 2] = (v >> 16).reshape(4096, 4096)
    return img


@pytest.mark.parametrize("code", [44, 75, 56, 79, 50, 51, 58, 81])            # L*a*b* both ways, then L*u*v*: BGR2Luv, RGB2Luv, Luv2BGR, Luv2LRGB
def test_every_8bit_colour(code):
    img = all_colours()
    want = orc.ref_cvtColor(img, code, 3)
    got = orc.orc_cvtColorLab(img, code)
    bad = np.flatnonzero((want != got).any(axis=2).ravel())
    assert bad.size == 0, (code, bad.size, img.reshape(-1, 3)[bad[:5]], want.reshape(-1, 3)[bad[:5]], got.reshape(-1, 3)[bad[:5]])


@pytest.mark.parametrize("code", [44, 45, 74, 75, 50, 51])
@pytest.mark.parametrize("scn", [3, 4])
def test_forward_codes_and_channels(code, scn):
    rng = np.random.default_rng(code * 10 + scn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641)]:
        img = rng.integers(0, 256, (h, w, scn), dtype=np.uint8)
        assert np.array_equal(orc.orc_cvtColorLab(img, code), orc.ref_cvtColor(img, code, 3)), (code, scn, h, w)


@pytest.mark.parametrize("code", [56, 57, 78, 79, 58, 59, 80, 81])
@pytest.mark.parametrize("dcn", [3, 4])
def test_inverse_codes_and_channels(code, dcn):
    rng = np.random.default_rng(code * 10 + dcn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(orc.orc_cvtColorLab(img, code, dcn), orc.ref_cvtColor(img, code, dcn)), (code, dcn, h, w)


def test_library_tables_equal_the_oracle_tables():
    """the host-built tables of libmi355cv (csrc/color_lab.hip; no GPU involved) against the oracle's, which the tests above pin to the reference"""
    import ctypes
    from opencv_amd import _lib
    o = orc.oracle()
    o.orc_labTable.argtypes = [ctypes.c_int, ctypes.c_void_p]
    for which, n in [(0, 256), (1, 3072), (2, 4096)]:
        a, b = np.zeros(n, np.uint16), np.zeros(n, np.uint16)
        assert _lib.lib.mi355cv_labTable(which, a.ctypes.data) == n and o.orc_labTable(which, b.ctypes.data) == n
        assert np.array_equal(a, b), which
    yf, pairs = np.zeros(256, np.uint32), np.zeros(512, np.uint16)
    assert _lib.lib.mi355cv_labTable(3, yf.ctypes.data) == 256 and o.orc_labTable(3, pairs.ctypes.data) == 512
    assert np.array_equal(yf & 0xffff, pairs[0::2]) and np.array_equal(yf >> 16, pairs[1::2])
    # the a/b -> X/Z table the kernel evaluates instead of loading (initLUTforABXZ): linear ramp (C division, truncating) below 3390, cube above
    ab = np.zeros(36864, np.int32)
    assert o.orc_labTable(4, ab.ctypes.data) == 36864
    i = np.arange(-8145, 36864 - 8145, dtype=np.int64)
    ramp = np.sign(i * 108) * (np.abs(i * 108) // 841) - 290
    cube = ((i * i >> 14) * i) >> 14
    assert np.array_equal(ab, np.where(i <= 3390, ramp, cube))
    # L*u*v*: the 33^3 grid (library: (L, u, v, 0) per point) and the two 256 x 256 tables of the inverse
    grid, want = np.zeros(33 ** 3 * 4, np.int16), np.zeros(33 ** 3 * 3, np.int16)
    assert _lib.lib.mi355cv_labTable(4, grid.ctypes.data) == grid.size and o.orc_labTable(5, want.ctypes.data) == want.size
    assert np.array_equal(grid.reshape(-1, 4)[:, :3], want.reshape(-1, 3)) and not grid.reshape(-1, 4)[:, 3].any()
    for mine, theirs in ((5, 6), (6, 7)):
        a, b = np.zeros(65536, np.int32), np.zeros(65536, np.int32)
        assert _lib.lib.mi355cv_labTable(mine, a.ctypes.data) == 65536 and o.orc_labTable(theirs, b.ctypes.data) == 65536
        assert np.array_equal(a, b), mine


def _err(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(1.0, np.abs(b))))


def _float_inputs(rng, h, w, cn):
    img = (rng.random((h, w, cn), dtype=np.float32) * 1.2 - 0.1).astype(np.float32)          # a tenth of the range outside [0, 1]: clipped by the conversion
    img.reshape(-1, cn)[:6, :3] = [[0, 0, 0], [1, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.001, 0.002, 0.0005]]
    return img


@pytest.mark.parametrize("code", [44, 45, 74, 75, 50, 51, 76, 77])
@pytest.mark.parametrize("scn", [3, 4])
def test_float_forward(code, scn):
    """CV_32F L*a*b* and L*u*v*: identical to the reference, vector bodies and scalar row tails (the last width % 8 pixels) alike"""
    rng = np.random.default_rng(code + scn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641), (50, 1032)]:
        img = _float_inputs(rng, h, w, scn) if h * w > 6 else rng.random((h, w, scn), dtype=np.float32)
        got, want = orc.orc_cvtColorLab(img, code), orc.ref_cvtColor(img, code, 3)
        assert np.array_equal(got, want), (code, scn, h, w, _err(got, want))


def _lab_or_luv(rng, h, w, code):
    lab = np.empty((h, w, 3), np.float32)
    lab[..., 0] = rng.random((h, w)) * 100
    if code in (56, 57, 78, 79):
        lab[..., 1:] = rng.random((h, w, 2)) * 254 - 127
    else:
        lab[..., 1] = rng.random((h, w)) * 354 - 134
        lab[..., 2] = rng.random((h, w)) * 262 - 140
    return lab


@pytest.mark.parametrize("code", [56, 57, 78, 79, 58, 59, 80, 81])
@pytest.mark.parametrize("dcn", [3, 4])
def test_float_inverse(code, dcn):
    rng = np.random.default_rng(code + dcn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641)]:
        lab = _lab_or_luv(rng, h, w, code)
        got, want = orc.orc_cvtColorLab(lab, code, dcn), orc.ref_cvtColor(lab, code, dcn)
        assert np.array_equal(got, want), (code, dcn, h, w, _err(got, want))


def test_float_tables_equal_the_oracle_tables():
    import ctypes
    from opencv_amd import _lib
    o = orc.oracle()
    o.orc_labTable.argtypes = [ctypes.c_int, ctypes.c_void_p]
    grid, want = np.zeros(33 ** 3 * 4, np.int16), np.zeros(33 ** 3 * 3, np.int16)
    assert _lib.lib.mi355cv_labTable(7, grid.ctypes.data) == grid.size and o.orc_labTable(8, want.ctypes.data) == want.size
    assert np.array_equal(grid.reshape(-1, 4)[:, :3], want.reshape(-1, 3))
    for mine, theirs in ((8, 9), (9, 10), (10, 11)):
        a, b = np.zeros(4096, np.float32), np.zeros(4096, np.float32)
        assert _lib.lib.mi355cv_labTable(mine, a.ctypes.data) == 4096 and o.orc_labTable(theirs, b.ctypes.data) == 4096
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), mine


@pytest.mark.parametrize("code", [76, 77])
@pytest.mark.parametrize("scn", [3, 4])
def test_luv_from_linear_rgb_8u(code, scn):
    """CV_8U L*u*v* from LINEAR RGB goes through the float conversion (RGB2Luv_b's float branch): bytes / 255 in, scaled and rounded out"""
    rng = np.random.default_rng(code + scn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641), (33, 1032)]:
        img = rng.integers(0, 256, (h, w, scn), dtype=np.uint8)
        assert np.array_equal(orc.orc_cvtColorLab(img, code), orc.ref_cvtColor(img, code, 3)), (code, scn, h, w)
    # every colour once
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.empty((4096, 4096, 3), np.uint8)
    img[..., 0] = (v & 255).reshape(4096, 4096); img[..., 1] = ((v >> 8) & 255).reshape(4096, 4096); img[..., 2] = (v >> 16).reshape(4096, 4096)
    if scn == 3:
        assert np.array_equal(orc.orc_cvtColorLab(img, code), orc.ref_cvtColor(img, code, 3)), code
