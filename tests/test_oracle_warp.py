"""Pins oracle/warp.c (resize / warpAffine / warpPerspective / remap) against the real reference (CPU only).
8U / 16U / 16S: bit-exact (test_imgwarp_strict.cpp demands 0 for 8U warpAffine :1089-1092); 32F: 1e-6 relative."""
import numpy as np
import pytest

DT = [np.uint8, np.uint16, np.int16, np.float32]


def rnd(orc, shape, dtype, seed):
    hi = {np.uint8: 256, np.uint16: 65536, np.int16: 32767, np.float32: 1.0}[dtype]
    lo = -32768 if dtype == np.int16 else 0
    return orc.ref_rng_fill(shape, dtype, seed, lo, hi)


def same(orc, got, want, tol=1e-6):
    if want.dtype == np.float32:
        assert orc.rel_err(got, want) <= tol
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_linear_nearest(orc, ref, dtype, cn):
    src = rnd(orc, (37, 53, cn) if cn > 1 else (37, 53), dtype, 5 + cn)
    for dsize in [(80, 55), (35, 25), (53, 37), (106, 74), (17, 9), (1, 1), (200, 3)]:
        for interp in (0, 1):
            same(orc, orc.orc_resize(src, dsize, interpolation=interp), orc.ref_resize(src, dsize, interpolation=interp))
    for fx, fy in [(0.5, 0.5), (1.5, 1.5), (0.75, 1.25), (2.0, 2.0)]:
        same(orc, orc.orc_resize(src, None, fx, fy, 1), orc.ref_resize(src, None, fx, fy, 1))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_resize_area_fast(orc, ref, dtype, cn):
    for (w, h) in [(48, 72), (60, 36), (120, 24)]:
        src = rnd(orc, (h, w, cn) if cn > 1 else (h, w), dtype, 9 + cn + w)
        for s in (2, 3, 4):
            dsize = (w // s, h // s)                       # exact integer scales -> resizeAreaFast_
            same(orc, orc.orc_resize(src, dsize, interpolation=3), orc.ref_resize(src, dsize, interpolation=3))
        same(orc, orc.orc_resize(src, (w // 2, h // 2), interpolation=1), orc.ref_resize(src, (w // 2, h // 2), interpolation=1))
        same(orc, orc.orc_resize(src, (w * 2, h * 2), interpolation=3), orc.ref_resize(src, (w * 2, h * 2), interpolation=3))   # AREA upscale -> linear-like
    # fx = 0.5 on odd sizes whose half rounds UP (cvRound(27.5) = 28): ragged last column/row (resize.cpp:3027-3050)
    src = rnd(orc, (59, 55, cn) if cn > 1 else (59, 55), dtype, 77)
    for interp in (1, 3):
        same(orc, orc.orc_resize(src, None, 0.5, 0.5, interp), orc.ref_resize(src, None, 0.5, 0.5, interp))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_resize_area_general(orc, ref, dtype, cn):
    """true INTER_AREA: shrinking by non-integer ratios, and by an integer ratio on one axis only (resizeArea_, resize.cpp:3181)"""
    for (w, h), dsizes in [((53, 37), [(20, 11), (52, 36), (17, 37), (53, 9)]), ((100, 100), [(33, 33), (99, 51)]), ((64, 48), [(32, 17), (21, 24)])]:
        src = rnd(orc, (h, w, cn) if cn > 1 else (h, w), dtype, 3 + cn + w)
        for dsize in dsizes:
            same(orc, orc.orc_resize(src, dsize, interpolation=3), orc.ref_resize(src, dsize, interpolation=3))
    src = rnd(orc, (45, 77, cn) if cn > 1 else (45, 77), dtype, 8)
    same(orc, orc.orc_resize(src, None, 0.3, 0.7, 3), orc.ref_resize(src, None, 0.3, 0.7, 3))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.uint16, np.int16])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_cubic(orc, ref, dtype, cn):
    """INTER_CUBIC: up- and down-scaling, destination row lengths with and without a SIMD tail (dw*cn % 8, % 4), tiny sources
    whose taps clamp on both sides"""
    for (w, h), dsizes in [((53, 37), [(80, 55), (20, 11), (106, 74), (161, 3)]), ((9, 9), [(31, 29), (8, 8)]), ((5, 3), [(17, 13)]), ((64, 48), [(32, 24), (100, 7)])]:
        src = rnd(orc, (h, w, cn) if cn > 1 else (h, w), dtype, 6 + cn + w)
        for dsize in dsizes:
            got, want = orc.orc_resize(src, dsize, interpolation=2), orc.ref_resize(src, dsize, interpolation=2)
            assert np.array_equal(got, want), (w, h, dsize, dtype, cn)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.uint16, np.int16])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_lanczos4(orc, ref, dtype, cn):
    """INTER_LANCZOS4: 8x8 taps, integer for CV_8U, SIMD body / scalar tail orders for CV_32F; sources smaller than the kernel"""
    for (w, h), dsizes in [((53, 37), [(80, 55), (20, 11), (106, 74), (161, 3)]), ((9, 9), [(31, 29), (8, 8)]), ((5, 3), [(17, 13)]), ((64, 48), [(32, 24), (100, 7)])]:
        src = rnd(orc, (h, w, cn) if cn > 1 else (h, w), dtype, 9 + cn + w)
        for dsize in dsizes:
            got, want = orc.orc_resize(src, dsize, interpolation=4), orc.ref_resize(src, dsize, interpolation=4)
            assert np.array_equal(got, want), (w, h, dsize, dtype, cn)


def mats(orc, w, h):
    out = []
    for ang, sc in [(7.0, 0.95), (33.0, 1.3), (-120.0, 0.6), (0.0, 1.0)]:
        M = orc.ref_getRotationMatrix2D((w / 2.0, h / 2.0), ang, sc)
        out.append(M)
    out.append(np.array([[1, 0, 3.25], [0, 1, -2.5]], np.float64))
    out.append(np.array([[0.3, 0.1, -20.0], [-0.2, 0.4, 30.0]], np.float64))
    return out


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_warp_affine(orc, ref, dtype, cn):
    src = rnd(orc, (45, 61, cn) if cn > 1 else (45, 61), dtype, 15 + cn)
    for M in mats(orc, 61, 45):
        for dsize in [(61, 45), (100, 30)]:
            for interp in (0, 1):
                for border, bval in [(0, 0.0), (0, (10, 200, 30, 77)), (1, 0), (2, 0), (3, 0), (4, 0)]:
                    want = orc.ref_warpAffine(src, M, dsize, interp | 16, border, bval)
                    got = orc.orc_warpAffine(src, M, dsize, interp, border, bval)
                    same(orc, got, want)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_warp_transparent(orc, ref, dtype, cn):
    """BORDER_TRANSPARENT: unmapped pixels keep dst's previous contents; points on the source's last row / column are blended from the
    neighbours they have (remapBilinear imgwarp.cpp:786-815)"""
    src = rnd(orc, (45, 61, cn) if cn > 1 else (45, 61), dtype, 115 + cn)
    Ms = list(mats(orc, 61, 45)) + [np.array([[1.0, 0, 0.25], [0, 1.0, 0.5]]), np.array([[0.5, 0, 30.0], [0, 0.5, 22.0]]), np.array([[1.0, 0, 0], [0, 1.0, 0]])]
    for M in Ms:
        for dsize in [(61, 45), (100, 30)]:
            prev = rnd(orc, (dsize[1], dsize[0], cn) if cn > 1 else (dsize[1], dsize[0]), dtype, 7)
            for interp in (0, 1):
                want = orc.ref_warpAffine(src, M, dsize, interp | 16, 5, 0.0, dst=prev)
                got = orc.orc_warpAffine(src, M, dsize, interp, 5, 0.0, dst=prev)
                same(orc, got, want)
    P = np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]])
    prev = rnd(orc, (45, 61, cn) if cn > 1 else (45, 61), dtype, 8)
    for interp in (0, 1):
        same(orc, orc.orc_warpPerspective(src, P, (61, 45), interp, 5, 0.0, dst=prev), orc.ref_warpPerspective(src, P, (61, 45), interp | 16, 5, 0.0, dst=prev))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("cn", [1, 3])
def test_warp_perspective(orc, ref, dtype, cn):
    src = rnd(orc, (45, 61, cn) if cn > 1 else (45, 61), dtype, 25 + cn)
    Ms = [np.array([[1.1, 0.05, -3.0], [0.02, 0.9, 4.0], [1e-4, -2e-4, 1.0]]),
          np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]]),
          np.eye(3)]
    for M in Ms:
        for dsize in [(61, 45), (150, 40), (7, 70)]:
            for interp in (0, 1):
                for border, bval in [(0, 5.0), (1, 0), (4, 0)]:
                    want = orc.ref_warpPerspective(src, M, dsize, interp | 16, border, bval)
                    got = orc.orc_warpPerspective(src, M, dsize, interp, border, bval)
                    same(orc, got, want)


@pytest.mark.parametrize("dtype", DT)
def test_remap32f(orc, ref, dtype):
    src = rnd(orc, (40, 50, 3), dtype, 35)
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:33, 0:47].astype(np.float32)
    mapx = (xx * 1.07 + rng.uniform(-3, 3, xx.shape)).astype(np.float32)
    mapy = (yy * 1.2 - 2 + rng.uniform(-3, 3, yy.shape)).astype(np.float32)
    for interp in (0, 1):
        for border, bval in [(0, 9.0), (1, 0), (2, 0), (4, 0)]:
            same(orc, orc.orc_remap(src, mapx, mapy, interp, border, bval), orc.ref_remap(src, mapx, mapy, interp, border, bval))


def _float_maps(seed, w=47, h=33, sw=50, sh=40):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    mapx = (xx * (sw / w) + rng.uniform(-3, 3, xx.shape)).astype(np.float32)
    mapy = (yy * (sh / h) - 1 + rng.uniform(-3, 3, yy.shape)).astype(np.float32)
    return mapx, mapy


def test_convert_maps(orc, ref):
    """cv::convertMaps in both directions, every representation (imgwarp.cpp:1925-2260)"""
    mapx, mapy = _float_maps(5)
    mapx[3, 4] = 17.515625; mapy[3, 4] = 8.984375           # exact 1/64 steps: ties of cvRound(x * 32)
    xy = np.ascontiguousarray(np.stack([mapx, mapy], axis=-1))
    for nn in (False, True):
        for m1, m2 in ((mapx, mapy), (xy, None)):
            w1, w2 = orc.ref_convertMaps(m1, m2, "16sc2", nn)
            g1, g2 = orc.orc_convertMaps(m1, m2, "16sc2", nn)
            assert np.array_equal(g1, w1) and (nn or np.array_equal(g2, w2))
    f1, f2 = orc.ref_convertMaps(mapx, mapy, "16sc2", False)
    for dt in ("32fc1", "32fc2"):
        w1, w2 = orc.ref_convertMaps(f1, f2, dt)
        g1, g2 = orc.orc_convertMaps(f1, f2, dt)
        assert np.array_equal(g1, w1) and (w2 is None or np.array_equal(g2, w2))


@pytest.mark.parametrize("dtype", DT)
def test_remap_other_map_types(orc, ref, dtype):
    """cv::remap with a CV_32FC2 map and with the fixed-point maps of convertMaps (bilinear and nearest), incl. BORDER_TRANSPARENT"""
    src = rnd(orc, (40, 50, 3), dtype, 36)
    mapx, mapy = _float_maps(4)
    xy = np.ascontiguousarray(np.stack([mapx, mapy], axis=-1))
    f1, f2 = orc.ref_convertMaps(mapx, mapy, "16sc2", False)
    n1, _ = orc.ref_convertMaps(mapx, mapy, "16sc2", True)
    prev = rnd(orc, (33, 47, 3), dtype, 37)
    for interp in (0, 1):
        for border, bval in [(0, 9.0), (1, 0), (2, 0), (4, 0), (5, 0)]:
            d0 = prev if border == 5 else None
            same(orc, orc.orc_remapMaps(src, xy, None, interp, border, bval, dst=d0), orc.ref_remapMaps(src, xy, None, interp, border, bval, dst=d0))
            same(orc, orc.orc_remapMaps(src, f1, f2, interp, border, bval, dst=d0), orc.ref_remapMaps(src, f1, f2, interp, border, bval, dst=d0))
        same(orc, orc.orc_remapMaps(src, n1, None, 0, 1, 0), orc.ref_remapMaps(src, n1, None, 0, 1, 0))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_polar_forward(orc, ref, dtype):
    """cv::warpPolar, Cartesian -> polar / semilog-polar (imgwarp.cpp:3731-3793), with and without WARP_FILL_OUTLIERS"""
    src = rnd(orc, (60, 80, 3), dtype, 41)
    for flags in (1, 1 | 8, 0 | 8, 1 | 256, 1 | 256 | 8):
        for dsize, center, rad in [((64, 90), (40.0, 30.0), 35.0), ((50, 157), (10.5, 50.25), 60.0)]:
            same(orc, orc.orc_warpPolar(src, dsize, center, rad, flags), orc.ref_warpPolar(src, dsize, center, rad, flags))


def test_log_and_cart_to_polar_rows(orc, ref):
    """the two float approximations cv::warpPolar's inverse map is built from, restated in oracle/warp.c in the form the AVX2 build of core runs them in:
    cv::log over all 2^23 mantissas and random exponents, cv::cartToPolar (radians) over random vectors -- bit patterns compared, for row lengths on
    both sides of the vector widths (8 / 16) and of cartToPolar's 1024-element blocks"""
    import ctypes
    o, r = orc.oracle(), orc.load_ref()
    vp = ctypes.c_void_p
    o.orc_log32fRow.argtypes = [vp, vp, ctypes.c_int]; r.ref_log32f.argtypes = [vp, vp, ctypes.c_int]
    o.orc_cartToPolarRow.argtypes = [vp] * 4 + [ctypes.c_int]; r.ref_cartToPolar32f.argtypes = [vp] * 4 + [ctypes.c_int, ctypes.c_int]
    P = lambda a: a.ctypes.data_as(vp)
    rng = np.random.default_rng(1)
    x = (np.arange(1 << 23, dtype=np.uint32) | (127 << 23)).view(np.float32)
    a, b = np.zeros_like(x), np.zeros_like(x)
    o.orc_log32fRow(P(x), P(a), x.size); assert r.ref_log32f(P(x), P(b), x.size) == 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for n in (1, 3, 7, 8, 9, 15, 16, 17, 1000, 1024, 1030, 1039, 1040, 50000):
        x = np.exp(rng.random(n) * 40 - 20).astype(np.float32)
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        o.orc_log32fRow(P(x), P(a), n); assert r.ref_log32f(P(x), P(b), n) == 0
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("log", n)
        xx = (rng.random(n, dtype=np.float32) * 200 - 100).astype(np.float32); yy = (rng.random(n, dtype=np.float32) * 200 - 100).astype(np.float32)
        xx[: min(n, 3)] = [0.0, -0.0, 5.0][: min(n, 3)]; yy[: min(n, 3)] = [0.0, 3.0, 0.0][: min(n, 3)]
        m1, a1, m2, a2 = (np.zeros(n, np.float32) for _ in range(4))
        o.orc_cartToPolarRow(P(xx), P(yy), P(m1), P(a1), n); assert r.ref_cartToPolar32f(P(xx), P(yy), P(m2), P(a2), n, 0) == 0
        assert np.array_equal(m1.view(np.uint32), m2.view(np.uint32)) and np.array_equal(a1.view(np.uint32), a2.view(np.uint32)), ("cartToPolar", n)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_polar_inverse(orc, ref, dtype):
    """cv::warpPolar with WARP_INVERSE_MAP (imgwarp.cpp:3795-3845), linear and semilog, nearest and bilinear, with WARP_FILL_OUTLIERS (without it the
    reference leaves the outliers of its freshly allocated destination undefined).  Widths on both sides of the vector widths of cartToPolar / log."""
    rng = np.random.default_rng(5)
    for (sh, sw), dsize, center, rad in [((120, 160), (200, 150), (100.3, 74.6), 90.0), ((64, 100), (7, 9), (3.2, 4.1), 5.0),
                                          ((256, 300), (1100, 40), (500.5, 20.2), 600.0), ((90, 77), (130, 131), (64.0, 66.0), 80.0)]:
        for cn in (1, 3):
            shape = (sh, sw) if cn == 1 else (sh, sw, cn)
            src = rng.integers(0, 256, shape, dtype=np.uint8) if dtype == np.uint8 else rng.random(shape, dtype=np.float32)
            for flags in (16 | 1 | 8, 16 | 0 | 8, 16 | 1 | 256 | 8, 16 | 0 | 256 | 8):
                same(orc, orc.orc_warpPolar(src, dsize, center, rad, flags), orc.ref_warpPolar(src, dsize, center, rad, flags))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16])
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_resize_linear_exact(orc, ref, dtype, cn):
    """INTER_LINEAR_EXACT (resize_bitExact, resize.cpp:789-950): up, down, exactly half (the 2x2 area mean, except for 2 channels), 1x1, thin"""
    src = rnd(orc, (37, 53, cn) if cn > 1 else (37, 53), dtype, 7 + cn)
    for dsize in [(80, 60), (20, 11), (53, 37), (106, 74), (26, 18), (27, 19), (1, 1), (200, 5), (7, 90)]:
        assert np.array_equal(orc.orc_resize(src, dsize, interpolation=5), orc.ref_resize(src, dsize, interpolation=5)), (dsize, dtype, cn)
    one = rnd(orc, (1, 9, cn) if cn > 1 else (1, 9), dtype, 3)
    assert np.array_equal(orc.orc_resize(one, (20, 4), interpolation=5), orc.ref_resize(one, (20, 4), interpolation=5))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_resize_nearest_exact(orc, ref, dtype, cn):
    """INTER_NEAREST_EXACT (resizeNN_bitexact, resize.cpp:1174-1289): 16.16 steps, pixel centres; up, down, odd and even sizes, thin images"""
    src = rnd(orc, (37, 53, cn) if cn > 1 else (37, 53), dtype, 11 + cn)
    for dsize in [(80, 60), (20, 11), (53, 37), (106, 74), (26, 18), (27, 19), (1, 1), (200, 5), (7, 90), (159, 111)]:
        assert np.array_equal(orc.orc_resize(src, dsize, interpolation=6), orc.ref_resize(src, dsize, interpolation=6)), (dsize, dtype, cn)
    even = rnd(orc, (36, 52, cn) if cn > 1 else (36, 52), dtype, 5)
    for dsize in [(13, 9), (104, 72), (51, 35)]:
        assert np.array_equal(orc.orc_resize(even, dsize, interpolation=6), orc.ref_resize(even, dsize, interpolation=6)), (dsize, dtype, cn)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_remap_relative_maps(orc, ref, dtype):
    """WARP_RELATIVE_MAP (imgwarp.cpp:1724; Imgproc_RemapRelative.validity in the reference's suite): the maps hold offsets from the destination pixel, added to the
    integer coordinates after their saturation to short -- every map representation, nearest and bilinear, several borders"""
    REL = 32
    src = rnd(orc, (40, 50, 3), dtype, 61)
    rng = np.random.default_rng(8)
    offx = rng.uniform(-6, 6, (33, 47)).astype(np.float32); offy = rng.uniform(-6, 6, (33, 47)).astype(np.float32)
    offx[2, 3] = 40000.0; offy[5, 6] = -40000.0                 # saturate to short BEFORE the pixel's own coordinate is added
    xy = np.ascontiguousarray(np.stack([offx, offy], axis=-1))
    f1, f2 = orc.ref_convertMaps(offx, offy, "16sc2", False)
    n1, _ = orc.ref_convertMaps(offx, offy, "16sc2", True)
    for interp in (0, 1):
        for border, bval in [(0, 9.0), (1, 0), (2, 0), (4, 0)]:
            same(orc, orc.orc_remap(src, offx, offy, interp | REL, border, bval), orc.ref_remap(src, offx, offy, interp | REL, border, bval))
            same(orc, orc.orc_remapMaps(src, xy, None, interp | REL, border, bval), orc.ref_remapMaps(src, xy, None, interp | REL, border, bval))
            same(orc, orc.orc_remapMaps(src, f1, f2, interp | REL, border, bval), orc.ref_remapMaps(src, f1, f2, interp | REL, border, bval))
    same(orc, orc.orc_remapMaps(src, n1, None, 0 | REL, 1, 0), orc.ref_remapMaps(src, n1, None, 0 | REL, 1, 0))
    # an identity in relative form is the image itself
    z = np.zeros((40, 50), np.float32)
    assert np.array_equal(orc.orc_remap(src, z, z, 1 | REL, 1, 0), src)


def test_area_fast_2x2_float_summation_orders(orc, ref):
    """INTER_AREA by exactly 2 x 2 on CV_32F: the reference's vector body sums pairwise, (s00 + s01) + (s10 + s11) -- 1 channel below the last multiple of its 4
    lanes, every 4-channel pixel --, the scalar loop behind it (the 1-channel tail, 2 and 3 channels) in order, ((s00 + s01) + s10) + s11.  Data of wide dynamic range
    makes the two orders differ; the restatement equals the reference BIT FOR BIT"""
    rng = np.random.default_rng(5)
    for shape in [(40, 46), (40, 44), (22, 30, 3), (22, 30, 4), (22, 30, 2), (10, 14), (6, 4)]:
        src = (rng.standard_normal(shape) * 10 ** rng.uniform(-3, 3, shape)).astype(np.float32)
        h, w = shape[:2]
        assert np.array_equal(orc.orc_resize(src, (w // 2, h // 2), interpolation=3), orc.ref_resize(src, (w // 2, h // 2), interpolation=3)), shape


def _bits(got, want):
    """bit for bit, CV_32F included (the restatement keeps the reference's order of float operations); NaN-safe"""
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_warp_cubic_lanczos(orc, ref, dtype, cn):
    """INTER_CUBIC (2) / INTER_LANCZOS4 (4) in warpAffine and warpPerspective (remapBicubic / remapLanczos4 imgwarp.cpp:905-1120 behind the warps' fixed-point
    coordinates): every border rule incl. BORDER_TRANSPARENT, maps that leave the source on every side; the weight tables (initInterTab2D :213-262) are pinned
    through the results"""
    src = rnd(orc, (45, 61, cn) if cn > 1 else (45, 61), dtype, 515 + cn)
    P = np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]])
    for interp in (2, 4):
        for M in mats(orc, 61, 45):
            for dsize in [(61, 45), (100, 30)]:
                prev = rnd(orc, (dsize[1], dsize[0], cn) if cn > 1 else (dsize[1], dsize[0]), dtype, 9)
                for border, bval in [(0, 0.0), (0, (10, 200, 30, 77)), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0)]:
                    d0 = prev if border == 5 else None
                    _bits(orc.orc_warpAffine(src, M, dsize, interp, border, bval, dst=d0), orc.ref_warpAffine(src, M, dsize, interp | 16, border, bval, dst=d0))
        for border, bval in [(0, 5.0), (1, 0), (4, 0), (5, 0)]:
            prev = rnd(orc, (45, 61, cn) if cn > 1 else (45, 61), dtype, 10)
            d0 = prev if border == 5 else None
            _bits(orc.orc_warpPerspective(src, P, (61, 45), interp, border, bval, dst=d0), orc.ref_warpPerspective(src, P, (61, 45), interp | 16, border, bval, dst=d0))


@pytest.mark.parametrize("dtype", DT)
def test_remap_cubic_lanczos(orc, ref, dtype):
    """cv::remap with INTER_CUBIC / INTER_LANCZOS4: CV_32FC1 pairs, a CV_32FC2 map, the fixed-point maps, and all of them with WARP_RELATIVE_MAP"""
    REL = 32
    src = rnd(orc, (40, 50, 3), dtype, 536)
    mapx, mapy = _float_maps(14)
    xy = np.ascontiguousarray(np.stack([mapx, mapy], axis=-1))
    f1, f2 = orc.ref_convertMaps(mapx, mapy, "16sc2", False)
    rng = np.random.default_rng(18)
    offx = rng.uniform(-6, 6, (33, 47)).astype(np.float32); offy = rng.uniform(-6, 6, (33, 47)).astype(np.float32)
    offx[2, 3] = 40000.0; offy[5, 6] = -40000.0
    oxy = np.ascontiguousarray(np.stack([offx, offy], axis=-1))
    o1, o2 = orc.ref_convertMaps(offx, offy, "16sc2", False)
    prev = rnd(orc, (33, 47, 3), dtype, 37)
    for interp in (2, 4):
        for border, bval in [(0, 9.0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0)]:
            d0 = prev if border == 5 else None
            if border != 5:
                _bits(orc.orc_remap(src, mapx, mapy, interp, border, bval), orc.ref_remap(src, mapx, mapy, interp, border, bval))
                _bits(orc.orc_remap(src, offx, offy, interp | REL, border, bval), orc.ref_remap(src, offx, offy, interp | REL, border, bval))
            _bits(orc.orc_remapMaps(src, xy, None, interp, border, bval, dst=d0), orc.ref_remapMaps(src, xy, None, interp, border, bval, dst=d0))
            _bits(orc.orc_remapMaps(src, f1, f2, interp, border, bval, dst=d0), orc.ref_remapMaps(src, f1, f2, interp, border, bval, dst=d0))
            _bits(orc.orc_remapMaps(src, oxy, None, interp | REL, border, bval, dst=d0), orc.ref_remapMaps(src, oxy, None, interp | REL, border, bval, dst=d0))
            _bits(orc.orc_remapMaps(src, o1, o2, interp | REL, border, bval, dst=d0), orc.ref_remapMaps(src, o1, o2, interp | REL, border, bval, dst=d0))


def test_warps_on_64f_images(orc, ref):
    """CV_64F images (remapNearest<double>, remapBilinear / remapBicubic / remapLanczos4 with WT = double and the float weight tables, imgwarp.cpp:1736-1790): warpAffine,
    warpPerspective and cv::remap with every map representation, plain and relative -- the 60 Imgproc_RemapRelative cases the ledger still showed on the fallback"""
    REL = 32
    rng = np.random.default_rng(5)
    src = rng.random((40, 50, 3)) * 1000 - 300
    M = orc.ref_getRotationMatrix2D((25.0, 20.0), 33.0, 1.3)
    P = np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]])
    prev = rng.random((33, 47, 3))
    mapx, mapy = _float_maps(24)
    xy = np.ascontiguousarray(np.stack([mapx, mapy], axis=-1))
    f1, f2 = orc.ref_convertMaps(mapx, mapy, "16sc2", False)
    offx = rng.uniform(-6, 6, (33, 47)).astype(np.float32); offy = rng.uniform(-6, 6, (33, 47)).astype(np.float32)
    for interp in (0, 1, 2, 4):
        for border, bval in [(0, (1.5, -2.25, 1e3 / 3, 0)), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0)]:
            d0 = prev if border == 5 else None
            _bits(orc.orc_warpAffine(src, M, (47, 33), interp, border, bval, dst=d0), orc.ref_warpAffine(src, M, (47, 33), interp | 16, border, bval, dst=d0))
            _bits(orc.orc_warpPerspective(src, P, (47, 33), interp, border, bval, dst=d0), orc.ref_warpPerspective(src, P, (47, 33), interp | 16, border, bval, dst=d0))
            if border != 5:
                _bits(orc.orc_remap(src, mapx, mapy, interp, border, bval), orc.ref_remap(src, mapx, mapy, interp, border, bval))
                _bits(orc.orc_remap(src, offx, offy, interp | REL, border, bval), orc.ref_remap(src, offx, offy, interp | REL, border, bval))
            _bits(orc.orc_remapMaps(src, xy, None, interp, border, bval, dst=d0), orc.ref_remapMaps(src, xy, None, interp, border, bval, dst=d0))
            if interp:
                _bits(orc.orc_remapMaps(src, f1, f2, interp, border, bval, dst=d0), orc.ref_remapMaps(src, f1, f2, interp, border, bval, dst=d0))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [5, 6, 9, 14])
def test_more_than_four_channels(orc, ref, dtype, cn):
    """5-14 channels (Imgproc_Warp.multichannel, Resize.nearest_regression_15075): the restatement's channel loops against the reference -- nearest / bilinear warps
    (border value of channel k = borderValue[k & 3]) and every resize mode but true area; the reference itself asserts on bicubic / Lanczos warps (imgwarp.cpp:2795)
    and on true INTER_AREA (resize.cpp:4045) of more than 4 channels."""
    src = rnd(orc, (37, 53, cn), dtype, 40 + cn)
    M = mats(orc, 53, 37)[1]
    P = np.vstack([M, [1e-4, 2e-4, 1.0]])
    for interp in (0, 1):
        for border, bval in [(0, (10, 200, 30, 77)), (1, 0), (2, 0), (3, 0), (4, 0)]:
            same(orc, orc.orc_warpAffine(src, M, (61, 41), interp, border, bval), orc.ref_warpAffine(src, M, (61, 41), interp | 16, border, bval))
            same(orc, orc.orc_warpPerspective(src, P, (61, 41), interp, border, bval), orc.ref_warpPerspective(src, P, (61, 41), interp | 16, border, bval))
    for interp in [0, 1, 2, 4, 6] + ([5] if dtype != np.float32 else []):
        for dsize in [(80, 55), (31, 23), (106, 74)]:
            same(orc, orc.orc_resize(src, dsize, interpolation=interp), orc.ref_resize(src, dsize, interpolation=interp), tol=1e-5 if interp in (2, 4) else 1e-6)
    s2 = np.ascontiguousarray(src[:36, :52])
    same(orc, orc.orc_resize(s2, (26, 18), interpolation=3), orc.ref_resize(s2, (26, 18), interpolation=3))
    for bad in (lambda: orc.ref_warpAffine(src, M, (61, 41), 2 | 16), lambda: orc.ref_resize(src, (31, 23), interpolation=3)):
        with pytest.raises(Exception):
            bad()
