"""GPU parity for rows a7-a9: resize, warpAffine, warpPerspective, remap -- through the C ABI against the oracle.
8U/16U/16S bit-exact (test_imgwarp_strict.cpp:1089-1092 demands 0 for 8U warpAffine), 32F within 1e-4 relative
(in practice identical: same operation order, no FMA)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DT = [np.uint8, np.uint16, np.int16, np.float32]


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def dev(a):
    return torch.from_numpy(a).cuda()


def rnd(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    if dtype == np.float32:
        return rng.random(shape, dtype=np.float32)
    info = np.iinfo(dtype)
    return rng.integers(info.min, int(info.max) + 1, shape, dtype=dtype)


def check(got, want, tol=1e-6):
    import orc
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.dtype == want.dtype and got.shape == want.shape
    if want.dtype == np.float32:
        assert orc.rel_err(got, want) <= tol
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize(cv, orc, dtype, cn):
    src = rnd((37, 53, cn) if cn > 1 else (37, 53), dtype, 5 + cn)
    n0 = cv.call_count("resize")
    for dsize in [(80, 55), (35, 25), (106, 74), (17, 9), (1, 1), (200, 3)]:
        for interp in (0, 1):
            check(cv.resize(dev(src), dsize, interpolation=interp), orc.orc_resize(src, dsize, interpolation=interp))
    for fx, fy in [(0.5, 0.5), (1.5, 1.5), (0.75, 1.25), (2.0, 2.0)]:
        check(cv.resize(dev(src), None, fx, fy, 1), orc.orc_resize(src, None, fx, fy, 1))
    check(cv.resize(src, (80, 55)), orc.orc_resize(src, (80, 55)))                           # host arrays
    assert cv.call_count("resize") > n0
    for (w, h) in [(48, 72), (60, 36)]:
        s2 = rnd((h, w, cn) if cn > 1 else (h, w), dtype, 9 + w)
        for s in (2, 3, 4):
            check(cv.resize(dev(s2), (w // s, h // s), interpolation=3), orc.orc_resize(s2, (w // s, h // s), interpolation=3))
        check(cv.resize(dev(s2), (w * 2, h * 2), interpolation=3), orc.orc_resize(s2, (w * 2, h * 2), interpolation=3))
    s3 = rnd((59, 55, cn) if cn > 1 else (59, 55), dtype, 77)
    check(cv.resize(dev(s3), None, 0.5, 0.5, 1), orc.orc_resize(s3, None, 0.5, 0.5, 1))
    assert torch.equal(cv.resize(dev(src), (53, 37)), dev(src))                            # same size -> copy


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_area_2x2_u8(cv, orc, cn):
    """CV_8U halved exactly (INTER_AREA, and INTER_LINEAR which becomes it, resize.cpp:4011): the four-pixels-per-lane kernel -- widths whose quarter
    is / is not whole (the last lane's scalar tail), views with a parent's step, a batch; odd sizes keep the generic kernel"""
    for (w, h) in [(64, 48), (70, 50), (8, 2), (1000, 36), (3840, 16)]:
        src = rnd((h, w, cn), np.uint8, 31 + w + cn) if cn > 1 else rnd((h, w), np.uint8, 31 + w)
        for interp in (3, 1):
            check(cv.resize(dev(src), (w // 2, h // 2), interpolation=interp), orc.orc_resize(src, (w // 2, h // 2), interpolation=interp))
    big = rnd((3, 60, 88, cn), np.uint8, 5) if cn > 1 else rnd((3, 60, 88), np.uint8, 5)
    d = dev(big)
    view = d[1, 4:52, 8:72]
    check(cv.resize(view, (32, 24), interpolation=3), orc.orc_resize(np.ascontiguousarray(big[1, 4:52, 8:72]), (32, 24), interpolation=3))
    out = cv.resizeBatch(d, (44, 30), interpolation=3)
    for f in range(3):
        check(out[f], orc.orc_resize(big[f], (44, 30), interpolation=3))
    odd = rnd((51, 67, cn), np.uint8, 6) if cn > 1 else rnd((51, 67), np.uint8, 6)
    check(cv.resize(dev(odd), (33, 25), interpolation=3), orc.orc_resize(odd, (33, 25), interpolation=3))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_area_general(cv, orc, dtype, cn):
    """true INTER_AREA (non-integer shrink ratios): bit-exact for the integer depths, and for float too since every multiply /
    add runs in the reference's order without contraction"""
    for (w, h), dsizes in [((53, 37), [(20, 11), (52, 36), (17, 37), (53, 9)]), ((100, 100), [(33, 33), (99, 51)]), ((640, 480), [(213, 160), (400, 111)])]:
        src = rnd((h, w, cn) if cn > 1 else (h, w), dtype, 3 + cn + w)
        for dsize in dsizes:
            got = cv.resize(dev(src), dsize, interpolation=3).cpu().numpy()
            assert np.array_equal(got, orc.orc_resize(src, dsize, interpolation=3)), (w, h, dsize, dtype, cn)
    src = rnd((45, 77, cn) if cn > 1 else (45, 77), dtype, 8)
    assert np.array_equal(cv.resize(src, None, 0.3, 0.7, 3), orc.orc_resize(src, None, 0.3, 0.7, 3))          # host pointers


@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.uint16, np.int16])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_cubic(cv, orc, dtype, cn):
    """INTER_CUBIC, bit-exact incl. the reference's SIMD-body / scalar-tail split of the vertical pass (8U fixed point; 32F, 16U and 16S float)"""
    for (w, h), dsizes in [((53, 37), [(80, 55), (20, 11), (106, 74), (161, 3)]), ((9, 9), [(31, 29), (8, 8)]), ((5, 3), [(17, 13)]), ((640, 480), [(1280, 960), (333, 222)])]:
        src = rnd((h, w, cn) if cn > 1 else (h, w), dtype, 6 + cn + w)
        for dsize in dsizes:
            got = cv.resize(dev(src), dsize, interpolation=2).cpu().numpy()
            assert np.array_equal(got, orc.orc_resize(src, dsize, interpolation=2)), (w, h, dsize, dtype, cn)
    big = rnd((480, 640, cn) if cn > 1 else (480, 640), dtype, 3)
    for dsize in [(64, 48), (700, 31)]:                               # strong minification: the per-output kernel (a tile would need > 64 source rows)
        assert np.array_equal(cv.resize(dev(big), dsize, interpolation=2).cpu().numpy(), orc.orc_resize(big, dsize, interpolation=2)), (dsize, dtype, cn)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.uint16, np.int16])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_lanczos4(cv, orc, dtype, cn):
    """INTER_LANCZOS4, bit-exact (CV_8U integer; CV_32F / CV_16S with the reference's body / tail summation orders, CV_16U left to right throughout)"""
    for (w, h), dsizes in [((53, 37), [(80, 55), (20, 11), (106, 74), (161, 3)]), ((9, 9), [(31, 29), (8, 8)]), ((5, 3), [(17, 13)]), ((640, 480), [(1280, 960), (333, 222)])]:
        src = rnd((h, w, cn) if cn > 1 else (h, w), dtype, 9 + cn + w)
        for dsize in dsizes:
            got = cv.resize(dev(src), dsize, interpolation=4).cpu().numpy()
            assert np.array_equal(got, orc.orc_resize(src, dsize, interpolation=4)), (w, h, dsize, dtype, cn)
    big = rnd((480, 640, cn) if cn > 1 else (480, 640), dtype, 4)
    for dsize in [(64, 48), (700, 31)]:
        assert np.array_equal(cv.resize(dev(big), dsize, interpolation=4).cpu().numpy(), orc.orc_resize(big, dsize, interpolation=4)), (dsize, dtype, cn)
    with pytest.raises(NotImplementedError):
        cv.resize(dev(rnd((20, 30), np.uint16, 1)), (40, 60), interpolation=7)          # no such interpolation: declined, never a CPU fallback


def mats(cv, w, h):
    out = [cv.getRotationMatrix2D((w / 2.0, h / 2.0), a, s) for a, s in [(7.0, 0.95), (33.0, 1.3), (-120.0, 0.6), (0.0, 1.0)]]
    out.append(np.array([[1, 0, 3.25], [0, 1, -2.5]], np.float64))
    out.append(np.array([[0.3, 0.1, -20.0], [-0.2, 0.4, 30.0]], np.float64))
    return out


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_warp_affine(cv, orc, dtype, cn):
    src = rnd((45, 61, cn) if cn > 1 else (45, 61), dtype, 15 + cn)
    for M in mats(cv, 61, 45):
        for dsize in [(61, 45), (100, 30)]:
            for interp in (0, 1):
                for border, bval in [(0, 0.0), (0, (10, 200, 30, 77)), (1, 0), (2, 0), (3, 0), (4, 0)]:
                    want = orc.orc_warpAffine(src, M, dsize, interp, border, bval)
                    check(cv.warpAffine(dev(src), M, dsize, interp | cv.WARP_INVERSE_MAP, border, bval), want)
    M = mats(cv, 61, 45)[0]
    check(cv.warpAffine(src, M, (61, 45), 1 | cv.WARP_INVERSE_MAP), orc.orc_warpAffine(src, M, (61, 45)))   # host arrays
    # forward matrix: the wrapper inverts it as cv::warpAffine does (imgwarp.cpp:2824-2834)
    check(cv.warpAffine(dev(src), M, (61, 45)), orc.orc_warpAffine(src, cv.invertAffineTransform(M), (61, 45)))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_tile_orders(cv, orc, dtype):
    """destination sizes whose tile count is not a multiple of 8 (the XCD-banded tile order of the CV_32F kernels must still cover every tile:
    the reference's own Imgproc_WarpAffine.accuracy caught a hole here), affine and perspective"""
    src = rnd((180, 230), dtype, 55)
    M = cv.getRotationMatrix2D((115.0, 90.0), 11.0, 0.9)
    P = np.array([[1.05, 0.04, -6.0], [0.03, 0.95, 5.0], [1e-4, -1e-4, 1.0]])
    for dsize in [(700, 300), (65, 33), (129, 97), (641, 479), (1000, 40)]:
        check(cv.warpAffine(dev(src), M, dsize, 1 | cv.WARP_INVERSE_MAP, 1), orc.orc_warpAffine(src, M, dsize, 1, 1))
        check(cv.warpPerspective(dev(src), P, dsize, 1 | cv.WARP_INVERSE_MAP, 0, 3.0), orc.orc_warpPerspective(src, P, dsize, 1, 0, 3.0))


def test_warp_32f_wide_sources(cv, orc):
    """CV_32FC1 sources several tiles wide: rotations, strong magnification / minification, shear, borders, ragged destination sizes, perspective,
    a view inside a parent, a batch -- the grouped-load bilinear kernel (k_warp_lin) on sources larger than one workgroup's footprint"""
    src = rnd((200, 256), np.float32, 4242)
    Ms = [cv.getRotationMatrix2D((128.0, 100.0), a, sc) for a, sc in [(7.0, 0.95), (33.0, 1.3), (-120.0, 0.6), (0.0, 1.0), (90.0, 1.0), (45.0, 3.5), (3.0, 0.3)]]
    Ms += [np.array([[1, 0, 3.25], [0, 1, -2.5]], np.float64), np.array([[0.3, 0.1, -20.0], [-0.2, 0.4, 30.0]], np.float64),
           np.array([[1.0, 0.9, -60.0], [0.0, 1.0, 0.0]], np.float64)]
    P = [np.array([[1.05, 0.04, -6.0], [0.03, 0.95, 5.0], [1e-4, -1e-4, 1.0]]), np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]])]
    d = dev(src)
    for dsize in [(256, 200), (300, 77), (65, 33), (640, 480)]:
        for border, bval in [(0, 0.0), (0, 2.5), (1, 0), (4, 0)]:
            for M in Ms:
                got = cv.warpAffine(d, M, dsize, 1 | cv.WARP_INVERSE_MAP, border, bval)
                check(got, orc.orc_warpAffine(src, M, dsize, 1, border, bval))
            for M in P:
                got = cv.warpPerspective(d, M, dsize, 1 | cv.WARP_INVERSE_MAP, border, bval)
                check(got, orc.orc_warpPerspective(src, M, dsize, 1, border, bval))
    # a view with a 16-byte aligned, non-zero origin and a parent's step; frames of a batch
    big = rnd((3, 220, 272), np.float32, 99)
    dbig = dev(big)
    view = dbig[1, 8:208, 8:264]
    check(cv.warpAffine(view, Ms[0], (256, 200), 1 | cv.WARP_INVERSE_MAP, 0, 0.0), orc.orc_warpAffine(np.ascontiguousarray(big[1, 8:208, 8:264]), Ms[0], (256, 200), 1, 0, 0.0))
    outs = cv.warpAffineBatch(dbig, Ms[1], (272, 220), 1 | cv.WARP_INVERSE_MAP, 1)
    for f in range(3):
        check(outs[f], orc.orc_warpAffine(big[f], Ms[1], (272, 220), 1, 1))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_warp_transparent(cv, orc, dtype, cn):
    """BORDER_TRANSPARENT incl. the partial-overlap blend of the source's last row / column (imgwarp.cpp:786-815), device and host dst"""
    src = rnd((45, 61, cn) if cn > 1 else (45, 61), dtype, 115 + cn)
    Ms = mats(cv, 61, 45) + [np.array([[1.0, 0, 0.25], [0, 1.0, 0.5]]), np.array([[0.5, 0, 30.0], [0, 0.5, 22.0]]), np.array([[1.0, 0, 0], [0, 1.0, 0]])]
    for M in Ms:
        for dsize in [(61, 45), (100, 30)]:
            prev = rnd((dsize[1], dsize[0], cn) if cn > 1 else (dsize[1], dsize[0]), dtype, 7)
            for interp in (0, 1):
                want = orc.orc_warpAffine(src, M, dsize, interp, 5, 0.0, dst=prev)
                check(cv.warpAffine(dev(src), M, dsize, interp | cv.WARP_INVERSE_MAP, 5, 0.0, dst=dev(prev.copy())), want)
                check(cv.warpAffine(src, M, dsize, interp | cv.WARP_INVERSE_MAP, 5, 0.0, dst=prev.copy()), want)
    P = np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]])
    prev = rnd((45, 61, cn) if cn > 1 else (45, 61), dtype, 8)
    for interp in (0, 1):
        check(cv.warpPerspective(dev(src), P, (61, 45), interp | cv.WARP_INVERSE_MAP, 5, 0.0, dst=dev(prev.copy())),
              orc.orc_warpPerspective(src, P, (61, 45), interp, 5, 0.0, dst=prev))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("cn", [1, 3])
def test_warp_perspective_and_remap(cv, orc, dtype, cn):
    src = rnd((45, 61, cn) if cn > 1 else (45, 61), dtype, 25 + cn)
    Ms = [np.array([[1.1, 0.05, -3.0], [0.02, 0.9, 4.0], [1e-4, -2e-4, 1.0]]),
          np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]]), np.eye(3)]
    for M in Ms:
        for dsize in [(61, 45), (150, 40), (7, 70)]:
            for interp in (0, 1):
                for border, bval in [(0, 5.0), (1, 0), (4, 0)]:
                    check(cv.warpPerspective(dev(src), M, dsize, interp | cv.WARP_INVERSE_MAP, border, bval),
                          orc.orc_warpPerspective(src, M, dsize, interp, border, bval))
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:33, 0:47].astype(np.float32)
    mapx = (xx * 1.07 + rng.uniform(-3, 3, xx.shape)).astype(np.float32)
    mapy = (yy * 1.2 - 2 + rng.uniform(-3, 3, yy.shape)).astype(np.float32)
    for interp in (0, 1):
        for border, bval in [(0, 9.0), (1, 0), (2, 0), (4, 0)]:
            check(cv.remap(dev(src), dev(mapx), dev(mapy), interp, border, bval), orc.orc_remap(src, mapx, mapy, interp, border, bval))


def test_config3_8k_float(cv, orc):
    """BASELINE config 3: resize (bilinear) + warpAffine on 7680x4320 CV_32F (checked on crops the oracle finishes quickly)."""
    src = rnd((4320, 7680), np.float32, 809564)
    d = dev(src)
    up = cv.resize(d, (5120, 2880))
    half = cv.resize(d, (3840, 2160))
    # the top-left 600x400 of each result depends only on the top-left of the source
    crop = np.ascontiguousarray(src[:700, :1000])
    want = orc.orc_resize(crop, None, 5120 / 7680, 2880 / 4320, 1)
    check(up[:400, :600], np.ascontiguousarray(want[:400, :600]))
    want2 = orc.orc_resize(crop, (500, 350), interpolation=1)
    check(half[:350, :500], want2)
    M = cv.getRotationMatrix2D((7680 / 2.0, 4320 / 2.0), 7.0, 0.95)
    Minv = cv.invertAffineTransform(M)
    out = cv.warpAffine(d, M, (7680, 4320))
    want3 = orc_band(orc, src, Minv)        # the oracle handles one band of rows fast enough
    check(out[2000:2200], want3)


def orc_band(orc, src, Minv, y0=2000, y1=2200):
    """rows [y0,y1) of warpAffine(src, Minv, same size): shift the row origin into the matrix (x' = M0 x + M1 (y+y0) + M2)"""
    Mb = np.array(Minv, np.float64).copy()
    full = orc.orc_warpAffine  # noqa: F841
    # the reference evaluates (M1*y + M2) per absolute row; emulate by running the oracle on the whole height would take minutes,
    # so instead run it row-exactly through a tall-enough destination of which only the band is computed:
    import ctypes
    o = orc.oracle()
    h, w = src.shape
    dst = np.empty((y1, w), np.float32)
    bv = np.zeros(4, np.float64)
    # compute rows 0..y1-1 lazily: the oracle is O(rows); 2200 rows of 7680 px ~ 17 Mpix -> a few seconds
    rc = o.orc_warpAffine(orc.P(src), orc.step(src), w, h, orc.P(dst), orc.step(dst), w, y1, 5, 1, orc.P(np.ascontiguousarray(Mb)), 1, 0, orc.P(bv))
    assert rc == 0
    return np.ascontiguousarray(dst[y0:y1])


def _float_maps(seed, w=47, h=33, sw=50, sh=40):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    mapx = (xx * (sw / w) + rng.uniform(-3, 3, xx.shape)).astype(np.float32)
    mapy = (yy * (sh / h) - 1 + rng.uniform(-3, 3, yy.shape)).astype(np.float32)
    return mapx, mapy


def test_convert_maps(cv, orc):
    """cv::convertMaps both ways, device and host maps: bit-exact with the restatement (pinned to the reference in test_oracle_warp.py)"""
    mapx, mapy = _float_maps(5)
    mapx[3, 4] = 17.515625; mapy[3, 4] = 8.984375
    xy = np.ascontiguousarray(np.stack([mapx, mapy], axis=-1))
    T16SC2, T32FC1, T32FC2 = cv.CV_MAKETYPE(cv.CV_16S, 2), cv.CV_MAKETYPE(cv.CV_32F, 1), cv.CV_MAKETYPE(cv.CV_32F, 2)
    for nn in (False, True):
        for m1, m2 in ((mapx, mapy), (xy, None)):
            w1, w2 = orc.orc_convertMaps(m1, m2, "16sc2", nn)
            g1, g2 = cv.convertMaps(dev(m1), dev(m2) if m2 is not None else None, T16SC2, nn)
            assert np.array_equal(g1.cpu().numpy(), w1) and (nn or np.array_equal(g2.cpu().numpy(), w2))
            h1, h2 = cv.convertMaps(m1, m2, T16SC2, nn)                                      # host maps are staged
            assert np.array_equal(h1, w1) and (nn or np.array_equal(h2, w2))
    f1, f2 = orc.orc_convertMaps(mapx, mapy, "16sc2", False)
    for code, name in ((T32FC1, "32fc1"), (T32FC2, "32fc2")):
        w1, w2 = orc.orc_convertMaps(f1, f2, name)
        g1, g2 = cv.convertMaps(dev(f1), dev(f2), code)
        assert np.array_equal(g1.cpu().numpy(), w1) and (w2 is None or np.array_equal(g2.cpu().numpy(), w2))


@pytest.mark.parametrize("dtype", DT)
def test_remap_other_map_types(cv, orc, dtype):
    """cv::remap with a CV_32FC2 map and with fixed-point maps (CV_16SC2 + CV_16UC1; CV_16SC2 alone for nearest)"""
    src = rnd((40, 50, 3), dtype, 36)
    mapx, mapy = _float_maps(4)
    xy = np.ascontiguousarray(np.stack([mapx, mapy], axis=-1))
    f1, f2 = orc.orc_convertMaps(mapx, mapy, "16sc2", False)
    n1, _ = orc.orc_convertMaps(mapx, mapy, "16sc2", True)
    prev = rnd((33, 47, 3), dtype, 37)
    for interp in (0, 1):
        for border, bval in [(0, 9.0), (1, 0), (2, 0), (4, 0), (5, 0)]:
            for m1, m2 in ((xy, None), (f1, f2)):
                want = orc.orc_remapMaps(src, m1, m2, interp, border, bval, dst=prev if border == 5 else None)
                d0 = dev(prev.copy()) if border == 5 else None
                check(cv.remap(dev(src), dev(m1), dev(m2) if m2 is not None else None, interp, border, bval, dst=d0), want)
        check(cv.remap(dev(src), dev(n1), None, 0, 1, 0), orc.orc_remapMaps(src, n1, None, 0, 1, 0))
    check(cv.remap(src, f1, f2, 1, 4, 0), orc.orc_remapMaps(src, f1, f2, 1, 4, 0))          # host arrays
    with pytest.raises(NotImplementedError):
        cv.remap(dev(src), dev(n1), None, 1, 0, 0)                                           # CV_16SC2 alone is a nearest-only representation


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_polar_forward(cv, orc, dtype):
    src = rnd((60, 80, 3), dtype, 41)
    for flags in (1, 1 | 8, 0 | 8, 1 | 256, 1 | 256 | 8):
        for dsize, center, rad in [((64, 90), (40.0, 30.0), 35.0), ((50, 157), (10.5, 50.25), 60.0)]:
            check(cv.warpPolar(dev(src), dsize, center, rad, flags), orc.orc_warpPolar(src, dsize, center, rad, flags))
    big = rnd((1080, 1920), np.uint8, 42)
    check(cv.warpPolar(dev(big), (1024, 2048), (960.0, 540.0), 600.0, 1 | 8), orc.orc_warpPolar(big, (1024, 2048), (960.0, 540.0), 600.0, 1 | 8))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16])
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_resize_linear_exact(cv, orc, dtype, cn):
    """INTER_LINEAR_EXACT on the GPU = the reference's fixed-point arithmetic, bit-exact (Resize_Bitexact.Linear8U of the reference's own test binary
    now exercises this kernel instead of falling back)"""
    src = rnd((37, 53, cn) if cn > 1 else (37, 53), dtype, 7 + cn)
    for dsize in [(80, 60), (20, 11), (53, 37), (106, 74), (26, 18), (27, 19), (1, 1), (200, 5), (7, 90)]:
        assert np.array_equal(cv.resize(dev(src), dsize, interpolation=5).cpu().numpy(), orc.orc_resize(src, dsize, interpolation=5)), (dsize, dtype, cn)
    big = rnd((1080, 1920, cn) if cn > 1 else (1080, 1920), dtype, 9)
    assert np.array_equal(cv.resize(dev(big), (1280, 720), interpolation=5).cpu().numpy(), orc.orc_resize(big, (1280, 720), interpolation=5))
    fr = torch.from_numpy(np.stack([src, src[::-1].copy()])).cuda()
    got = cv.resizeBatch(fr, (80, 60), interpolation=5)
    assert np.array_equal(got[1].cpu().numpy(), orc.orc_resize(np.ascontiguousarray(src[::-1]), (80, 60), interpolation=5))
    with pytest.raises(NotImplementedError):
        cv.resize(dev(rnd((20, 30), np.float32, 1)), (40, 60), interpolation=5)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_nearest_exact(cv, orc, dtype, cn):
    """INTER_NEAREST_EXACT (resizeNN_bitexact: 16.16 fixed-point steps on pixel centres) -- equal to the restatement, which is pinned to cv::resize"""
    src = rnd((37, 53, cn) if cn > 1 else (37, 53), dtype, 11 + cn)
    for dsize in [(80, 60), (20, 11), (53, 37), (106, 74), (27, 19), (1, 1), (200, 5), (7, 90)]:
        assert np.array_equal(cv.resize(dev(src), dsize, interpolation=6).cpu().numpy(), orc.orc_resize(src, dsize, interpolation=6)), (dsize, dtype, cn)
    big = rnd((1080, 1920, cn) if cn > 1 else (1080, 1920), dtype, 9)
    assert np.array_equal(cv.resize(dev(big), (3840, 2160), interpolation=6).cpu().numpy(), orc.orc_resize(big, (3840, 2160), interpolation=6))
    assert np.array_equal(cv.resize(dev(big), (1281, 719), interpolation=6).cpu().numpy(), orc.orc_resize(big, (1281, 719), interpolation=6))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_polar_inverse(cv, orc, dtype):
    """WARP_INVERSE_MAP: polar / semi-log polar image -> Cartesian image.  The map comes from the reference's float approximations of cartToPolar and log
    (their vector forms; the scalar forms for rows narrower than 16 / 8 pixels), restated in the kernel; the restatement is pinned bit for bit to
    cv::warpPolar (tests/test_oracle_warp.py), so the outputs compare as usual: 8-bit equal, CV_32F to 1e-4 of the range."""
    for cn in (1, 3):
        src = rnd((90, 64, cn) if cn > 1 else (90, 64), dtype, 43 + cn)
        for flags in (1 | 16 | 8, 0 | 16 | 8, 1 | 16 | 256 | 8, 0 | 16 | 256 | 8):
            for dsize, center, rad in [((80, 60), (40.0, 30.0), 35.0), ((12, 30), (5.5, 14.25), 20.0), ((7, 9), (3.0, 4.0), 6.0), ((1100, 40), (600.5, 20.0), 500.0)]:
                check(cv.warpPolar(dev(src), dsize, center, rad, flags), orc.orc_warpPolar(src, dsize, center, rad, flags))
    # without WARP_FILL_OUTLIERS the untouched pixels keep the destination's previous contents
    src = rnd((90, 64), dtype, 47)
    prev = rnd((60, 80), dtype, 48)
    got = cv.warpPolar(dev(src), (80, 60), (40.0, 30.0), 20.0, 1 | 16, dst=dev(prev.copy()))
    want = orc.orc_warpPolar(src, (80, 60), (40.0, 30.0), 20.0, 1 | 16 | 8)
    yy, xx = np.mgrid[0:60, 0:80]
    far = (xx - 40.0) ** 2 + (yy - 30.0) ** 2 > (20.0 + 3) ** 2                               # beyond maxRadius (+ the bilinear rim): the map leaves the source
    near = (xx - 40.0) ** 2 + (yy - 30.0) ** 2 < (20.0 - 3) ** 2
    g = got.cpu().numpy()
    assert np.array_equal(g[far], prev[far]) and far.sum() > 1000
    assert np.array_equal(g[near], want[near]) if dtype == np.uint8 else orc.rel_err(g[near], want[near]) <= 1e-6
    big = rnd((2048, 1024), np.uint8, 49)
    check(cv.warpPolar(dev(big), (1920, 1080), (960.0, 540.0), 600.0, 1 | 16 | 8), orc.orc_warpPolar(big, (1920, 1080), (960.0, 540.0), 600.0, 1 | 16 | 8))
    check(cv.warpPolar(big, (1920, 1080), (960.0, 540.0), 600.0, 1 | 16 | 8), orc.orc_warpPolar(big, (1920, 1080), (960.0, 540.0), 600.0, 1 | 16 | 8))   # host image


@pytest.mark.parametrize("cn", [1, 3])
def test_resize_bilinear_u8_on_the_lean_kernel(cv, orc, cn):
    """CV_8U INTER_LINEAR (and INTER_AREA upscales) with 4-byte aligned rows run on k_resize8_lean (LDS pipeline of the lean warp kernel): identical to the
    restatement for up- and downscales, partial lanes / tiles, a batch, and the predicated loader at the image's last row"""
    from opencv_amd import _lib
    for (sw, sh, dw, dh, interp) in [(480, 270, 960, 540, 1), (480, 270, 720, 404, 1), (640, 360, 480, 272, 1), (400, 300, 1000, 700, 1), (336, 200, 1336, 804, 1),
                                     (480, 270, 960, 540, 3), (300, 200, 452, 301, 1), (128, 64, 512, 256, 1)]:
        src = rnd((sh, sw, cn) if cn > 1 else (sh, sw), np.uint8, sw + cn)
        got = cv.resize(dev(src), (dw, dh), interpolation=interp)
        k = _lib.lib.mi355cv_lastKernel().decode()
        assert np.array_equal(got.cpu().numpy(), orc.orc_resize(src, (dw, dh), interpolation=interp)), (cn, sw, sh, dw, dh, interp, k)
        if (sw * cn) % 4 == 0 and (dw * cn) % 4 == 0 and not (cn == 3 and dw < sw):
            assert "k_resize8_lean" in k, (cn, sw, dw, k)
    frames = np.stack([rnd((270, 480, cn) if cn > 1 else (270, 480), np.uint8, 40 + i) for i in range(3)])
    out = cv.resizeBatch(dev(frames), (960, 540))
    for i in range(3):
        assert np.array_equal(out[i].cpu().numpy(), orc.orc_resize(frames[i], (960, 540))), i


def test_dispatch_lands_on_the_lds_tile_kernels(cv, orc):
    """the kernels whose arithmetic the host emulation (tests/test_hostemu.py) checks line by line are the ones the GPU calls land on --
    asserted by name (mi355cv_lastKernel) next to the result's equality with the restatement: k_warp8_lean (8-bit affine, 1 / 3 channels), k_warp8_tile (what
    the lean plan does not take), k_resize_tab8 (8-bit cubic upscale)."""
    from opencv_amd import _lib
    last = lambda: _lib.lib.mi355cv_lastKernel().decode()       # noqa: E731
    for cn in (1, 3):
        src = rnd((540, 960, cn) if cn > 1 else (540, 960), np.uint8, 70 + cn)
        for ang, sc in ((7.0, 1.0), (90.0, 0.95), (-33.0, 1.1)):
            M = cv.getRotationMatrix2D((480.0, 270.0), ang, sc)
            got = cv.warpAffine(dev(src), M, (960, 540))
            k = last()
            # the lean plan takes what fits its two LDS buffers; the -33 degree, 3-channel box (110 x 79 pixels) does not and stays on the general tile kernel
            assert ("k_warp8_lean<%d," % cn in k) or (cn == 3 and ang == -33.0 and "k_warp8_tile<3,0," in k), (cn, ang, k)
            assert np.array_equal(got.cpu().numpy(), orc.orc_warpAffine(src, cv.invertAffineTransform(M), (960, 540))), (cn, ang, k)
        up = cv.resize(dev(src[:270, :480]), (960, 540), interpolation=2)
        k = last()
        assert "k_resize_tab8<4>" in k, (cn, k)
        assert np.array_equal(up.cpu().numpy(), orc.orc_resize(np.ascontiguousarray(src[:270, :480]), (960, 540), interpolation=2)), (cn, k)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_remap_relative_maps(cv, orc, dtype):
    """WARP_RELATIVE_MAP through cv_hal_remap32f and mi355cv_remap (240 cases of the reference's Imgproc_RemapRelative ran on the fallback): offsets
    from the destination pixel, every map representation, nearest and bilinear, device and host images"""
    REL = 32
    src = rnd((40, 50, 3), dtype, 61)
    rng = np.random.default_rng(8)
    offx = rng.uniform(-6, 6, (33, 47)).astype(np.float32); offy = rng.uniform(-6, 6, (33, 47)).astype(np.float32)
    offx[2, 3] = 40000.0; offy[5, 6] = -40000.0
    xy = np.ascontiguousarray(np.stack([offx, offy], axis=-1))
    f1, f2 = orc.orc_convertMaps(offx, offy, "16sc2", False)
    n1, _ = orc.orc_convertMaps(offx, offy, "16sc2", True)
    n0 = cv.call_count("remap32f")
    for interp in (0, 1):
        for border, bval in [(0, 9.0), (1, 0), (2, 0), (4, 0)]:
            check(cv.remap(dev(src), dev(offx), dev(offy), interp | REL, border, bval), orc.orc_remap(src, offx, offy, interp | REL, border, bval))
            check(cv.remap(dev(src), dev(xy), None, interp | REL, border, bval), orc.orc_remapMaps(src, xy, None, interp | REL, border, bval))
            check(cv.remap(dev(src), dev(f1), dev(f2), interp | REL, border, bval), orc.orc_remapMaps(src, f1, f2, interp | REL, border, bval))
    check(cv.remap(dev(src), dev(n1), None, 0 | REL, 1, 0), orc.orc_remapMaps(src, n1, None, 0 | REL, 1, 0))
    check(cv.remap(src, offx, offy, 1 | REL, 1, 0), orc.orc_remap(src, offx, offy, 1 | REL, 1, 0))               # host arrays
    assert cv.call_count("remap32f") == n0 + 9
    big = rnd((1080, 1920), dtype, 62)
    z = torch.zeros((1080, 1920), dtype=torch.float32, device="cuda")
    assert np.array_equal(cv.remap(dev(big), z, z, 1 | REL, 1, 0).cpu().numpy(), big)                          # the relative identity


def test_area_fast_2x2_float_summation_orders(cv, orc):
    """INTER_AREA 2 x 2 on CV_32F: bit-exact incl. the columns / channel counts where the reference sums in order instead of pairwise (k_area2x2_f32's tail, the generic kernel)"""
    rng = np.random.default_rng(5)
    for shape in [(40, 46), (40, 44), (22, 30, 3), (22, 30, 4), (22, 30, 2), (10, 14), (540, 962), (270, 480)]:
        src = (rng.standard_normal(shape) * 10 ** rng.uniform(-3, 3, shape)).astype(np.float32)
        h, w = shape[:2]
        got = cv.resize(dev(src), (w // 2, h // 2), interpolation=3).cpu().numpy()
        assert np.array_equal(got, orc.orc_resize(src, (w // 2, h // 2), interpolation=3)), shape


def _bits(got, want):
    """bit for bit, CV_32F included: the kernel keeps the reference's order of float operations"""
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_warp_cubic_lanczos(cv, orc, dtype, cn):
    """INTER_CUBIC / INTER_LANCZOS4 in warpAffine and warpPerspective (k_warp_taps; remapBicubic / remapLanczos4 imgwarp.cpp:905-1120): every border rule incl.
    BORDER_TRANSPARENT, maps that leave the source on every side, device and host images"""
    from opencv_amd import _lib
    src = rnd((45, 61, cn) if cn > 1 else (45, 61), dtype, 515 + cn)
    P = np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]])
    for interp in (2, 4):
        for M in mats(cv, 61, 45):
            for dsize in [(61, 45), (100, 30)]:
                prev = rnd((dsize[1], dsize[0], cn) if cn > 1 else (dsize[1], dsize[0]), dtype, 9)
                for border, bval in [(0, 0.0), (0, (10, 200, 30, 77)), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0)]:
                    want = orc.orc_warpAffine(src, M, dsize, interp, border, bval, dst=prev if border == 5 else None)
                    _bits(cv.warpAffine(dev(src), M, dsize, interp | cv.WARP_INVERSE_MAP, border, bval, dst=dev(prev.copy()) if border == 5 else None), want)
        last = _lib.lib.mi355cv_lastKernel().decode()
        want_k = "k_warp_taps_lds<%d" % (4 if interp == 2 else 8) if cn != 2 else "k_warp_taps<%d>" % (4 if interp == 2 else 8)
        tile_k = None
        if dtype == np.uint8 and interp == 2 and (cn == 1 or (cn == 3 and os.environ.get("MI355CV_WARP_TAPS_TILE") == "1")) and os.environ.get("MI355CV_WARP_TAPS_TILE") != "0":
            tile_k = "k_warp8_cubic<%d>" % cn                 # CV_8UC1 bicubic: the LDS-tile sampler of warp8.h is the default since round 5 (three channels: opt-in, it is slower there);
        assert want_k in last or (tile_k and tile_k in last), last   # maps whose tiles its plan cannot take (these 61 x 45 images: some do) stay on the tap-row kernel
        if tile_k:
            big = rnd((480, 640, cn) if cn > 1 else (480, 640), dtype, 77)
            Mb = cv.getRotationMatrix2D((320.0, 240.0), 7.0, 0.95)
            _bits(cv.warpAffine(dev(big), Mb, (640, 480), interp | cv.WARP_INVERSE_MAP, 0, 0.0), orc.orc_warpAffine(big, Mb, (640, 480), interp, 0, 0.0))
            assert tile_k in _lib.lib.mi355cv_lastKernel().decode(), _lib.lib.mi355cv_lastKernel().decode()
        for border, bval in [(0, 5.0), (1, 0), (4, 0), (5, 0)]:
            prev = rnd((45, 61, cn) if cn > 1 else (45, 61), dtype, 10)
            want = orc.orc_warpPerspective(src, P, (61, 45), interp, border, bval, dst=prev if border == 5 else None)
            _bits(cv.warpPerspective(dev(src), P, (61, 45), interp | cv.WARP_INVERSE_MAP, border, bval, dst=dev(prev.copy()) if border == 5 else None), want)
        M = mats(cv, 61, 45)[1]
        _bits(cv.warpAffine(src, M, (61, 45), interp | cv.WARP_INVERSE_MAP, 4), orc.orc_warpAffine(src, M, (61, 45), interp, 4))          # host arrays


@pytest.mark.parametrize("dtype", DT)
def test_remap_cubic_lanczos(cv, orc, dtype):
    """cv::remap with INTER_CUBIC / INTER_LANCZOS4: CV_32FC1 pairs (cv_hal_remap32f), a CV_32FC2 map, the fixed-point maps -- plain and with WARP_RELATIVE_MAP
    (the 150 cases of the reference's Imgproc_RemapRelative that round 4's ledger still showed on the fallback)"""
    REL = 32
    src = rnd((40, 50, 3), dtype, 536)
    mapx, mapy = _float_maps(14)
    xy = np.ascontiguousarray(np.stack([mapx, mapy], axis=-1))
    f1, f2 = orc.orc_convertMaps(mapx, mapy, "16sc2", False)
    rng = np.random.default_rng(18)
    offx = rng.uniform(-6, 6, (33, 47)).astype(np.float32); offy = rng.uniform(-6, 6, (33, 47)).astype(np.float32)
    offx[2, 3] = 40000.0; offy[5, 6] = -40000.0
    oxy = np.ascontiguousarray(np.stack([offx, offy], axis=-1))
    o1, o2 = orc.orc_convertMaps(offx, offy, "16sc2", False)
    prev = rnd((33, 47, 3), dtype, 37)
    for interp in (2, 4):
        for border, bval in [(0, 9.0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0)]:
            d0 = lambda: dev(prev.copy()) if border == 5 else None
            p0 = prev if border == 5 else None
            if border != 5:
                _bits(cv.remap(dev(src), dev(mapx), dev(mapy), interp, border, bval), orc.orc_remap(src, mapx, mapy, interp, border, bval))
                _bits(cv.remap(dev(src), dev(offx), dev(offy), interp | REL, border, bval), orc.orc_remap(src, offx, offy, interp | REL, border, bval))
            _bits(cv.remap(dev(src), dev(xy), None, interp, border, bval, dst=d0()), orc.orc_remapMaps(src, xy, None, interp, border, bval, dst=p0))
            _bits(cv.remap(dev(src), dev(f1), dev(f2), interp, border, bval, dst=d0()), orc.orc_remapMaps(src, f1, f2, interp, border, bval, dst=p0))
            _bits(cv.remap(dev(src), dev(oxy), None, interp | REL, border, bval, dst=d0()), orc.orc_remapMaps(src, oxy, None, interp | REL, border, bval, dst=p0))
            _bits(cv.remap(dev(src), dev(o1), dev(o2), interp | REL, border, bval, dst=d0()), orc.orc_remapMaps(src, o1, o2, interp | REL, border, bval, dst=p0))
        _bits(cv.remap(src, mapx, mapy, interp, 1, 0), orc.orc_remap(src, mapx, mapy, interp, 1, 0))                     # host arrays


def test_warp_cubic_whole_frames(cv, orc):
    """a whole 1080p frame per interpolation and depth: 7-degree rotation, BORDER_REFLECT_101, against the restatement"""
    M = cv.getRotationMatrix2D((960.0, 540.0), 7.0, 0.95)
    for dtype, interp in ((np.uint8, 2), (np.uint8, 4), (np.float32, 2)):
        src = rnd((1080, 1920), dtype, 77)
        _bits(cv.warpAffine(dev(src), M, (1920, 1080), interp | cv.WARP_INVERSE_MAP, 4), orc.orc_warpAffine(src, M, (1920, 1080), interp, 4))


def test_warps_on_64f_images(cv, orc):
    """CV_64F images through k_warp64 (double sums over the float weight tables, the reference's order): warpAffine, warpPerspective, cv::remap with every map representation,
    plain and relative, every interpolation and border rule; bit for bit"""
    from opencv_amd import _lib
    REL = 32
    rng = np.random.default_rng(5)
    src = rng.random((40, 50, 3)) * 1000 - 300
    M = cv.getRotationMatrix2D((25.0, 20.0), 33.0, 1.3)
    P = np.array([[0.7, -0.3, 20.0], [0.25, 0.8, -5.0], [-1e-3, 5e-4, 1.2]])
    prev = rng.random((33, 47, 3))
    mapx, mapy = _float_maps(24)
    xy = np.ascontiguousarray(np.stack([mapx, mapy], axis=-1))
    f1, f2 = orc.orc_convertMaps(mapx, mapy, "16sc2", False)
    offx = rng.uniform(-6, 6, (33, 47)).astype(np.float32); offy = rng.uniform(-6, 6, (33, 47)).astype(np.float32)
    for interp in (0, 1, 2, 4):
        for border, bval in [(0, (1.5, -2.25, 1e3 / 3, 0)), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0)]:
            d0 = lambda: dev(prev.copy()) if border == 5 else None
            p0 = prev if border == 5 else None
            _bits(cv.warpAffine(dev(src), M, (47, 33), interp | cv.WARP_INVERSE_MAP, border, bval, dst=d0()), orc.orc_warpAffine(src, M, (47, 33), interp, border, bval, dst=p0))
            _bits(cv.warpPerspective(dev(src), P, (47, 33), interp | cv.WARP_INVERSE_MAP, border, bval, dst=d0()), orc.orc_warpPerspective(src, P, (47, 33), interp, border, bval, dst=p0))
            if border != 5:
                _bits(cv.remap(dev(src), dev(mapx), dev(mapy), interp, border, bval), orc.orc_remap(src, mapx, mapy, interp, border, bval))
                _bits(cv.remap(dev(src), dev(offx), dev(offy), interp | REL, border, bval), orc.orc_remap(src, offx, offy, interp | REL, border, bval))
            _bits(cv.remap(dev(src), dev(xy), None, interp, border, bval, dst=d0()), orc.orc_remapMaps(src, xy, None, interp, border, bval, dst=p0))
            if interp:
                _bits(cv.remap(dev(src), dev(f1), dev(f2), interp, border, bval, dst=d0()), orc.orc_remapMaps(src, f1, f2, interp, border, bval, dst=p0))
    assert "k_warp64" in _lib.lib.mi355cv_lastKernel().decode()
    _bits(cv.warpAffine(src, M, (47, 33), 2 | cv.WARP_INVERSE_MAP, 4), orc.orc_warpAffine(src, M, (47, 33), 2, 4))                     # host arrays


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cn", [5, 6, 9, 14])
def test_more_than_four_channels(cv, orc, dtype, cn):
    """cv::warpAffine / warpPerspective / remap / resize of 5-14 channel images (Imgproc_Warp.multichannel, Resize.nearest_regression_15075): nearest and bilinear
    sampling loop over the channels (border value of channel k = borderValue[k & 3], imgwarp.cpp:340 / :692); the reference asserts <= 4 channels for bicubic / Lanczos
    warps and for true INTER_AREA, and those are declined.  Against the restatement (== the reference for these channel counts, checked on the CPU when oracle/_ref is there)."""
    src = rnd((37, 53, cn), dtype, 40 + cn)
    M = mats(cv, 53, 37)[1]
    P = np.vstack([M, [1e-4, 2e-4, 1.0]])
    for interp in (0, 1, 3):                                                       # (3 = INTER_AREA: bilinear in the warps)
        for border, bval in [(0, (10, 200, 30, 77)), (1, 0), (2, 0), (3, 0), (4, 0)]:
            want = orc.orc_warpAffine(src, M, (61, 41), 1 if interp == 3 else interp, border, bval)
            check(cv.warpAffine(dev(src), M, (61, 41), interp | cv.WARP_INVERSE_MAP, border, bval), want)
            want = orc.orc_warpPerspective(src, P, (61, 41), 1 if interp == 3 else interp, border, bval)
            check(cv.warpPerspective(dev(src), P, (61, 41), interp | cv.WARP_INVERSE_MAP, border, bval), want)
            if orc.load_ref() is not None:
                assert np.array_equal(want, orc.ref_warpPerspective(src, P, (61, 41), (1 if interp == 3 else interp) | 16, border, bval))
    for interp in (2, 4):
        with pytest.raises(NotImplementedError):
            cv.warpAffine(dev(src), M, (61, 41), interp | cv.WARP_INVERSE_MAP)
    yy, xx = np.mgrid[0:41, 0:61].astype(np.float32)
    mx, my = (xx * 0.8 + yy * 0.1 - 2).astype(np.float32), (yy * 0.9 - xx * 0.05 + 1.5).astype(np.float32)
    for interp in (0, 1):
        check(cv.remap(dev(src), dev(mx), dev(my), interp, 1), orc.orc_remap(src, mx, my, interp, 1))
    modes = [0, 1, 2, 4, 6] + ([5] if dtype != np.float32 else [])                 # nearest, bilinear, bicubic, Lanczos, nearest-exact, linear-exact (integer depths)
    for interp in modes:
        for dsize in [(80, 55), (31, 23), (106, 74)]:
            want = orc.orc_resize(src, dsize, interpolation=interp)
            check(cv.resize(dev(src), dsize, interpolation=interp), want, tol=1e-5 if interp in (2, 4) else 1e-6)
            if orc.load_ref() is not None and want.dtype != np.float32:
                assert np.array_equal(want, orc.ref_resize(src, dsize, interpolation=interp)), (interp, dsize)
    s2 = np.ascontiguousarray(src[:36, :52])
    check(cv.resize(dev(s2), (26, 18), interpolation=3), orc.orc_resize(s2, (26, 18), interpolation=3))          # integer-factor area: the fast kernel, any channel count
    with pytest.raises(NotImplementedError):
        cv.resize(dev(src), (31, 23), interpolation=3)                             # true area: the reference asserts cn <= 4 (resize.cpp:4045)
