"""GPU parity (bit-exact) for the remaining integer colour hooks and the histogram-driven point operations, through
the C ABI against the oracle: 4:2:0 / 4:2:2 encoders, the 4:2:2 decoder, XYZ (8U / 16U), 16-bit packed formats, premultiplied alpha,
equalizeHist, THRESH_OTSU."""
import numpy as np
import pytest
import torch

from test_oracle_colormisc import src_for

pytestmark = pytest.mark.gpu

SIZES = [(2, 2), (34, 6), (130, 4), (642, 482), (1920, 1080)]


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def dev(a):
    return torch.from_numpy(a).cuda()


def test_misc_codes(cv, orc):
    for code in sorted(orc.MISC_CODES):
        rng = np.random.default_rng(code)
        for (w, h) in SIZES:
            src = src_for(code, w, h, rng)
            got = cv.cvtColor(dev(src), code).cpu().numpy()
            want = orc.orc_cvtColorMisc(src, code)
            assert got.shape == want.shape and np.array_equal(got, want), (code, w, h, int((got != want).sum()))
        src = src_for(code, 320, 240, rng)
        assert np.array_equal(cv.cvtColor(src, code), orc.orc_cvtColorMisc(src, code)), code              # host pointers


def test_xyz_16u_and_alpha(cv, orc):
    rng = np.random.default_rng(2)
    for code in (32, 33, 34, 35):
        src = rng.integers(0, 65536, (90, 401, 3), dtype=np.uint16)
        assert np.array_equal(cv.cvtColor(dev(src), code).cpu().numpy(), orc.orc_cvtColorMisc(src, code)), code
        ex = np.tile(np.array([[[0, 0, 255], [255, 0, 0], [0, 255, 0], [255, 255, 255], [0, 0, 0], [255, 255, 0], [3, 250, 7]]], np.uint8), (2, 7, 1))
        assert np.array_equal(cv.cvtColor(dev(ex), code).cpu().numpy(), orc.orc_cvtColorMisc(ex, code)), code
    src4 = rng.integers(0, 256, (33, 77, 4), dtype=np.uint8)
    assert np.array_equal(cv.cvtColor(dev(src4), 32).cpu().numpy(), orc.orc_cvtColorMisc(src4, 32))           # BGRA source
    v, a = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    src = np.stack([v, v[:, ::-1], 255 - v, a], axis=-1).copy()
    for code in (125, 126):
        assert np.array_equal(cv.cvtColor(dev(src), code).cpu().numpy(), orc.orc_cvtColorMisc(src, code)), code


def test_xyz_32f(cv, orc):
    """CV_32F XYZ both ways (round 5; Imgproc_ColorXYZ.accuracy declined 310 calls): bit for bit against the restatement, which tests/test_oracle_colormisc.py pins to the
    reference -- row lengths that end in 0-3 tail pixels, 4-channel sources / destinations, values outside [0, 1], host arrays"""
    rng = np.random.default_rng(32)
    for code in (32, 33, 34, 35):
        for (w, h) in [(1, 1), (3, 2), (4, 3), (5, 3), (401, 90), (1023, 7), (1920, 64)]:
            for cn in ((3, 4) if code in (32, 33) else (3,)):
                src = (rng.random((h, w, cn), dtype=np.float32) * 3 - 1).astype(np.float32)
                got = cv.cvtColor(dev(src), code).cpu().numpy()
                assert got.dtype == np.float32 and np.array_equal(got, orc.orc_cvtColorMisc(src, code)), (code, w, h, cn)
        src = rng.random((33, 77, 3), dtype=np.float32)
        assert np.array_equal(cv.cvtColor(src, code), orc.orc_cvtColorMisc(src, code)), code                      # host pointers
    x = rng.random((20, 30, 3), dtype=np.float32)
    got = cv.cvtColor(dev(x), 34, dstCn=4).cpu().numpy()                                                          # XYZ -> BGRA: alpha = 1
    assert got.shape == (20, 30, 4) and np.array_equal(got[..., :3], orc.orc_cvtColorMisc(x, 34)) and np.all(got[..., 3] == 1.0)


def test_hsv_to_bgr(cv, orc):
    rng = np.random.default_rng(12)
    for code in (54, 55, 70, 71):
        for (w, h) in [(1, 1), (31, 3), (32, 2), (70, 5), (641, 9), (1920, 1080)]:
            src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            for dcn in (3, 4):
                got = cv.cvtColor(dev(src), code, dstCn=dcn).cpu().numpy()
                assert np.array_equal(got, orc.orc_cvtHSVtoBGR(src, code, dcn, 8)), (code, w, h, dcn)


def test_two_plane_encode_round_trip(cv, orc):
    rng = np.random.default_rng(4)
    for (w, h) in [(2, 2), (130, 4), (642, 482), (1920, 1080)]:
        for scn in (3, 4):
            src = rng.integers(0, 256, (h, w, scn), dtype=np.uint8)
            for swap in (False, True):
                for nv21 in (False, True):
                    got = cv.cvtColorBGR2NV(dev(src), swap, nv21).cpu().numpy()
                    assert np.array_equal(got, orc.orc_cvtBGRtoTwoPlaneYUV(src, swap, 2 if nv21 else 1)), (w, h, scn, swap, nv21)
    # encode -> decode comes back within the quantisation of the 4:2:0 format for a flat-chroma image
    flat = np.empty((64, 64, 3), np.uint8); flat[...] = (40, 120, 200)
    back = cv.cvtColor(cv.cvtColorBGR2NV(dev(flat)), 91).cpu().numpy()             # COLOR_YUV2BGR_NV12
    assert np.abs(back.astype(int) - flat).max() <= 3


def test_equalize_hist(cv, orc):
    rng = np.random.default_rng(3)
    n0 = cv.call_count("equalize_hist")
    for (w, h) in [(1, 1), (7, 5), (64, 48), (641, 481), (1937, 1081)]:
        for lo, hi in [(0, 256), (100, 140), (17, 18)]:
            src = rng.integers(lo, hi, (h, w), dtype=np.uint8)
            assert np.array_equal(cv.equalizeHist(dev(src)).cpu().numpy(), orc.orc_equalizeHist(src)), (w, h, lo, hi)
    big = rng.integers(0, 256, (2160, 3840), dtype=np.uint8)
    sub = dev(big)[5:2005, 3:3003]                                                 # unaligned ROI view of a larger device image
    assert np.array_equal(cv.equalizeHist(sub).cpu().numpy(), orc.orc_equalizeHist(np.ascontiguousarray(big[5:2005, 3:3003])))
    host = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    assert np.array_equal(cv.equalizeHist(host), orc.orc_equalizeHist(host))
    assert cv.call_count("equalize_hist") > n0


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_threshold_otsu(cv, orc, dtype):
    rng = np.random.default_rng(11)
    top = 256 if dtype == np.uint8 else 65536
    for (w, h) in [(9, 7), (320, 240), (1283, 721)]:
        for mode in range(4):
            if mode == 0: src = rng.integers(0, top, (h, w)).astype(dtype)
            elif mode == 1: src = np.where(rng.random((h, w)) < 0.3, rng.integers(top // 8, top // 4, (h, w)), rng.integers(top // 2, top - 1, (h, w))).astype(dtype)
            elif mode == 2: src = np.full((h, w), top // 3, dtype)
            else: src = np.where(rng.random((h, w)) < 0.5, 0, top - 1).astype(dtype)
            for ttype in range(5):
                tv, td = orc.orc_thresholdOtsu(src, 200.4, ttype)
                gv, gd = cv.threshold(dev(src), 0, 200.4, ttype | cv.THRESH_OTSU)
                assert gv == tv and np.array_equal(gd.cpu().numpy(), td), (dtype, w, h, mode, ttype, gv, tv)
    # THRESH_DRYRUN through the C ABI (thresh.cpp:1550-1557 hands the hook an EMPTY destination): the level alone, nothing written
    import ctypes
    from opencv_amd import _lib
    src = rng.integers(0, top, (240, 320)).astype(dtype)
    d = dev(src)
    level = ctypes.c_double(-1)
    rc = _lib.lib.mi355cv_threshold_otsu(ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(d.stride(0) * d.element_size()), None, ctypes.c_size_t(0), 320, 240,
                                         0 if dtype == np.uint8 else 2, ctypes.c_double(255.0), 0, ctypes.byref(level))
    assert rc == 0 and level.value == orc.orc_thresholdOtsu(src, 255.0, 0)[0]


@pytest.mark.parametrize("code", [52, 53, 68, 69, 60, 61, 72, 73])
def test_hls_8u(cv, orc, code):
    """BGR/RGB(A) <-> HLS, CV_8U (declined, Imgproc_ColorHLS ran on the fallback): bit-exact against the restatement that tests/test_oracle_hls.py pins to
    the reference on all 2^24 inputs -- rows that split into the reference's vector body and scalar tail at every 256-pixel block, device and host images"""
    from opencv_amd import _lib
    rng = np.random.default_rng(code)
    fwd = code in orc._HLS_FWD
    n0 = cv.call_count("cvtBGRtoHSV" if fwd else "cvtHSVtoBGR")
    for (w, h, cn) in [(256, 9, 3), (263, 7, 3), (1000, 5, 4), (7, 5, 3), (40, 3, 4), (519, 4, 4), (1, 1, 3), (1920, 270, 3)]:
        src = rng.integers(0, 256, (h, w, cn if fwd else 3), dtype=np.uint8)
        dcn = 3 if fwd else cn
        got = cv.cvtColor(torch.from_numpy(src).cuda(), code, dstCn=dcn).cpu().numpy()
        assert np.array_equal(got, orc.orc_cvtColorHxx(src, code, dcn)), (code, w, h, cn)
        assert "HLS 8U" in _lib.lib.mi355cv_lastKernel().decode()
    src = rng.integers(0, 256, (33, 300, 3), dtype=np.uint8)
    assert np.array_equal(cv.cvtColor(src, code), orc.orc_cvtColorHxx(src, code, 3))          # host pointers
    assert cv.call_count("cvtBGRtoHSV" if fwd else "cvtHSVtoBGR") == n0 + 9
    # the ties that depend on the vector-body / scalar-tail split: every colour with b = 0 .. 255, g = 0 .. 255, r = 7 in a row of 65536 + 3 pixels
    c = np.arange(1 << 16, dtype=np.uint32)
    row = np.stack([c & 255, c >> 8, np.full_like(c, 7)], axis=-1).astype(np.uint8)
    row = np.concatenate([row, row[:3]])[None]
    assert np.array_equal(cv.cvtColor(torch.from_numpy(row).cuda(), code).cpu().numpy(), orc.orc_cvtColorHxx(row, code, 3))


@pytest.mark.parametrize("code", [52, 53, 40, 41, 60, 61, 54, 55])
def test_hls_hsv_32f(cv, orc, code):
    """CV_32F HSV and HLS, both directions: within 1e-5 (norm-relative; north_star asks 1e-4) of the restatement, which is within 1e-5 of the reference"""
    rng = np.random.default_rng(code)
    fwd = code in orc._HLS_FWD or code in orc._HSV
    for (w, h, cn) in [(263, 31, 3), (64, 5, 4), (7, 3, 3), (1920, 135, 3)]:
        if fwd:
            src = rng.random((h, w, cn), dtype=np.float32); src[0, :5] = 0.25; src[2, :2] = 0
            dcn = 3
        else:
            src = rng.random((h, w, 3), dtype=np.float32); src[..., 0] *= 359.9
            src[0, :4, 2 if code in (60, 61) else 1] = 0
            dcn = cn
        got = cv.cvtColor(torch.from_numpy(src).cuda(), code, dstCn=dcn).cpu().numpy()
        want = orc.orc_cvtColorHxx(src, code, dcn)
        assert got.shape == want.shape and orc.rel_err(got, want) <= 1e-5 and np.abs(got - want).max() <= 2e-4 * max(1.0, float(np.abs(want).max())), (code, w, h, cn, orc.rel_err(got, want))


def test_host_images_converted_in_place(cv, orc):
    """cv::cvtColor(m, m, code) on a host image (what the reference's own accuracy tests do in a quarter of their cases, the C wrappers' Mats sharing one buffer):
    the hook stages source and destination through separate device buffers, so the conversion is served and equals the out-of-place result; a DEVICE image
    converted in place is still declined (a stencil-free kernel would do, but the contract is kept narrow)"""
    rng = np.random.default_rng(41)
    src = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    for code in (cv.COLOR_BGR2RGB, cv.COLOR_BGR2YCrCb, cv.COLOR_YCrCb2BGR, cv.COLOR_BGR2HSV, 54, 52, 53, cv.COLOR_BGR2Lab, 56, 50, 58, 32, 34):
        want = cv.cvtColor(dev(src), code).cpu().numpy()                     # out of place on the device: pinned against the restatement by the other tests
        buf = src.copy()
        got = cv.cvtColor(buf, code, dst=buf)
        assert got is buf and np.array_equal(buf, want), code
    d = dev(src.copy())
    with pytest.raises(NotImplementedError):
        cv.cvtColor(d, cv.COLOR_BGR2RGB, dst=d)
