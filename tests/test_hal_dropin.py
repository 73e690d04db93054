"""Drop-in boundary: the reference itself, rebuilt with include/mi355cv_hal.hpp registered as its custom HAL
(oracle/_ref/libocvref_hal.so, recipe oracle/ref/Makefile `hal`), calls into libmi355cv.so from its own CALL_HAL sites.

 * CPU (here): no GPU -> every hook answers CV_HAL_ERROR_NOT_IMPLEMENTED and the stock CPU path runs: results identical
   to the plain build (this is exactly what the reference's samples/hal/c_hal exists to exercise).
 * GPU (-m gpu): the same cv:: calls are served by the MI355X kernels; results identical for integer images, 1e-4 for
   CV_32F; the library's call counters prove the GPU path ran.
"""
import os

import numpy as np
import pytest

import orc as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _calls(o, src8, src8c3, srcf):
    """a tour of the hot path through the cv:: API of whatever build is active"""
    out = {}
    out["gauss5"] = o.ref_GaussianBlur(src8c3, 5, 0, 0, 4)
    out["gauss3"] = o.ref_GaussianBlur(src8, 3, 0, 0, 1)
    out["filter2d"] = o.ref_filter2D(src8, -1, np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32))
    out["sep"] = o.ref_sepFilter2D(src8c3, -1, [0.25, 0.5, 0.25], [0.25, 0.5, 0.25])
    out["sobel16s"] = o.ref_Sobel(src8, 3, 1, 0, 3)
    out["sobel32f"] = o.ref_Sobel(srcf, -1, 0, 1, 5)
    out["box"] = o.ref_boxFilter(src8c3, -1, (5, 5))
    out["gray"] = o.ref_cvtColor(src8c3, 6, 1)
    out["resize"] = o.ref_resize(src8c3, (100, 70))
    out["resize_f"] = o.ref_resize(srcf, (150, 110))
    M = o.ref_getRotationMatrix2D((64, 48), 7.0, 0.95)
    out["warp"] = o.ref_warpAffine(src8c3, M, (128, 96), 1 | 16, 0, 0.0)
    out["warp_f"] = o.ref_warpAffine(srcf, M, (128, 96), 1 | 16, 1, 0.0)
    P3 = np.array([[1.1, 0.05, -3.0], [0.02, 0.9, 4.0], [1e-4, -2e-4, 1.0]])
    out["persp"] = o.ref_warpPerspective(src8, P3, (128, 96), 1 | 16, 0, 0.0)
    out["pyr"] = o.ref_pyrDown(src8c3)
    out["harris"] = o.ref_cornerHarris(src8, 2, 3, 0.04)
    out["mt"] = o.ref_matchTemplate(src8, np.ascontiguousarray(src8[10:26, 20:52]), 3)
    out["yuv"] = o.ref_cvtColorYUV(src8c3, 82)
    out["ycrcb2rgb"] = o.ref_cvtColorYUV(src8c3, 39)
    out["nv12"] = o.ref_cvtColorYUV(src8, 91)
    out["i420"] = o.ref_cvtColorYUV(src8, 101)
    out["hsv"] = o.ref_cvtColorYUV(src8c3, 40)
    out["hsv2bgr"] = o.ref_cvtColorYUV(src8c3, 54)
    # a submatrix with real neighbours, sigma > 0: the reference skips its fixed-point branch there (smooth.dispatch.cpp:658) and runs sepFilter2D
    # with float taps; cv_hal_gaussianBlur must decline that case and the sepFilter hook (ROI offsets) must reproduce it
    out["gauss_roi_sigma"] = o.ref_GaussianBlurROI(src8c3, (9, 7, 100, 70), 5, 1.2, 1.2, 4)
    out["gauss_roi_binomial"] = o.ref_GaussianBlurROI(src8, (3, 5, 64, 48), 5, 0, 0, 4)
    out["adaptive"] = o.ref_adaptiveThreshold(src8, 255.0, 0, 0, 7, 2.0)
    out["adaptive_gauss"] = o.ref_adaptiveThreshold(src8, 255.0, 1, 1, 11, -1.5)
    out["moments"] = o.ref_moments(src8)
    out["moments_bin16"] = o.ref_moments((src8.astype(np.uint16) * 200), True)
    out["bilateral"] = o.ref_bilateralFilter(src8, 9, 50.0, 5.0)
    out["bilateral3"] = o.ref_bilateralFilter(src8c3, 5, 30.0, 2.0, 1)
    out["canny"] = o.ref_Canny(src8, 30, 90)
    out["canny3"] = o.ref_Canny(src8c3, 200, 400, 3, True)
    out["i420enc"] = o.ref_cvtColorMisc(src8c3, 128)
    out["yuy2dec"] = o.ref_cvtColorMisc(np.ascontiguousarray(src8c3[..., :2]), 116)
    out["uyvyenc"] = o.ref_cvtColorMisc(src8c3, 144)
    out["xyz"] = o.ref_cvtColorMisc(src8c3, 32)
    out["xyz2rgb"] = o.ref_cvtColorMisc(src8c3, 35)
    out["lab"] = o.ref_cvtColor(src8c3, 44, 3)
    out["lab2lrgb"] = o.ref_cvtColor(src8c3, 79, 3)
    out["luv"] = o.ref_cvtColor(src8c3, 51, 3)
    f3 = (src8c3.astype(np.float32) / 255).astype(np.float32)
    out["lab32f"] = o.ref_cvtColor(f3, 44, 3)
    out["lab32f_lin"] = o.ref_cvtColor(f3, 75, 3)
    out["lab2bgr32f"] = o.ref_cvtColor(out["lab32f"], 56, 3)
    out["luv2bgr"] = o.ref_cvtColor(src8c3, 58, 3)
    out["bgr565"] = o.ref_cvtColorMisc(src8c3, 12)
    out["bgr5552bgra"] = o.ref_cvtColorMisc(np.ascontiguousarray(src8c3[..., :2]), 28)
    out["5652gray"] = o.ref_cvtColorMisc(np.ascontiguousarray(src8c3[..., :2]), 21)
    out["gray2555"] = o.ref_cvtColorMisc(src8, 30)
    rgba = np.concatenate([src8c3, src8[..., None]], axis=-1)
    out["premul"] = o.ref_cvtColorMisc(rgba, 125)
    out["unpremul"] = o.ref_cvtColorMisc(rgba, 126)
    out["nv12enc"] = o.ref_cvtBGRtoTwoPlaneYUV(src8c3, 0, 1)
    out["equalize"] = o.ref_equalizeHist(src8)
    ov, od = o.ref_threshold(src8, 0.0, 255.0, 0 | 8)
    out["otsu"] = od; out["otsu_level"] = np.array([ov])
    shifted = np.roll(src8, (1, 2), axis=(0, 1))
    lkpts = (O.ref_rng_fill((120, 2), np.float32, 8, 0, 1) * np.float32([128, 96])).astype(np.float32)
    lk = o.ref_calcOpticalFlowPyrLK(src8, shifted, lkpts, (21, 21), 2)
    out["lk_status"] = lk[1]; out["lk_err"] = lk[2]
    out["lk_pts"] = np.where(lk[1][:, None] > 0, lk[0], 0).astype(np.float32)
    # ALGO_HINT_APPROX: the *Approx hooks are tried first; they are bound to the exact kernels
    out["approx_yuv"] = o.ref_cvtColorApprox(src8c3, 82, np.empty_like(src8c3))
    out["approx_yuv2bgr"] = o.ref_cvtColorApprox(src8c3, 84, np.empty_like(src8c3))
    out["approx_nv12"] = o.ref_cvtColorApprox(src8, 91, np.empty((64, 128, 3), np.uint8))
    out["approx_i420"] = o.ref_cvtColorApprox(src8, 101, np.empty((64, 128, 3), np.uint8))
    out["approx_i420enc"] = o.ref_cvtColorApprox(src8c3, 128, np.empty((144, 128), np.uint8))
    out["approx_yuy2"] = o.ref_cvtColorApprox(np.ascontiguousarray(src8c3[..., :2]), 116, np.empty_like(src8c3))
    out["approx_uyvy"] = o.ref_cvtColorApprox(src8c3, 144, np.empty((96, 128, 2), np.uint8))
    out["median3"] = o.ref_medianBlur(src8c3, 3)
    out["median5"] = o.ref_medianBlur(src8, 5)
    out["dilate"] = o.ref_morph(1, src8)
    out["erode5"] = o.ref_morph(0, src8c3, np.ones((5, 5), np.uint8), (-1, -1), 1, 4)
    out["thresh"] = o.ref_threshold(src8c3, 100.5, 200, 0)[1]
    out["thresh_f"] = o.ref_threshold(srcf, 0.4, 1.0, 3)[1]
    return out


def _inputs():
    src8 = O.ref_rng_fill((96, 128), np.uint8, 1, 0, 256)
    src8 = O.ref_GaussianBlur(src8, 5, 0, 0, 4)           # some structure for Harris
    src8c3 = O.ref_rng_fill((96, 128, 3), np.uint8, 2, 0, 256)
    srcf = O.ref_rng_fill((96, 128), np.float32, 3, 0, 1)
    return src8, src8c3, srcf


def _compare(a, b):
    for k in a:
        if a[k].dtype == np.float32:
            assert O.rel_err(a[k], b[k]) <= 1e-4, k
        else:
            assert np.array_equal(a[k], b[k]), k


def test_fallback_intact_without_gpu(ref):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: see test_reference_runs_on_the_gpu")
    hal = O.load_ref_hal()
    if hal is None:
        pytest.skip("oracle/_ref/libocvref_hal.so not built")
    src8, src8c3, srcf = _inputs()
    plain = _calls(O, src8, src8c3, srcf)
    with O.use_ref(hal):
        through = _calls(O, src8, src8c3, srcf)
    for k in plain:
        assert np.array_equal(plain[k], through[k]), k


@pytest.mark.gpu
def test_reference_runs_on_the_gpu(ref):
    import opencv_amd as cv
    hal = O.load_ref_hal()
    assert hal is not None, "oracle/_ref/libocvref_hal.so missing"
    src8, src8c3, srcf = _inputs()
    plain = _calls(O, src8, src8c3, srcf)
    names = ["gaussianBlurBinomial", "filter", "sepFilter", "sobel", "boxFilter", "cvtBGRtoGray", "resize", "warpAffine",
             "warpPerspective", "pyrdown", "integral", "threshold", "morph", "medianBlur", "cvtBGRtoYUV", "cvtYUVtoBGR", "cvtTwoPlaneYUVtoBGR", "cvtThreePlaneYUVtoBGR", "cvtBGRtoHSV", "cvtHSVtoBGR", "adaptiveThreshold", "bilateralFilter", "imageMoments", "canny",
             "cvtBGRtoTwoPlaneYUV", "cvtBGRtoThreePlaneYUV", "cvtOnePlaneYUVtoBGR", "cvtOnePlaneBGRtoYUV", "cvtBGRtoXYZ", "cvtXYZtoBGR", "cvtBGRtoLab", "cvtLabtoBGR", "cvtBGRtoBGR5x5",
             "cvtBGR5x5toBGR", "cvtBGR5x5toGray", "cvtGraytoBGR5x5", "cvtRGBAtoMultipliedRGBA", "cvtMultipliedRGBAtoRGBA", "equalize_hist", "threshold_otsu", "ScharrDeriv", "LKOpticalFlowLevel"]
    before = {n: cv.call_count(n) for n in names}
    with O.use_ref(hal):
        through = _calls(O, src8, src8c3, srcf)
    _compare(plain, through)
    for n in names:
        assert cv.call_count(n) > before[n], f"cv_hal_{n} was not served by the GPU"


# ----------------------------------------------------------------------------- include/mi355cv_cv.hpp (functions without a HAL hook)
def _wrap_calls(hal, src8):
    """mi355cv::cornerHarris / cornerMinEigenVal / goodFeaturesToTrack / buildPyramid / matchTemplate (cv:: signatures),
    compiled against the reference's headers into libocvref_hal.so (oracle/ref/cvwrap_shim.cpp)"""
    import ctypes
    c_int, c_sz, c_dbl, vp = ctypes.c_int, ctypes.c_size_t, ctypes.c_double, ctypes.c_void_p
    h, w = src8.shape
    out = {}
    d = np.empty((h, w), np.float32)
    assert hal.wrap_cornerHarris(O.P(src8), O.step(src8), O.P(d), O.step(d), w, h, 0, 2, 3, c_dbl(0.04), 4) == 0
    out["harris"] = d.copy()
    assert hal.wrap_cornerMinEigenVal(O.P(src8), O.step(src8), O.P(d), O.step(d), w, h, 0, 3, 3, 4) == 0
    out["mineig"] = d.copy()
    pts = np.zeros((200, 2), np.float32)
    n = hal.wrap_goodFeaturesToTrack(O.P(src8), O.step(src8), w, h, 0, pts.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 50,
                                     c_dbl(0.01), c_dbl(5.0), 3, 3, 0, c_dbl(0.04))
    assert n > 0
    out["gftt"] = pts[:n].copy()
    lv, sz = [], (w, h)
    for _ in range(3):
        sz = ((sz[0] + 1) // 2, (sz[1] + 1) // 2)
        lv.append(np.empty((sz[1], sz[0]), np.uint8))
    arr = (vp * 3)(*[l.ctypes.data for l in lv])
    assert hal.wrap_buildPyramid(O.P(src8), O.step(src8), w, h, 0, arr, 3, 4) == 0
    for i, l in enumerate(lv):
        out[f"pyr{i + 1}"] = l
    tpl = np.ascontiguousarray(src8[10:26, 20:52])
    r = np.empty((h - 16 + 1, w - 32 + 1), np.float32)
    assert hal.wrap_matchTemplate(O.P(src8), O.step(src8), w, h, O.P(tpl), O.step(tpl), 32, 16, 0, O.P(r), O.step(r), 5) == 0
    out["mt"] = r
    if hasattr(hal, "wrap_matchTemplateMask"):
        mask = np.zeros((16, 32), np.uint8); mask[:, :20] = 3
        rm = np.empty_like(r)
        assert hal.wrap_matchTemplateMask(O.P(src8), O.step(src8), w, h, O.P(tpl), O.step(tpl), 32, 16, 0, O.P(mask), O.step(mask), 0, O.P(rm), O.step(rm), 3) == 0
        out["mtmask"] = rm
    return out


def _wrap_expected(src8):
    exp = {"harris": O.ref_cornerHarris(src8, 2, 3, 0.04), "mineig": O.ref_cornerMinEigenVal(src8, 3, 3),
           "gftt": O.ref_goodFeaturesToTrack(src8, 50, 0.01, 5.0, 3, 3, False, 0.04),
           "mt": O.ref_matchTemplate(src8, np.ascontiguousarray(src8[10:26, 20:52]), 5)}
    mask = np.zeros((16, 32), np.uint8); mask[:, :20] = 3
    exp["mtmask"] = O.ref_matchTemplateMask(src8, np.ascontiguousarray(src8[10:26, 20:52]), 3, mask)
    l = src8
    for i in range(3):
        l = O.ref_pyrDown(l)
        exp[f"pyr{i + 1}"] = l
    return exp


def test_cv_signature_wrappers_fall_back_without_gpu(ref):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: see test_cv_signature_wrappers_on_the_gpu")
    hal = O.load_ref_hal()
    if hal is None or not hasattr(hal, "wrap_cornerHarris"):
        pytest.skip("oracle/_ref/libocvref_hal.so not built")
    src8, _, _ = _inputs()
    got, exp = _wrap_calls(hal, src8), _wrap_expected(src8)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k                       # the stock cv:: code ran


@pytest.mark.gpu
def test_cv_signature_wrappers_on_the_gpu(ref):
    import opencv_amd as cv
    hal = O.load_ref_hal()
    assert hal is not None and hasattr(hal, "wrap_cornerHarris"), "oracle/_ref/libocvref_hal.so missing or stale"
    src8, _, _ = _inputs()
    names = ["cornerHarris", "cornerMinEigenVal", "goodFeaturesToTrack", "buildPyramid", "matchTemplate", "matchTemplateMask"]
    before = {n: cv.call_count(n) for n in names}
    got, exp = _wrap_calls(hal, src8), _wrap_expected(src8)
    for n in names:
        assert cv.call_count(n) > before[n], f"mi355cv::{n} was not served by the GPU"
    for k in ("harris", "mineig", "mt", "mtmask"):
        assert O.rel_err(got[k], exp[k]) <= 1e-4, k
    for k in ("pyr1", "pyr2", "pyr3"):
        assert np.array_equal(got[k], exp[k]), k
    assert got["gftt"].shape == exp["gftt"].shape and set(map(tuple, got["gftt"])) == set(map(tuple, exp["gftt"]))


def _alloc_tour(hal, kind, src):
    import ctypes
    out = np.empty_like(src)
    rc = hal.wrap_frameAllocatorTour(kind, O.P(src), O.step(src), src.shape[1], src.shape[0], O.cvtype(src), O.P(out), O.step(out))
    assert rc == 0, rc
    return out


@pytest.mark.parametrize("kind", [0, 1])
def test_frame_allocator(ref, kind):
    """mi355cv::FrameAllocator (pinned / managed cv::MatAllocator): matrices it backs go through the reference's own code
    (create, copyTo, ROI clone, GaussianBlur via the HAL) and give the stock result -- on a CPU-only host through its fastMalloc branch"""
    hal = O.load_ref_hal()
    if hal is None:
        pytest.skip("oracle/_ref/libocvref_hal.so not built")
    src = O.ref_rng_fill((240, 320, 3), np.uint8, 5, 0, 256)
    want = O.ref_GaussianBlur(O.ref_GaussianBlur(src, 5, 0, 0, 4), 3, 0, 0, 1)
    assert np.array_equal(_alloc_tour(hal, kind, src), want)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1])
def test_frame_allocator_gpu(ref, kind):
    import opencv_amd as cv
    hal = O.load_ref_hal()
    assert hal is not None
    src = O.ref_rng_fill((480, 640, 3), np.uint8, 5, 0, 256)
    want = O.ref_GaussianBlur(O.ref_GaussianBlur(src, 5, 0, 0, 4), 3, 0, 0, 1)
    n0 = cv.call_count("gaussianBlurBinomial")
    assert np.array_equal(_alloc_tour(hal, kind, src), want)
    assert cv.call_count("gaussianBlurBinomial") >= n0 + 2
    import ctypes
    L = cv._lib.lib
    p = L.mi355cv_hostAlloc(1 << 20, kind)
    assert p, "hostAlloc failed on a GPU box"
    assert L.mi355cv_hostFree(p, kind) == 0


@pytest.mark.gpu
def test_device_frame_allocator_runs_in_place_gpu(ref):
    """mi355cv::FrameAllocator::Device (cv::MatAllocator handing out HBM): upload once, two cv::GaussianBlur calls of the
    HAL-enabled reference served on the matrices where they live -- no image byte crosses PCIe between upload and download -- download once"""
    import opencv_amd as cv
    hal = O.load_ref_hal()
    assert hal is not None
    src = O.ref_rng_fill((1080, 1920, 3), np.uint8, 6, 0, 256)
    want = O.ref_GaussianBlur(O.ref_GaussianBlur(src, 5, 0, 0, 4), 3, 0, 0, 1)
    out = np.empty_like(src)
    L = cv._lib.lib
    n0, b0 = cv.call_count("gaussianBlurBinomial"), L.mi355cv_stagedBytes()
    rc = hal.wrap_deviceAllocatorTour(O.P(src), O.step(src), src.shape[1], src.shape[0], O.cvtype(src), O.P(out), O.step(out))
    assert rc == 0, rc
    assert np.array_equal(out, want)
    assert cv.call_count("gaussianBlurBinomial") == n0 + 2
    assert L.mi355cv_stagedBytes() == b0, "a hook staged image bytes over PCIe although the matrices live in HBM"


def test_cv_signature_wrappers_on_a_submatrix(ref):
    """mi355cv::cornerHarris on a ROI without BORDER_ISOLATED reads the parent's real pixels exactly as cv::cornerHarris does (the fused entry
    point sees no margins, so the wrapper leaves that case to cv::); with BORDER_ISOLATED the ROI is a whole image for both"""
    hal = O.load_ref_hal()
    if hal is None:
        pytest.skip("oracle/_ref/libocvref_hal.so not built")
    import ctypes
    parent = O.ref_rng_fill((90, 120), np.uint8, 9, 0, 256)
    x, y, w, h = 10, 7, 64, 48
    for border in (4, 4 | 16, 1):
        got = np.empty((h, w), np.float32)
        rc = hal.wrap_cornerHarrisRoi(O.P(parent), O.step(parent), 120, 90, O.cvtype(parent), x, y, w, h, O.P(got), O.step(got), 2, 3, ctypes.c_double(0.04), border)
        assert rc == 0, rc
        want = O.ref_cornerHarris(parent, 2, 3, 0.04, border, roi=(x, y, w, h))
        assert O.rel_err(got, want) <= 1e-4, border


def _wrap_warp_family(hal):
    import ctypes
    src = O.ref_rng_fill((60, 80, 3), np.uint8, 3, 0, 256)
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:45, 0:70].astype(np.float32)
    mapx = (xx * 1.1 + rng.uniform(-2, 2, xx.shape)).astype(np.float32); mapy = (yy * 1.3 + rng.uniform(-2, 2, yy.shape)).astype(np.float32)
    out = {}
    for interp in (0, 1):
        d = np.zeros((45, 70, 3), np.uint8)
        rc = hal.wrap_remapFixed(O.P(src), O.step(src), 80, 60, O.cvtype(src), O.P(mapx), O.step(mapx), O.P(mapy), O.step(mapy), 70, 45, O.P(d), O.step(d), interp, 0)
        assert rc == 0, rc
        f1, f2 = O.ref_convertMaps(mapx, mapy, "16sc2", False)
        assert np.array_equal(d, O.ref_remapMaps(src, f1, f2, interp, 0, (7, 8, 9, 10))), interp
    for flags in (1 | 8, 1 | 256):
        d = np.zeros((90, 64, 3), np.uint8)
        rc = hal.wrap_warpPolar(O.P(src), O.step(src), 80, 60, O.cvtype(src), O.P(d), O.step(d), 64, 90, ctypes.c_float(40.0), ctypes.c_float(30.0), ctypes.c_double(35.0), flags)
        assert rc == 0, rc
        assert np.array_equal(d, O.ref_warpPolar(src, (64, 90), (40.0, 30.0), 35.0, flags)), flags


def test_cv_signature_warp_family(ref):
    """mi355cv::convertMaps / remap (fixed-point maps) / warpPolar with cv:: signatures equal the stock functions -- here through their fallback,
    on the GPU box (test below) through the fused entry points"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: see the _gpu variant")
    hal = O.load_ref_hal()
    if hal is None:
        pytest.skip("oracle/_ref/libocvref_hal.so not built")
    _wrap_warp_family(hal)


@pytest.mark.gpu
def test_cv_signature_warp_family_gpu(ref):
    import opencv_amd as cv
    hal = O.load_ref_hal()
    assert hal is not None
    n = {k: cv.call_count(k) for k in ("remap", "convertMaps", "warpPolar")}
    _wrap_warp_family(hal)
    for k in n:
        assert cv.call_count(k) > n[k], k


def _fast_tour(hal, through_hal_too):
    src = O.ref_GaussianBlur(O.ref_rng_fill((200, 260), np.uint8, 21, 0, 256), 3, 0, 0, 4)
    for thr, nonmax in ((10, True), (20, False), (35, True)):
        want = O.ref_FAST(src, thr, nonmax, 2)
        out = np.zeros((len(want) + 100, 3), np.float32)
        n = hal.wrap_FAST(O.P(src), O.step(src), 260, 200, thr, 1 if nonmax else 0, 2, O.P(out), len(out))       # mi355cv::FAST (cv:: signature)
        assert n == len(want) and np.array_equal(out[:n], want), (thr, nonmax, n, len(want))
        if through_hal_too:                                                                                     # cv::FAST of the HAL-enabled build
            with O.use_ref(hal):
                assert np.array_equal(O.ref_FAST(src, thr, nonmax, 2), want), (thr, nonmax)
    # a type the hooks decline: the stock path must answer
    w5 = O.ref_FAST(src, 10, True, 0)
    with O.use_ref(hal):
        assert np.array_equal(O.ref_FAST(src, 10, True, 0), w5)


def test_cv_signature_fast_wrapper(ref):
    """mi355cv::FAST and cv::FAST of the HAL-enabled build give the stock keypoint lists (here through the fallbacks)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: see the _gpu variant")
    hal = O.load_ref_hal()
    if hal is None:
        pytest.skip("oracle/_ref/libocvref_hal.so not built")
    _fast_tour(hal, True)


@pytest.mark.gpu
def test_cv_signature_fast_wrapper_gpu(ref):
    import opencv_amd as cv
    hal = O.load_ref_hal()
    assert hal is not None
    n = {k: cv.call_count(k) for k in ("FAST", "FAST_dense", "FAST_NMS")}
    _fast_tour(hal, True)
    for k in n:
        assert cv.call_count(k) > n[k], f"{k} was not served by the GPU"


def _orb_tour(hal):
    """mi355cv::ORB_create (cv::ORB's interface, include/mi355cv_cv.hpp) against the stock cv::ORB: detectAndCompute, detect, compute, with a mask;
    and cv::ORB of the HAL-enabled build itself (its resize / FAST / sepFilter calls go through the hooks)"""
    for (w, h, seed, kw, masked) in [(640, 480, 0, {}, False), (400, 300, 6, dict(firstLevel=1, nfeatures=700), True), (500, 375, 4, dict(WTA_K=3, scoreType=1), False)]:
        img = O.orb_scene(w, h, seed)
        mask = O.orb_mask(w, h, seed) if masked else None
        p = dict(O.ORB_DEFAULTS, **kw)
        wk, wd = O.ref_ORB(img, mask=mask, **kw)
        gk, gd = O._orb_call(hal.wrap_ORB, True, img, None, True, 20000, p, mask)
        assert len(wk) > 100 and gk.tobytes() == wk.tobytes() and np.array_equal(gd, wd), (w, h, kw)
        dk, _ = O._orb_call(hal.wrap_ORB, True, img, None, False, 20000, p, mask)                      # detect only
        assert dk.tobytes() == wk.tobytes()
        sub = wk[::3].copy()[::-1].copy()                                                              # compute(): out of level order
        ck, cd = O._orb_call(hal.wrap_ORB, True, img, sub, True, 20000, p)
        rk, rd = O.ref_ORB(img, keypoints=sub, **kw)
        assert ck.tobytes() == rk.tobytes() and np.array_equal(cd, rd)
        with O.use_ref(hal):                                                                           # cv::ORB of the HAL-enabled build
            hk, hd = O.ref_ORB(img, mask=mask, **kw)
        assert hk.tobytes() == wk.tobytes() and np.array_equal(hd, wd)
    # setScaleFactor(double) on the wrapper: the double goes through to the library (and to the stock object)
    img = O.orb_scene(300, 220, 31)
    p = dict(O.ORB_DEFAULTS, setScaleFactor=1.8, nlevels=5, nfeatures=900)
    wk, wd = O.ref_ORB(img, setScaleFactor=1.8, nlevels=5, nfeatures=900)
    gk, gd = O._orb_call(hal.wrap_ORB, True, img, None, True, 20000, p)
    assert gk.tobytes() == wk.tobytes() and np.array_equal(gd, wd)


def test_cv_signature_orb_wrapper(ref):
    """here (no device) mi355cv::ORB hands every call to the stock implementation it wraps: identical results"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: see the _gpu variant")
    hal = O.load_ref_hal()
    if hal is None or not hasattr(hal, "wrap_ORB"):
        pytest.skip("oracle/_ref/libocvref_hal.so not built (or built before wrap_ORB)")
    _orb_tour(hal)


@pytest.mark.gpu
def test_cv_signature_orb_wrapper_gpu(ref):
    import opencv_amd as cv
    hal = O.load_ref_hal()
    assert hal is not None and hasattr(hal, "wrap_ORB")
    n = cv.call_count("ORB_detectAndCompute")
    _orb_tour(hal)
    assert cv.call_count("ORB_detectAndCompute") >= n + 9, "mi355cv::ORB was not served by the GPU"


def _wrap_lk(hal, A, B, p, win, maxLevel, flags=0, guess=None):
    import ctypes
    n = len(p)
    out = np.array(guess, np.float32) if guess is not None else np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8); er = np.zeros(n, np.float32)
    rc = hal.wrap_calcOpticalFlowPyrLK(O.P(A), O.step(A), O.P(B), O.step(B), A.shape[1], A.shape[0], O.cvtype(A), O.P(p), O.P(out), n, O.P(st), O.P(er),
                                       win[0], win[1], maxLevel, 3, 30, ctypes.c_double(0.01), flags, ctypes.c_double(1e-4))
    assert rc == 0, rc
    return out, st, er


def test_cv_signature_lk_wrapper(ref):
    """mi355cv::calcOpticalFlowPyrLK (include/mi355cv_cv.hpp) with std::vector arguments: identical to cv::calcOpticalFlowPyrLK -- on the GPU
    box through the one-call entry point, here (no device) through its fallback to the stock function"""
    import torch
    from test_oracle_lk import frames, points, same
    hal = O.load_ref_hal()
    if hal is None:
        pytest.skip("oracle/_ref/libocvref_hal.so not built")
    for cn in (1, 3):
        A, B = frames(200, 260, cn, 13 + cn)
        p = points(200, 260, 200, 2)
        same(_wrap_lk(hal, A, B, p, (21, 21), 3), O.ref_calcOpticalFlowPyrLK(A, B, p, (21, 21), 3), ("wrapper", cn))
        guess = p + np.float32([1.5, -1.0])
        same(_wrap_lk(hal, A, B, p, (15, 15), 2, 4, guess), O.ref_calcOpticalFlowPyrLK(A, B, p, (15, 15), 2, flags=4, nextPts=guess), ("wrapper initial", cn))


@pytest.mark.gpu
def test_cv_signature_lk_wrapper_gpu(ref):
    import opencv_amd as cv
    from test_oracle_lk import frames, points, same
    hal = O.load_ref_hal()
    assert hal is not None
    A, B = frames(360, 480, 1, 17)
    p = points(360, 480, 800, 2)
    n0 = cv.call_count("calcOpticalFlowPyrLK")
    same(_wrap_lk(hal, A, B, p, (21, 21), 3), O.ref_calcOpticalFlowPyrLK(A, B, p, (21, 21), 3), "wrapper gpu")
    assert cv.call_count("calcOpticalFlowPyrLK") == n0 + 1


@pytest.mark.gpu
def test_for_each_shard_cpp_helper_gpu(ref):
    """mi355cv::forEachShard (mi355cv_cv.hpp over mi355cv_runSharded): a batch of 9 host frames through cv::GaussianBlur of the HAL-enabled build on device slots
    (0, 0, 0) -- three host threads, each bound to its slot's device, the hooks serving every call -- equals the stock build frame by frame"""
    import ctypes
    import opencv_amd as cv
    hal = O.load_ref_hal()
    assert hal is not None
    frames = O.ref_rng_fill((9 * 270, 480), np.uint8, 7, 0, 256).reshape(9, 270, 480)
    out = np.zeros_like(frames)
    n0 = cv.call_count("gaussianBlurBinomial")
    rc = hal.wrap_shardedGaussian((ctypes.c_int * 3)(0, 0, 0), 3, O.P(frames), O.P(out), 9, 480, 270, O.cvtype(frames[0]))
    assert rc == 0, (rc, cv._lib.lib.mi355cv_lastError())
    assert cv.call_count("gaussianBlurBinomial") == n0 + 9
    for i in range(9):
        assert np.array_equal(out[i], O.ref_GaussianBlur(frames[i], 5, 0, 0, 4)), i


@pytest.mark.gpu
def test_default_host_policy():
    """the library's DEFAULT (no MI355CV_HOST_POLICY in the environment) is "auto" (VERDICT r4 item 8a): on images in plain host memory the bandwidth-bound hooks decline --
    the reference's multi-threaded CPU path beats two PCIe crossings --, the heavy ones (corners, warps, single-threaded FilterEngine paths) stage and serve, and anything
    in device memory is served as before.  A process of its own: the policy is read once."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, torch, sys
        sys.path.insert(0, %r)
        import opencv_amd as cv
        # opencv_amd itself opts in to staging when it loads the library (it has no CPU path, ADVICE r5); -1 = back to what a C / C++ host of the library sees:
        # the MI355CV_HOST_POLICY variable, absent here, i.e. the built-in default
        assert cv._lib.lib.mi355cv_hostPolicy() == 1 and cv._lib.lib.mi355cv_setHostPolicy(-1) == 0 and cv._lib.lib.mi355cv_hostPolicy() == 0
        rng = np.random.default_rng(3)
        img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
        for name, fn in (("GaussianBlur", lambda: cv.GaussianBlur(img, (5, 5), 0)), ("cvtColor", lambda: cv.cvtColor(np.dstack([img] * 3), cv.COLOR_BGR2GRAY)),
                         ("threshold", lambda: cv.threshold(img, 100, 255, 0))):
            try:
                fn(); print("SERVED", name)
            except NotImplementedError:
                print("DECLINED", name)
        try:
            cv.erode(img, np.ones((3, 3), np.uint8)); print("SERVED erode3x3")
        except NotImplementedError:
            print("DECLINED erode3x3")
        cross = np.zeros((5, 5), np.uint8); cross[2, :] = 1; cross[:, 2] = 1
        r = cv.erode(img, cross); print("SERVED erode-cross", r.shape)
        r = cv.cornerHarris(img, 2, 3, 0.04); print("SERVED cornerHarris", r.shape)
        M = cv.getRotationMatrix2D((320.0, 240.0), 7.0, 0.95)
        r = cv.warpAffine(img, M, (640, 480)); print("SERVED warpAffine", r.shape)
        r = cv.GaussianBlur(torch.from_numpy(img).cuda(), (5, 5), 0); print("SERVED GaussianBlur-device", tuple(r.shape))
    """ % ROOT)
    env = dict(os.environ); env.pop("MI355CV_HOST_POLICY", None); env.pop("MI355CV_MIN_PIXELS", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout
    for name in ("GaussianBlur", "cvtColor", "threshold", "erode3x3"):
        assert "DECLINED " + name in out, out
    for name in ("cornerHarris", "warpAffine", "GaussianBlur-device", "erode-cross"):
        assert "SERVED " + name in out, out
