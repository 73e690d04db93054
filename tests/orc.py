"""Test-side loaders for the two CPU checkers (TEST INFRASTRUCTURE, never imported by opencv_amd):

  * oracle/liboracle.so      -- our plain-C restatement of the reference algorithms (oracle/*.c)
  * oracle/_ref/libocvref.so -- the REAL reference (core+imgproc) built by oracle/ref/Makefile,
                                driven through our C facade oracle/ref/ref_shim.cpp
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
c_sz, c_int, c_dbl, vp = ctypes.c_size_t, ctypes.c_int, ctypes.c_double, ctypes.c_void_p

CV_8U, CV_8S, CV_16U, CV_16S, CV_32S, CV_32F, CV_64F = range(7)
_NP_DEPTH = {np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.uint16): 2, np.dtype(np.int16): 3,
             np.dtype(np.int32): 4, np.dtype(np.float32): 5, np.dtype(np.float64): 6}
_DEPTH_NP = {v: k for k, v in _NP_DEPTH.items()}


def cvtype(a):
    cn = a.shape[2] if a.ndim == 3 else 1
    return _NP_DEPTH[a.dtype] + ((cn - 1) << 3)


def _build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = ctypes.CDLL(_build_oracle())
        _oracle.orc_borderInterpolate.argtypes = [c_int, c_int, c_int]
    return _oracle


_ref = False


def load_ref():
    global _ref
    if _ref is False:
        p = os.path.join(ORACLE_DIR, "_ref", "libocvref.so")
        _ref = ctypes.CDLL(p) if os.path.exists(p) else None
        if _ref is not None:
            _ref.ref_buildInformation.restype = ctypes.c_char_p
    return _ref


_ref_hal = False


def load_ref_hal():
    """the reference built WITH include/mi355cv_hal.hpp registered as its custom HAL (oracle/ref/Makefile `hal`)"""
    global _ref_hal
    if _ref_hal is False:
        p = os.path.join(ORACLE_DIR, "_ref", "libocvref_hal.so")
        _ref_hal = ctypes.CDLL(p) if os.path.exists(p) else None
    return _ref_hal


class use_ref:
    """context manager: route the ref_* helpers of this module to another build of the reference"""
    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        global _ref
        self.old = _ref
        _ref = self.lib

    def __exit__(self, *a):
        global _ref
        _ref = self.old


def P(a):
    return vp(a.ctypes.data)


def step(a):
    return c_sz(a.strides[0])


def wh(a):
    return a.shape[1], a.shape[0]


def cn_of(a):
    return a.shape[2] if a.ndim == 3 else 1


# ----------------------------------------------------------------------------- oracle wrappers
def orc_sepSmoothFixedU8(src, kx, ky, border, margins=(0, 0, 0, 0), parent=None):
    """src: HxW[xC] uint8 (a view into `parent` when margins are non-zero)."""
    o = oracle()
    dst = np.empty(src.shape, np.uint8)
    w, h = wh(src)
    kx = np.ascontiguousarray(kx, np.uint16)
    ky = np.ascontiguousarray(ky, np.uint16)
    o.orc_sepSmoothFixedU8(P(src), step(src), P(dst), step(dst), w, h, cn_of(src),
                           margins[0], margins[1], margins[2], margins[3],
                           P(kx), len(kx), P(ky), len(ky), border)
    return dst


def orc_gaussianBlurBinomialU8(src, ksize, border, margins=(0, 0, 0, 0)):
    o = oracle()
    dst = np.empty(src.shape, np.uint8)
    w, h = wh(src)
    rc = o.orc_gaussianBlurBinomialU8(P(src), step(src), P(dst), step(dst), w, h, cn_of(src),
                                      margins[0], margins[1], margins[2], margins[3], ksize, border)
    assert rc == 0
    return dst


# ----------------------------------------------------------------------------- real-reference wrappers
def ref_rng_fill(shape, dtype, seed, lo, hi):
    r = load_ref()
    a = np.zeros(shape, dtype)
    w, h = wh(a)
    rc = r.ref_rngFill(P(a), step(a), w, h, cvtype(a), ctypes.c_ulonglong(seed), c_dbl(lo), c_dbl(hi))
    assert rc == 0
    return a


def ref_GaussianBlurROI(parent, roi, ksize, sigma1=0.0, sigma2=0.0, border=4):
    """cv::GaussianBlur on the submatrix parent(Rect(x, y, w, h)): non-isolated borders read the parent's pixels around the ROI"""
    r = load_ref()
    x, y, w, h = roi
    dst = np.empty((h, w) + parent.shape[2:], parent.dtype)
    pw, ph = wh(parent)
    kw, kh = (ksize, ksize) if isinstance(ksize, int) else ksize
    rc = r.ref_GaussianBlurROI(P(parent), step(parent), pw, ph, cvtype(parent), x, y, w, h, P(dst), step(dst), kw, kh, c_dbl(sigma1), c_dbl(sigma2), border)
    assert rc == 0, rc
    return dst


def ref_GaussianBlur(src, ksize, sigma1=0.0, sigma2=0.0, border=4):
    r = load_ref()
    dst = np.empty_like(src)
    w, h = wh(src)
    kw, kh = (ksize, ksize) if isinstance(ksize, int) else ksize
    rc = r.ref_GaussianBlur(P(src), step(src), P(dst), step(dst), w, h, cvtype(src), kw, kh,
                            c_dbl(sigma1), c_dbl(sigma2), border)
    assert rc == 0, rc
    return dst


def ref_copyMakeBorder(src, top, bottom, left, right, border):
    r = load_ref()
    h, w = src.shape[:2]
    shape = (h + top + bottom, w + left + right) + src.shape[2:]
    dst = np.zeros(shape, src.dtype)
    rc = r.ref_copyMakeBorder(P(src), step(src), w, h, cvtype(src), P(dst), step(dst), top, bottom, left, right,
                              border, None)
    assert rc == 0
    return dst


def orc_getGaussianKernel(n, sigma):
    o = oracle()
    k = np.zeros(n, np.float64)
    assert o.orc_getGaussianKernel(n, c_dbl(sigma), P(k)) == 0
    return k


def orc_getGaussianKernelQ(n, sigma, bits=8):
    o = oracle()
    k = np.zeros(n, np.int64)
    assert o.orc_getGaussianKernelQ(n, c_dbl(sigma), bits, P(k)) == 0
    return k


def ref_getGaussianKernel(n, sigma):
    r = load_ref()
    k = np.zeros(n, np.float64)
    assert r.ref_getGaussianKernel(n, c_dbl(sigma), P(k)) == 0
    return k


# ----------------------------------------------------------------------------- colour
_GRAY = {6: (3, 0), 7: (3, 1), 10: (4, 0), 11: (4, 1)}
_RGB = {0: (3, 4, 0), 1: (4, 3, 0), 2: (3, 4, 1), 3: (4, 3, 1), 4: (3, 3, 1), 5: (4, 4, 1)}


def orc_cvtColor(src, code):
    o = oracle()
    h, w = src.shape[:2]
    depth = _NP_DEPTH[src.dtype]
    if code in _GRAY:
        scn, swap = _GRAY[code]
        dst = np.empty((h, w), src.dtype)
        o.orc_cvtBGRtoGray(P(src), step(src), P(dst), step(dst), w, h, depth, scn, swap)
    elif code in (8, 9):
        dcn = 3 if code == 8 else 4
        dst = np.empty((h, w, dcn), src.dtype)
        o.orc_cvtGraytoBGR(P(src), step(src), P(dst), step(dst), w, h, depth, dcn)
    else:
        scn, dcn, swap = _RGB[code]
        dst = np.empty((h, w, dcn), src.dtype)
        o.orc_cvtBGRtoBGR(P(src), step(src), P(dst), step(dst), w, h, depth, scn, dcn, swap)
    return dst


# cv::ColorConversionCodes of the YUV family handled by oracle/color_yuv.c: code -> (kind, a, b, c)
_YUV_FWD = {82: (3, 0, 0), 83: (3, 1, 0), 36: (3, 0, 1), 37: (3, 1, 1)}               # BGR2YUV, RGB2YUV, BGR2YCrCb, RGB2YCrCb -> (scn, swapBlue, isCbCr)
_YUV_INV = {84: (3, 0, 0), 85: (3, 1, 0), 38: (3, 0, 1), 39: (3, 1, 1)}               # YUV2BGR, YUV2RGB, YCrCb2BGR, YCrCb2RGB -> (dcn, swapBlue, isCbCr)
_YUV_NV = {90: (3, 1, 0), 91: (3, 0, 0), 92: (3, 1, 1), 93: (3, 0, 1), 94: (4, 1, 0), 95: (4, 0, 0), 96: (4, 1, 1), 97: (4, 0, 1)}   # *_NV12 / *_NV21 -> (dcn, swapBlue, uIdx)


# *_YV12 (98, 99, 102, 103) and *_IYUV / *_I420 (100, 101, 104, 105) -> (dcn, swapBlue, uIdx)
_YUV_3P = {98: (3, 1, 1), 99: (3, 0, 1), 100: (3, 1, 0), 101: (3, 0, 0), 102: (4, 1, 1), 103: (4, 0, 1), 104: (4, 1, 0), 105: (4, 0, 0)}


_HSV = {40: (0, 0), 41: (1, 0), 66: (0, 1), 67: (1, 1)}                       # BGR2HSV, RGB2HSV, BGR2HSV_FULL, RGB2HSV_FULL -> (swapBlue, fullRange)


def orc_cvtColorYUV(src, code):
    o = oracle()
    h, w = src.shape[:2]
    if code in _HSV:
        swap, full = _HSV[code]
        dst = np.empty((h, w, 3), np.uint8)
        o.orc_cvtBGRtoHSV8u(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], swap, full)
        return dst
    if code in _YUV_FWD:
        scn, swap, cbcr = _YUV_FWD[code]
        dst = np.empty((h, w, 3), np.uint8)
        o.orc_cvtBGRtoYUV8u(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], swap, cbcr)
    elif code in _YUV_INV:
        dcn, swap, cbcr = _YUV_INV[code]
        dst = np.empty((h, w, dcn), np.uint8)
        o.orc_cvtYUVtoBGR8u(P(src), step(src), P(dst), step(dst), w, h, dcn, swap, cbcr)
    elif code in _YUV_3P:
        dcn, swap, uidx = _YUV_3P[code]
        dh = h * 2 // 3
        dst = np.empty((dh, w, dcn), np.uint8)
        o.orc_cvtThreePlaneYUVtoBGR(P(src), step(src), P(dst), step(dst), w, dh, dcn, swap, uidx)
    else:
        dcn, swap, uidx = _YUV_NV[code]
        dh = h * 2 // 3
        dst = np.empty((dh, w, dcn), np.uint8)
        uv = src[dh:]
        o.orc_cvtTwoPlaneYUVtoBGR(P(src), step(src), vp(uv.ctypes.data), step(src), P(dst), step(dst), w, dh, dcn, swap, uidx)
    return dst


# CIE L*a*b*, CV_8U: code -> (swapBlue, srgb); BGR2Lab, RGB2Lab, LBGR2Lab, LRGB2Lab / Lab2BGR, Lab2RGB, Lab2LBGR, Lab2LRGB
_LAB_FWD = {44: (0, 1), 45: (1, 1), 74: (0, 0), 75: (1, 0)}
_LAB_INV = {56: (0, 1), 57: (1, 1), 78: (0, 0), 79: (1, 0)}


_LUV_FWD = {50: 0, 51: 1}                                                     # BGR2Luv, RGB2Luv (sRGB; the linear codes 76 / 77 take the float path)
_LUV_FWD_ALL = {50: (0, 1), 51: (1, 1), 76: (0, 0), 77: (1, 0)}
_LUV_INV = {58: (0, 1), 59: (1, 1), 80: (0, 0), 81: (1, 0)}                    # Luv2BGR, Luv2RGB, Luv2LBGR, Luv2LRGB


def orc_cvtColorLab(src, code, dcn=3):
    o = oracle()
    h, w = src.shape[:2]
    if src.dtype == np.float32 and (code in _LUV_FWD_ALL or code in _LUV_INV):
        if code in _LUV_FWD_ALL:
            swap, srgb = _LUV_FWD_ALL[code]
            dst = np.empty((h, w, 3), np.float32)
            o.orc_cvtBGRtoLuv32f(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], swap, srgb)
        else:
            swap, srgb = _LUV_INV[code]
            dst = np.empty((h, w, dcn), np.float32)
            o.orc_cvtLuvtoBGR32f(P(src), step(src), P(dst), step(dst), w, h, dcn, swap, srgb)
        return dst
    if code in (76, 77):                                          # CV_8U L*u*v* from linear RGB: the float path underneath
        dst = np.empty((h, w, 3), np.uint8)
        o.orc_cvtLBGRtoLuv8u(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], code - 76)
        return dst
    if src.dtype == np.float32:
        if code in _LAB_FWD:
            swap, srgb = _LAB_FWD[code]
            dst = np.empty((h, w, 3), np.float32)
            o.orc_cvtBGRtoLab32f(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], swap, srgb)
        else:
            swap, srgb = _LAB_INV[code]
            dst = np.empty((h, w, dcn), np.float32)
            o.orc_cvtLabtoBGR32f(P(src), step(src), P(dst), step(dst), w, h, dcn, swap, srgb)
        return dst
    if code in _LUV_FWD:
        dst = np.empty((h, w, 3), np.uint8)
        o.orc_cvtBGRtoLuv8u(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], _LUV_FWD[code])
        return dst
    if code in _LUV_INV:
        swap, srgb = _LUV_INV[code]
        dst = np.empty((h, w, dcn), np.uint8)
        o.orc_cvtLuvtoBGR8u(P(src), step(src), P(dst), step(dst), w, h, dcn, swap, srgb)
        return dst
    if code in _LAB_FWD:
        swap, srgb = _LAB_FWD[code]
        dst = np.empty((h, w, 3), np.uint8)
        o.orc_cvtBGRtoLab8u(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], swap, srgb)
    else:
        swap, srgb = _LAB_INV[code]
        dst = np.empty((h, w, dcn), np.uint8)
        o.orc_cvtLabtoBGR8u(P(src), step(src), P(dst), step(dst), w, h, dcn, swap, srgb)
    return dst


def ref_cvtColorYUV(src, code):
    r = load_ref()
    h, w = src.shape[:2]
    if code in _YUV_NV or code in _YUV_3P:
        dcn = (_YUV_NV.get(code) or _YUV_3P[code])[0]
        dst = np.empty((h * 2 // 3, w, dcn), np.uint8)
    else:
        dcn = 3 if (code in _YUV_FWD or code in _HSV or code in (54, 55, 70, 71)) else _YUV_INV[code][0]
        dst = np.empty((h, w, dcn), np.uint8)
    rc = r.ref_cvtColorSz(P(src), step(src), w, h, cvtype(src), P(dst), step(dst), w, dst.shape[0], cvtype(dst), code)
    assert rc == 0, rc
    return dst


def ref_cvtColor(src, code, dcn):
    r = load_ref()
    h, w = src.shape[:2]
    dst = np.empty((h, w) if dcn == 1 else (h, w, dcn), src.dtype)
    rc = r.ref_cvtColor(P(src), step(src), P(dst), step(dst), w, h, cvtype(src), cvtype(dst), code)
    assert rc == 0, rc
    return dst


# ----------------------------------------------------------------------------- integral
_DEPTH_NP = {0: np.uint8, 2: np.uint16, 3: np.int16, 4: np.int32, 5: np.float32, 6: np.float64}


def _integral_out(src, sdepth, sqdepth, sqsum, tilted):
    h, w = src.shape[:2]
    shape = (h + 1, w + 1) + tuple(src.shape[2:])
    S = np.full(shape, 77, _DEPTH_NP[sdepth])
    Q = np.full(shape, 77, _DEPTH_NP[sqdepth]) if sqsum else None
    T = np.full(shape, 77, _DEPTH_NP[sdepth]) if tilted else None
    return S, Q, T


def orc_integral(src, sdepth, sqdepth=6, sqsum=False, tilted=False):
    """(sum, sqsum or None, tilted or None) by oracle/integral.c"""
    o = oracle()
    S, Q, T = _integral_out(src, sdepth, sqdepth, sqsum, tilted)
    rc = o.orc_integral(_NP_DEPTH[src.dtype], sdepth, sqdepth, P(src), step(src), P(S), step(S), P(Q) if sqsum else None, step(Q) if sqsum else c_sz(0),
                        P(T) if tilted else None, step(T) if tilted else c_sz(0), src.shape[1], src.shape[0], cn_of(src))
    assert rc == 0, rc
    return S, Q, T


def ref_integral(src, sdepth, sqdepth=6, sqsum=False, tilted=False):
    """the same by cv::integral itself (oracle/_ref)"""
    r = load_ref()
    S, Q, T = _integral_out(src, sdepth, sqdepth, sqsum, tilted)
    rc = r.ref_integral3(P(src), step(src), src.shape[1], src.shape[0], cvtype(src), P(S), step(S), sdepth, P(Q) if sqsum else None, step(Q) if sqsum else c_sz(0), sqdepth,
                         P(T) if tilted else None, step(T) if tilted else c_sz(0))
    assert rc == 0, rc
    return S, Q, T


# ----------------------------------------------------------------------------- threshold
def orc_threshold(src, thresh, maxval, type):
    o = oracle()
    h, w = src.shape[:2]
    dst = np.empty_like(src)
    rv = ctypes.c_double(0)
    rc = o.orc_threshold(P(src), step(src), P(dst), step(dst), w, h, _NP_DEPTH[src.dtype], cn_of(src), ctypes.c_double(thresh), ctypes.c_double(maxval),
                         type, ctypes.byref(rv))
    assert rc == 0, rc
    return rv.value, dst


def ref_threshold(src, thresh, maxval, type):
    r = load_ref()
    h, w = src.shape[:2]
    dst = np.empty_like(src)
    rv = ctypes.c_double(0)
    rc = r.ref_threshold(P(src), step(src), P(dst), step(dst), w, h, cvtype(src), ctypes.c_double(thresh), ctypes.c_double(maxval), type, ctypes.byref(rv))
    assert rc == 0, rc
    return rv.value, dst


# ----------------------------------------------------------------------------- median
def orc_medianBlur(src, ksize):
    o = oracle()
    h, w = src.shape[:2]
    dst = np.empty_like(src)
    rc = o.orc_medianBlur(P(src), step(src), P(dst), step(dst), w, h, _NP_DEPTH[src.dtype], cn_of(src), ksize)
    assert rc == 0, rc
    return dst


def ref_medianBlur(src, ksize):
    r = load_ref()
    h, w = src.shape[:2]
    dst = np.empty_like(src)
    rc = r.ref_medianBlur(P(src), step(src), P(dst), step(dst), w, h, cvtype(src), ksize)
    assert rc == 0, rc
    return dst


# ----------------------------------------------------------------------------- morphology
def _bvp(bv):
    if bv is None:
        return (ctypes.c_double * 4)(*([1.7976931348623157e308] * 4))      # DBL_MAX x4 = morphologyDefaultBorderValue()
    v = list(np.atleast_1d(np.asarray(bv, dtype=np.float64))) + [0.0] * 4
    return (ctypes.c_double * 4)(*v[:4])


def orc_morph(op, src, kernel=None, anchor=(-1, -1), border=0, borderValue=None, roi=None, iterations=1):
    """iterations > 1: the element applied that many times, each pass on the previous pass's whole image (ocvMorph, morph.dispatch.cpp:455-460)"""
    if iterations > 1:
        out = orc_morph(op, src, kernel, anchor, border, borderValue, roi)
        for _ in range(iterations - 1):
            out = orc_morph(op, out, kernel, anchor, border, borderValue)
        return out
    o = oracle()
    view, fullW, fullH, offX, offY = _roi(src, roi)
    h, w = view.shape[:2]
    k = np.ones((3, 3), np.uint8) if kernel is None else np.ascontiguousarray(kernel, dtype=np.uint8)
    kh, kw = k.shape
    ax = kw // 2 if anchor[0] < 0 else anchor[0]
    ay = kh // 2 if anchor[1] < 0 else anchor[1]
    dst = np.empty((h, w) + view.shape[2:], view.dtype)
    rc = o.orc_morph(op, P(view), step(view), P(dst), step(dst), w, h, _NP_DEPTH[view.dtype], cn_of(view), fullW, fullH, offX, offY,
                     P(k), c_sz(k.strides[0]), kw, kh, ax, ay, border & ~16, _bvp(borderValue))
    assert rc == 0, rc
    return dst


def ref_morph(op, src, kernel=None, anchor=(-1, -1), iterations=1, border=0, borderValue=None):
    r = load_ref()
    h, w = src.shape[:2]
    dst = np.empty_like(src)
    k = None if kernel is None else np.ascontiguousarray(kernel, dtype=np.uint8)
    bv = None if borderValue is None else _bvp(borderValue)
    rc = r.ref_morph(op, P(src), step(src), P(dst), step(dst), w, h, cvtype(src), P(k) if k is not None else None,
                     c_sz(k.strides[0]) if k is not None else c_sz(0), k.shape[1] if k is not None else 0, k.shape[0] if k is not None else 0,
                     anchor[0], anchor[1], iterations, border, bv)
    assert rc == 0, rc
    return dst


def orc_adaptiveThreshold(src, maxValue, type, blockSize, C, method=0):
    o = oracle()
    h, w = src.shape
    dst = np.empty_like(src)
    rc = (o.orc_adaptiveThresholdGaussian if method else o.orc_adaptiveThresholdMean)(P(src), step(src), P(dst), step(dst), w, h, ctypes.c_double(maxValue), type, blockSize, ctypes.c_double(C))
    assert rc == 0, rc
    return dst


def ref_adaptiveThreshold(src, maxValue, method, type, blockSize, C):
    r = load_ref()
    h, w = src.shape
    dst = np.empty_like(src)
    rc = r.ref_adaptiveThreshold(P(src), step(src), P(dst), step(dst), w, h, ctypes.c_double(maxValue), method, type, blockSize, ctypes.c_double(C))
    assert rc == 0, rc
    return dst


def orc_moments(src, binary=False):
    o = oracle()
    h, w = src.shape
    m = np.zeros(10, np.float64)
    fn = o.orc_imageMomentsF if src.dtype in (np.float32, np.float64) else o.orc_imageMoments
    rc = fn(P(src), step(src), _NP_DEPTH[src.dtype], w, h, int(binary), P(m))
    assert rc == 0, rc
    return m


def ref_moments(src, binary=False):
    r = load_ref()
    h, w = src.shape
    m = np.zeros(10, np.float64)
    rc = r.ref_moments(P(src), step(src), w, h, cvtype(src), int(binary), P(m))
    assert rc == 0, rc
    return m


def orc_bilateralFilter(src, d, sigmaColor, sigmaSpace, border=4):
    o = oracle()
    h, w = src.shape[:2]
    dst = np.empty_like(src)
    if src.dtype == np.float32:
        rc = o.orc_bilateralFilter32f(P(src), step(src), P(dst), step(dst), w, h, cn_of(src), d, ctypes.c_double(sigmaColor), ctypes.c_double(sigmaSpace), border & ~16)
        assert rc == 0, rc
        return dst
    rc = o.orc_bilateralFilter8u(P(src), step(src), P(dst), step(dst), w, h, cn_of(src), d, ctypes.c_double(sigmaColor), ctypes.c_double(sigmaSpace), border & ~16)
    assert rc == 0, rc
    return dst


def ref_bilateralFilter(src, d, sigmaColor, sigmaSpace, border=4):
    r = load_ref()
    h, w = src.shape[:2]
    dst = np.empty_like(src)
    if src.dtype == np.float32:
        rc = r.ref_bilateralFilterT(P(src), step(src), P(dst), step(dst), w, h, cvtype(src), d, ctypes.c_double(sigmaColor), ctypes.c_double(sigmaSpace), border)
        assert rc == 0, rc
        return dst
    rc = r.ref_bilateralFilter(P(src), step(src), P(dst), step(dst), w, h, cn_of(src), d, ctypes.c_double(sigmaColor), ctypes.c_double(sigmaSpace), border)
    assert rc == 0, rc
    return dst


def orc_Canny(src, t1, t2, aperture=3, L2=False):
    o = oracle()
    h, w = src.shape[:2]
    dst = np.empty((h, w), np.uint8)
    rc = o.orc_Canny(P(src), step(src), P(dst), step(dst), w, h, cn_of(src), ctypes.c_double(t1), ctypes.c_double(t2), aperture, int(L2))
    assert rc == 0, rc
    return dst


def ref_Canny(src, t1, t2, aperture=3, L2=False):
    r = load_ref()
    h, w = src.shape[:2]
    dst = np.empty((h, w), np.uint8)
    rc = r.ref_Canny(P(src), step(src), P(dst), step(dst), w, h, cvtype(src), ctypes.c_double(t1), ctypes.c_double(t2), aperture, int(L2))
    assert rc == 0, rc
    return dst


# ----------------------------------------------------------------------------- linear filters
def _roi(src, roi):
    """roi = (x0, y0, w, h) inside `src` (the parent) or None -> (view, fullW, fullH, offX, offY)"""
    H, W = src.shape[:2]
    if roi is None:
        return src, W, H, 0, 0
    x0, y0, w, h = roi
    return src[y0:y0 + h, x0:x0 + w], W, H, x0, y0


def _out(view, ddepth):
    dt = _DEPTH_NP[ddepth if ddepth >= 0 else _NP_DEPTH[view.dtype]]
    return np.empty(view.shape, dt)


def orc_filter2D(src, ddepth, kernel, anchor=(-1, -1), delta=0.0, border=4, roi=None):
    o = oracle()
    v, fw, fh, ox, oy = _roi(src, roi)
    if border & 16:
        fw, fh, ox, oy = v.shape[1], v.shape[0], 0, 0
    dst = _out(v, ddepth)
    if dst.dtype == np.float64:                                                          # double kernel, double sums (oracle/filter64.c)
        k = np.ascontiguousarray(kernel, np.float64)
        rc = o.orc_filter2D64(P(v), step(v), P(dst), step(dst), v.shape[1], v.shape[0], cn_of(v), _NP_DEPTH[v.dtype], fw, fh, ox, oy, P(k), k.shape[1], k.shape[0],
                              anchor[0], anchor[1], c_dbl(delta), border & ~16)
        assert rc == 0, rc
        return dst
    k = np.ascontiguousarray(kernel, np.float32)
    o.orc_filter2D(P(v), step(v), P(dst), step(dst), v.shape[1], v.shape[0], cn_of(v), _NP_DEPTH[v.dtype], _NP_DEPTH[dst.dtype],
                   fw, fh, ox, oy, P(k), k.shape[1], k.shape[0], anchor[0], anchor[1], c_dbl(delta), border & ~16)
    return dst


def orc_sepFilter2D(src, ddepth, kx, ky, anchor=(-1, -1), delta=0.0, border=4, roi=None):
    o = oracle()
    v, fw, fh, ox, oy = _roi(src, roi)
    if border & 16:
        fw, fh, ox, oy = v.shape[1], v.shape[0], 0, 0
    dst = _out(v, ddepth)
    kx = np.ascontiguousarray(np.asarray(kx).ravel(), np.float64)
    ky = np.ascontiguousarray(np.asarray(ky).ravel(), np.float64)
    if dst.dtype == np.float64:
        rc = o.orc_sepFilter2D64(P(v), step(v), P(dst), step(dst), v.shape[1], v.shape[0], cn_of(v), _NP_DEPTH[v.dtype], fw, fh, ox, oy, P(kx), len(kx), P(ky), len(ky),
                                 anchor[0], anchor[1], c_dbl(delta), border & ~16)
        assert rc == 0, rc
        return dst
    o.orc_sepFilter2D(P(v), step(v), P(dst), step(dst), v.shape[1], v.shape[0], cn_of(v), _NP_DEPTH[v.dtype], _NP_DEPTH[dst.dtype],
                      fw, fh, ox, oy, P(kx), len(kx), P(ky), len(ky), anchor[0], anchor[1], c_dbl(delta), border & ~16)
    return dst


def orc_Sobel(src, ddepth, dx, dy, ksize=3, scale=1.0, delta=0.0, border=4, roi=None):
    o = oracle()
    v, fw, fh, ox, oy = _roi(src, roi)
    if border & 16:
        fw, fh, ox, oy = v.shape[1], v.shape[0], 0, 0
    dst = _out(v, ddepth)
    rc = o.orc_Sobel(P(v), step(v), P(dst), step(dst), v.shape[1], v.shape[0], cn_of(v), _NP_DEPTH[v.dtype], _NP_DEPTH[dst.dtype],
                     fw, fh, ox, oy, dx, dy, ksize, c_dbl(scale), c_dbl(delta), border & ~16)
    assert rc == 0
    return dst


def orc_boxFilter(src, ddepth, ksize, anchor=(-1, -1), normalize=True, border=4, roi=None):
    o = oracle()
    v, fw, fh, ox, oy = _roi(src, roi)
    if border & 16:
        fw, fh, ox, oy = v.shape[1], v.shape[0], 0, 0
    dst = _out(v, ddepth)
    o.orc_boxFilter(P(v), step(v), P(dst), step(dst), v.shape[1], v.shape[0], cn_of(v), _NP_DEPTH[v.dtype], _NP_DEPTH[dst.dtype],
                    fw, fh, ox, oy, ksize[0], ksize[1], anchor[0], anchor[1], int(normalize), border & ~16)
    return dst


def _ref_dst(src, ddepth):
    return np.empty(src.shape, _DEPTH_NP[ddepth if ddepth >= 0 else _NP_DEPTH[src.dtype]])


def ref_filter2D(src, ddepth, kernel, anchor=(-1, -1), delta=0.0, border=4):
    r = load_ref()
    dst = _ref_dst(src, ddepth)
    k = np.ascontiguousarray(kernel)
    kt = _NP_DEPTH[k.dtype]
    rc = r.ref_filter2D(P(src), step(src), P(dst), step(dst), src.shape[1], src.shape[0], cvtype(src), ddepth,
                        P(k), k.shape[1], k.shape[0], kt, anchor[0], anchor[1], c_dbl(delta), border)
    assert rc == 0, rc
    return dst


def ref_sepFilter2D(src, ddepth, kx, ky, anchor=(-1, -1), delta=0.0, border=4):
    r = load_ref()
    dst = _ref_dst(src, ddepth)
    kx = np.ascontiguousarray(np.asarray(kx).ravel(), np.float32)
    ky = np.ascontiguousarray(np.asarray(ky).ravel(), np.float32)
    rc = r.ref_sepFilter2D(P(src), step(src), P(dst), step(dst), src.shape[1], src.shape[0], cvtype(src), ddepth,
                           P(kx), len(kx), P(ky), len(ky), 5, anchor[0], anchor[1], c_dbl(delta), border)
    assert rc == 0, rc
    return dst


def ref_Sobel(src, ddepth, dx, dy, ksize=3, scale=1.0, delta=0.0, border=4):
    r = load_ref()
    dst = _ref_dst(src, ddepth)
    if ksize <= 0:
        rc = r.ref_Scharr(P(src), step(src), P(dst), step(dst), src.shape[1], src.shape[0], cvtype(src), ddepth, dx, dy,
                          c_dbl(scale), c_dbl(delta), border)
    else:
        rc = r.ref_Sobel(P(src), step(src), P(dst), step(dst), src.shape[1], src.shape[0], cvtype(src), ddepth, dx, dy, ksize,
                         c_dbl(scale), c_dbl(delta), border)
    assert rc == 0, rc
    return dst


def ref_boxFilter(src, ddepth, ksize, anchor=(-1, -1), normalize=True, border=4):
    r = load_ref()
    dst = _ref_dst(src, ddepth)
    rc = r.ref_boxFilter(P(src), step(src), P(dst), step(dst), src.shape[1], src.shape[0], cvtype(src), ddepth,
                         ksize[0], ksize[1], anchor[0], anchor[1], int(normalize), border)
    assert rc == 0, rc
    return dst


def rel_err(a, b):
    """checkNormRelative of the reference's accelerated-backend tests (ts/ocl_test.hpp:309-314)"""
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.max(np.abs(a - b)) / max(np.finfo(np.float32).eps, max(np.max(np.abs(a)), np.max(np.abs(b)))))


# ----------------------------------------------------------------------------- geometric transforms
def _dst_geom(src, dsize):
    dw, dh = dsize
    return np.empty((dh, dw) + src.shape[2:], src.dtype)


def orc_resize(src, dsize, fx=0.0, fy=0.0, interpolation=1):
    o = oracle()
    sh, sw = src.shape[:2]
    if dsize is None or dsize[0] == 0:
        dsize = (int(np.rint(sw * fx)), int(np.rint(sh * fy)))
    else:
        fx, fy = dsize[0] / sw, dsize[1] / sh
    dst = _dst_geom(src, dsize)
    rc = o.orc_resize(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], _NP_DEPTH[src.dtype], cn_of(src),
                      c_dbl(fx), c_dbl(fy), interpolation)
    assert rc == 0, "oracle does not restate this resize mode"
    return dst


def ref_resize(src, dsize, fx=0.0, fy=0.0, interpolation=1):
    r = load_ref()
    sh, sw = src.shape[:2]
    if dsize is None or dsize[0] == 0:
        dsize = (int(np.rint(sw * fx)), int(np.rint(sh * fy)))
        dst = _dst_geom(src, dsize)
        rc = r.ref_resize(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], cvtype(src), c_dbl(fx), c_dbl(fy), interpolation)
    else:
        dst = _dst_geom(src, dsize)
        rc = r.ref_resize(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], cvtype(src), c_dbl(0), c_dbl(0), interpolation)
    assert rc == 0, rc
    return dst


def _bv(borderValue):
    """cv::Scalar semantics: a bare number is (v, 0, 0, 0)"""
    bv = np.zeros(4, np.float64)
    if np.ndim(borderValue) == 0:
        bv[0] = borderValue
    else:
        bv[:len(borderValue)] = borderValue
    return bv


def orc_warpAffine(src, M, dsize, flags=1, border=0, borderValue=0.0, dst=None):
    """M maps dst -> src (i.e. already inverted / WARP_INVERSE_MAP form), like hal::warpAffine receives it.  `dst`: previous contents of the
    destination (BORDER_TRANSPARENT leaves unmapped pixels untouched); copied, not written."""
    o = oracle()
    sh, sw = src.shape[:2]
    given = dst is not None
    dst = np.ascontiguousarray(dst).copy() if given else _dst_geom(src, dsize)
    if border == 5 and not given:
        dst[...] = 0
    M = np.ascontiguousarray(M, np.float64)
    bv = _bv(borderValue)
    rc = o.orc_warpAffine(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], _NP_DEPTH[src.dtype], cn_of(src),
                          P(M), flags & 7, border, P(bv))
    assert rc == 0
    return dst


def orc_warpPerspective(src, M, dsize, flags=1, border=0, borderValue=0.0, dst=None):
    o = oracle()
    sh, sw = src.shape[:2]
    dst = _dst_geom(src, dsize) if dst is None else np.ascontiguousarray(dst).copy()
    M = np.ascontiguousarray(M, np.float64)
    bv = _bv(borderValue)
    rc = o.orc_warpPerspective(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], _NP_DEPTH[src.dtype], cn_of(src),
                               P(M), flags & 7, border, P(bv))
    assert rc == 0
    return dst


def orc_remap(src, mapx, mapy, interpolation=1, border=0, borderValue=0.0):
    o = oracle()
    sh, sw = src.shape[:2]
    dh, dw = mapx.shape
    dst = _dst_geom(src, (dw, dh))
    bv = _bv(borderValue)
    rc = o.orc_remap32f(P(src), step(src), sw, sh, P(dst), step(dst), dw, dh, _NP_DEPTH[src.dtype], cn_of(src),
                        P(mapx), step(mapx), P(mapy), step(mapy), interpolation, border, P(bv))
    assert rc == 0
    return dst


def _map_type(a):
    return cvtype(a) if a is not None else 0


def _map_kind(map1, map2):
    if map1.dtype == np.float32 and map1.ndim == 3: return 3
    if map1.dtype == np.int16 and map2 is not None: return 4
    if map1.dtype == np.int16: return 5
    raise ValueError("map types")


def orc_remapMaps(src, map1, map2, interpolation=1, border=0, borderValue=0.0, dst=None):
    """cv::remap with a CV_32FC2 map or the fixed-point maps (CV_16SC2 [+ CV_16UC1])"""
    o = oracle()
    sh, sw = src.shape[:2]
    dh, dw = map1.shape[:2]
    dst = np.ascontiguousarray(dst).copy() if dst is not None else _dst_geom(src, (dw, dh))
    bv = _bv(borderValue)
    rc = o.orc_remapMaps(P(src), step(src), sw, sh, P(dst), step(dst), dw, dh, _NP_DEPTH[src.dtype], cn_of(src), P(map1), step(map1),
                         P(map2) if map2 is not None else None, step(map2) if map2 is not None else 0, _map_kind(map1, map2), interpolation, border, P(bv))
    assert rc == 0
    return dst


def ref_remapMaps(src, map1, map2, interpolation=1, border=0, borderValue=0.0, dst=None):
    r = load_ref()
    sh, sw = src.shape[:2]
    dh, dw = map1.shape[:2]
    dst = np.ascontiguousarray(dst).copy() if dst is not None else _dst_geom(src, (dw, dh))
    bv = _bv(borderValue)
    rc = r.ref_remapMaps(P(src), step(src), sw, sh, P(dst), step(dst), dw, dh, cvtype(src), P(map1), step(map1), _map_type(map1),
                         P(map2) if map2 is not None else None, step(map2) if map2 is not None else 0, _map_type(map2), interpolation, border, P(bv))
    assert rc == 0, rc
    return dst


def _convert_maps_out(map1, dsttype, nn):
    h, w = map1.shape[:2]
    if dsttype == "16sc2":
        return np.empty((h, w, 2), np.int16), (None if nn else np.empty((h, w), np.uint16))
    if dsttype == "32fc1":
        return np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    return np.empty((h, w, 2), np.float32), None


def orc_convertMaps(map1, map2, dsttype, nn=False):
    o = oracle()
    h, w = map1.shape[:2]
    d1, d2 = _convert_maps_out(map1, dsttype, nn)
    if dsttype == "16sc2":
        o.orc_convertMapsToFixed(P(map1), step(map1), P(map2) if map2 is not None else None, step(map2) if map2 is not None else 0, 1 if map1.ndim == 3 else 0,
                                 P(d1), step(d1), P(d2) if d2 is not None else None, step(d2) if d2 is not None else 0, w, h, 1 if nn else 0)
    else:
        o.orc_convertMapsToFloat(P(map1), step(map1), P(map2) if map2 is not None else None, step(map2) if map2 is not None else 0, P(d1), step(d1),
                                 P(d2) if d2 is not None else None, step(d2) if d2 is not None else 0, 1 if dsttype == "32fc2" else 0, w, h)
    return d1, d2


def ref_convertMaps(map1, map2, dsttype, nn=False):
    r = load_ref()
    h, w = map1.shape[:2]
    d1, d2 = _convert_maps_out(map1, dsttype, nn)
    rc = r.ref_convertMaps(P(map1), step(map1), cvtype(map1), P(map2) if map2 is not None else None, step(map2) if map2 is not None else 0, _map_type(map2),
                           P(d1), step(d1), cvtype(d1), P(d2) if d2 is not None else None, step(d2) if d2 is not None else 0, _map_type(d2), w, h, 1 if nn else 0)
    assert rc == 0, rc
    return d1, d2


def orc_warpPolar(src, dsize, center, maxRadius, flags):
    o = oracle()
    sh, sw = src.shape[:2]
    dst = np.zeros((dsize[1], dsize[0]) + src.shape[2:], src.dtype)
    rc = o.orc_warpPolar(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], _NP_DEPTH[src.dtype], cn_of(src), ctypes.c_float(center[0]),
                         ctypes.c_float(center[1]), c_dbl(maxRadius), flags)
    assert rc == 0
    return dst


def ref_warpPolar(src, dsize, center, maxRadius, flags):
    r = load_ref()
    sh, sw = src.shape[:2]
    dst = np.zeros((dsize[1], dsize[0]) + src.shape[2:], src.dtype)
    rc = r.ref_warpPolar(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], cvtype(src), ctypes.c_float(center[0]), ctypes.c_float(center[1]),
                         c_dbl(maxRadius), flags)
    assert rc == 0, rc
    return dst


def ref_warpAffine(src, M, dsize, flags=1 | 16, border=0, borderValue=0.0, dst=None):
    r = load_ref()
    sh, sw = src.shape[:2]
    dst = _dst_geom(src, dsize) if dst is None else np.ascontiguousarray(dst).copy()
    M = np.ascontiguousarray(M, np.float64)
    bv = _bv(borderValue)
    rc = r.ref_warpAffine(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], cvtype(src), P(M), flags, border, P(bv))
    assert rc == 0, rc
    return dst


def ref_warpPerspective(src, M, dsize, flags=1 | 16, border=0, borderValue=0.0, dst=None):
    r = load_ref()
    sh, sw = src.shape[:2]
    dst = _dst_geom(src, dsize) if dst is None else np.ascontiguousarray(dst).copy()
    M = np.ascontiguousarray(M, np.float64)
    bv = _bv(borderValue)
    rc = r.ref_warpPerspective(P(src), step(src), sw, sh, P(dst), step(dst), dsize[0], dsize[1], cvtype(src), P(M), flags, border, P(bv))
    assert rc == 0, rc
    return dst


def ref_remap(src, mapx, mapy, interpolation=1, border=0, borderValue=0.0):
    r = load_ref()
    sh, sw = src.shape[:2]
    dh, dw = mapx.shape
    dst = _dst_geom(src, (dw, dh))
    bv = _bv(borderValue)
    rc = r.ref_remap(P(src), step(src), sw, sh, P(dst), step(dst), dw, dh, cvtype(src), P(mapx), step(mapx), P(mapy), step(mapy),
                     interpolation, border, P(bv))
    assert rc == 0, rc
    return dst


def ref_getRotationMatrix2D(center, angle, scale):
    r = load_ref()
    M = np.zeros(6, np.float64)
    assert r.ref_getRotationMatrix2D(c_dbl(center[0]), c_dbl(center[1]), c_dbl(angle), c_dbl(scale), P(M)) == 0
    return M.reshape(2, 3)


# ----------------------------------------------------------------------------- corners / pyramids
def orc_cornerHarris(src, blockSize, ksize, k, border=4):
    o = oracle()
    h, w = src.shape
    dst = np.empty((h, w), np.float32)
    assert o.orc_cornerResponse(P(src), step(src), P(dst), step(dst), w, h, _NP_DEPTH[src.dtype], blockSize, ksize, c_dbl(k), border, 1) == 0
    return dst


def orc_cornerMinEigenVal(src, blockSize, ksize=3, border=4):
    o = oracle()
    h, w = src.shape
    dst = np.empty((h, w), np.float32)
    assert o.orc_cornerResponse(P(src), step(src), P(dst), step(dst), w, h, _NP_DEPTH[src.dtype], blockSize, ksize, c_dbl(0), border, 0) == 0
    return dst


def ref_cornerHarris(src, blockSize, ksize, k, border=4, roi=None):
    r = load_ref()
    if roi is not None:
        x, y, w, h = roi
        dst = np.empty((h, w), np.float32)
        assert r.ref_cornerHarrisRoi(P(src), step(src), src.shape[1], src.shape[0], cvtype(src), x, y, w, h, P(dst), step(dst), blockSize, ksize, c_dbl(k), border) == 0
        return dst
    h, w = src.shape
    dst = np.empty((h, w), np.float32)
    assert r.ref_cornerHarris(P(src), step(src), P(dst), step(dst), w, h, cvtype(src), blockSize, ksize, c_dbl(k), border) == 0
    return dst


def ref_cornerMinEigenVal(src, blockSize, ksize=3, border=4):
    r = load_ref()
    h, w = src.shape
    dst = np.empty((h, w), np.float32)
    assert r.ref_cornerMinEigenVal(P(src), step(src), P(dst), step(dst), w, h, cvtype(src), blockSize, ksize, border) == 0
    return dst


def orc_pyrDown(src, dsize=None, border=4, margins=(0, 0, 0, 0)):
    o = oracle()
    sh, sw = src.shape[:2]
    dw, dh = ((sw + 1) // 2, (sh + 1) // 2) if dsize is None else dsize
    dst = np.empty((dh, dw) + src.shape[2:], src.dtype)
    rc = o.orc_pyrDown(P(src), step(src), sw, sh, P(dst), step(dst), dw, dh, _NP_DEPTH[src.dtype], cn_of(src),
                       margins[0], margins[1], margins[2], margins[3], border & ~16)
    assert rc == 0
    return dst


def ref_pyrDown(src, dsize=None, border=4):
    r = load_ref()
    sh, sw = src.shape[:2]
    dw, dh = ((sw + 1) // 2, (sh + 1) // 2) if dsize is None else dsize
    dst = np.empty((dh, dw) + src.shape[2:], src.dtype)
    rc = r.ref_pyrDown(P(src), step(src), sw, sh, P(dst), step(dst), dw, dh, cvtype(src), border)
    assert rc == 0, rc
    return dst


def orc_goodFeaturesToTrack(src, maxCorners, qualityLevel, minDistance, mask=None, blockSize=3, gradientSize=3, useHarris=False, k=0.04):
    o = oracle()
    h, w = src.shape
    buf = np.zeros((w * h, 2), np.float32)
    n = o.orc_goodFeaturesToTrack(P(src), step(src), w, h, _NP_DEPTH[src.dtype], P(buf), maxCorners, c_dbl(qualityLevel), c_dbl(minDistance),
                                  P(mask) if mask is not None else None, step(mask) if mask is not None else c_sz(0),
                                  blockSize, gradientSize, int(useHarris), c_dbl(k))
    assert n >= 0
    return buf[:n].copy()


def ref_goodFeaturesToTrack(src, maxCorners, qualityLevel, minDistance, blockSize=3, gradientSize=3, useHarris=False, k=0.04):
    r = load_ref()
    h, w = src.shape
    buf = np.zeros((max(maxCorners, 1) if maxCorners > 0 else w * h, 2), np.float32)
    n = r.ref_goodFeaturesToTrack(P(src), step(src), w, h, cvtype(src), P(buf), maxCorners, c_dbl(qualityLevel), c_dbl(minDistance),
                                  blockSize, gradientSize, int(useHarris), c_dbl(k))
    assert n >= 0
    return buf[:n].copy()


# ----------------------------------------------------------------------------- matchTemplate
def orc_matchTemplate(img, templ, method):
    o = oracle()
    ih, iw = img.shape[:2]
    th, tw = templ.shape[:2]
    res = np.empty((ih - th + 1, iw - tw + 1), np.float32)
    rc = o.orc_matchTemplate(P(img), step(img), iw, ih, P(templ), step(templ), tw, th, _NP_DEPTH[img.dtype], cn_of(img), P(res), step(res), method)
    assert rc == 0
    return res


def ref_matchTemplate(img, templ, method):
    r = load_ref()
    ih, iw = img.shape[:2]
    th, tw = templ.shape[:2]
    res = np.empty((ih - th + 1, iw - tw + 1), np.float32)
    rc = r.ref_matchTemplate(P(img), step(img), iw, ih, P(templ), step(templ), tw, th, cvtype(img), P(res), step(res), method)
    assert rc == 0, rc
    return res


def _mask_args(mask, cn):
    assert mask.dtype in (np.uint8, np.float32)
    return P(mask), step(mask), _NP_DEPTH[mask.dtype], cn_of(mask)


def orc_matchTemplateMask(img, templ, method, mask):
    o = oracle()
    ih, iw = img.shape[:2]
    th, tw = templ.shape[:2]
    res = np.empty((ih - th + 1, iw - tw + 1), np.float32)
    mp, ms, md, mc = _mask_args(mask, cn_of(img))
    rc = o.orc_matchTemplateMask(P(img), step(img), iw, ih, P(templ), step(templ), tw, th, _NP_DEPTH[img.dtype], cn_of(img), mp, ms, md, mc,
                                 P(res), step(res), method)
    assert rc == 0
    return res


def ref_matchTemplateMask(img, templ, method, mask):
    r = load_ref()
    ih, iw = img.shape[:2]
    th, tw = templ.shape[:2]
    res = np.empty((ih - th + 1, iw - tw + 1), np.float32)
    rc = r.ref_matchTemplateMask(P(img), step(img), iw, ih, P(templ), step(templ), tw, th, cvtype(img), P(mask), step(mask), cvtype(mask),
                                 P(res), step(res), method)
    assert rc == 0, rc
    return res


# ----------------------------------------------------------------------------- remaining colour conversions (oracle/color_misc.c)
def _misc_table():
    t = {}
    for code, swap in ((32, 0), (33, 1)): t[code] = ("to_xyz", swap)
    for code, swap in ((34, 0), (35, 1)): t[code] = ("from_xyz", swap)
    # 16-bit packed: (kind, cn, swapBlue, greenBits)
    for code, cn, swap, gb in ((12, 3, 0, 6), (13, 3, 1, 6), (16, 4, 0, 6), (17, 4, 1, 6), (22, 3, 0, 5), (23, 3, 1, 5), (26, 4, 0, 5), (27, 4, 1, 5)):
        t[code] = ("to_5x5", cn, swap, gb)
    for code, cn, swap, gb in ((14, 3, 0, 6), (15, 3, 1, 6), (18, 4, 0, 6), (19, 4, 1, 6), (24, 3, 0, 5), (25, 3, 1, 5), (28, 4, 0, 5), (29, 4, 1, 5)):
        t[code] = ("from_5x5", cn, swap, gb)
    t[20] = ("gray_to_5x5", 6); t[30] = ("gray_to_5x5", 5); t[21] = ("5x5_to_gray", 6); t[31] = ("5x5_to_gray", 5)
    # 4:2:2 decode: (dcn, swapBlue, uIdx, ycn)
    for code, dcn, swap, uidx, ycn in ((107, 3, 1, 0, 1), (108, 3, 0, 0, 1), (111, 4, 1, 0, 1), (112, 4, 0, 0, 1), (115, 3, 1, 0, 0), (116, 3, 0, 0, 0),
                                       (117, 3, 1, 1, 0), (118, 3, 0, 1, 0), (119, 4, 1, 0, 0), (120, 4, 0, 0, 0), (121, 4, 1, 1, 0), (122, 4, 0, 1, 0)):
        t[code] = ("dec422", dcn, swap, uidx, ycn)
    for code, scn, swap, uidx, ycn in ((143, 3, 1, 0, 1), (144, 3, 0, 0, 1), (145, 4, 1, 0, 1), (146, 4, 0, 0, 1), (147, 3, 1, 0, 0), (148, 3, 0, 0, 0),
                                       (149, 3, 1, 1, 0), (150, 3, 0, 1, 0), (151, 4, 1, 0, 0), (152, 4, 0, 0, 0), (153, 4, 1, 1, 0), (154, 4, 0, 1, 0)):
        t[code] = ("enc422", scn, swap, uidx, ycn)
    # 4:2:0 planar encode: (scn, swapBlue, uIdx)   I420 / IYUV -> 1, YV12 -> 2
    for code, scn, swap, uidx in ((127, 3, 1, 1), (128, 3, 0, 1), (129, 4, 1, 1), (130, 4, 0, 1), (131, 3, 1, 2), (132, 3, 0, 2), (133, 4, 1, 2), (134, 4, 0, 2)):
        t[code] = ("enc420p", scn, swap, uidx)
    t[125] = ("premul",); t[126] = ("unpremul",)
    return t


MISC_CODES = _misc_table()


def misc_dst(src, code):
    """shape / dtype of cvtColor's result for the codes in MISC_CODES"""
    k = MISC_CODES[code]
    h, w = src.shape[:2]
    kind = k[0]
    if kind == "to_xyz": return np.empty((h, w, 3), src.dtype)
    if kind == "from_xyz": return np.empty((h, w, 3), src.dtype)
    if kind in ("to_5x5", "gray_to_5x5"): return np.empty((h, w, 2), np.uint8)
    if kind == "from_5x5": return np.empty((h, w, k[1]), np.uint8)
    if kind == "5x5_to_gray": return np.empty((h, w), np.uint8)
    if kind == "dec422": return np.empty((h, w, k[1]), np.uint8)
    if kind == "enc422": return np.empty((h, w, 2), np.uint8)
    if kind == "enc420p": return np.empty((h * 3 // 2, w), np.uint8)
    return np.empty((h, w, 4), np.uint8)


def orc_cvtColorMisc(src, code):
    o = oracle()
    k = MISC_CODES[code]
    kind = k[0]
    h, w = src.shape[:2]
    dst = misc_dst(src, code)
    a = (P(src), step(src), P(dst), step(dst), w, h)
    if kind == "to_xyz": assert o.orc_cvtBGRtoXYZ(*a, _NP_DEPTH[src.dtype], src.shape[2], k[1]) == 0
    elif kind == "from_xyz": assert o.orc_cvtXYZtoBGR(*a, _NP_DEPTH[src.dtype], 3, k[1]) == 0
    elif kind == "to_5x5": o.orc_cvtBGRtoBGR5x5(*a, k[1], k[2], k[3])
    elif kind == "from_5x5": o.orc_cvtBGR5x5toBGR(*a, k[1], k[2], k[3])
    elif kind == "gray_to_5x5": o.orc_cvtGraytoBGR5x5(*a, k[1])
    elif kind == "5x5_to_gray": o.orc_cvtBGR5x5toGray(*a, k[1])
    elif kind == "dec422": o.orc_cvtOnePlaneYUVtoBGR(*a, k[1], k[2], k[3], k[4])
    elif kind == "enc422": o.orc_cvtOnePlaneBGRtoYUV(*a, k[1], k[2], k[3], k[4])
    elif kind == "enc420p": o.orc_cvtBGRtoThreePlaneYUV(*a, k[1], k[2], k[3])
    elif kind == "premul": o.orc_cvtRGBAtoMultipliedRGBA(*a)
    else: o.orc_cvtMultipliedRGBAtoRGBA(*a)
    return dst


def ref_cvtColorMisc(src, code):
    r = load_ref()
    dst = misc_dst(src, code)
    rc = r.ref_cvtColorSz(P(src), step(src), src.shape[1], src.shape[0], cvtype(src), P(dst), step(dst), dst.shape[1], dst.shape[0], cvtype(dst), code)
    assert rc == 0, rc
    return dst


def orc_cvtBGRtoTwoPlaneYUV(src, swapBlue, uIdx):
    o = oracle()
    h, w = src.shape[:2]
    dst = np.empty((h * 3 // 2, w), np.uint8)
    uv = dst[h:]
    o.orc_cvtBGRtoTwoPlaneYUV(P(src), step(src), P(dst), step(dst), vp(uv.ctypes.data), step(dst), w, h, src.shape[2], int(swapBlue), uIdx)
    return dst


def orc_equalizeHist(src):
    o = oracle()
    dst = np.empty_like(src)
    o.orc_equalizeHist(P(src), step(src), P(dst), step(dst), src.shape[1], src.shape[0])
    return dst


def ref_equalizeHist(src):
    r = load_ref()
    dst = np.empty_like(src)
    assert r.ref_equalizeHist(P(src), step(src), P(dst), step(dst), src.shape[1], src.shape[0]) == 0
    return dst


def orc_thresholdOtsu(src, maxval, type):
    o = oracle()
    dst = np.empty_like(src)
    rv = ctypes.c_double(0)
    rc = o.orc_thresholdOtsu(P(src), step(src), P(dst), step(dst), src.shape[1], src.shape[0], _NP_DEPTH[src.dtype], ctypes.c_double(maxval), type, ctypes.byref(rv))
    assert rc == 0, rc
    return rv.value, dst


def ref_cvtBGRtoTwoPlaneYUV(src, swapBlue, uIdx):
    r = load_ref()
    h, w = src.shape[:2]
    dst = np.empty((h * 3 // 2, w), np.uint8)
    assert r.ref_cvtBGRtoTwoPlaneYUV(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], int(swapBlue), uIdx) == 0
    return dst


# ----------------------------------------------------------------------------- sparse pyramidal LK (oracle/lk.c)
OPTFLOW_USE_INITIAL_FLOW, OPTFLOW_LK_GET_MIN_EIGENVALS = 4, 8


def lk_criteria(critType, maxCount, eps):
    """SparsePyrLKOpticalFlowImpl::calc lkpyramid.cpp:1386-1395 -> (maxCount, squared epsilon)"""
    maxCount = 30 if (critType & 1) == 0 else min(max(maxCount, 0), 100)
    eps = 0.01 if (critType & 2) == 0 else min(max(eps, 0.0), 10.0)
    return maxCount, eps * eps


def lk_pad(img, winW, winH, mode):
    """copyMakeBorder by the window size on every side; returns (padded, interior view)"""
    pad = ((winH, winH), (winW, winW)) + (((0, 0),) if img.ndim == 3 else ())
    p = np.pad(img, pad, mode=mode) if mode != "constant" else np.pad(img, pad, mode="constant")
    return p, p[winH:winH + img.shape[0], winW:winW + img.shape[1]]


def lk_pyramid(img, winW, winH, maxLevel, pyrdown):
    """buildOpticalFlowPyramid lkpyramid.cpp:747-843 without derivatives: list of interior views of REFLECT_101-padded levels"""
    levels = []
    cur = img
    for level in range(maxLevel + 1):
        if level:
            cur = pyrdown(levels[-1])
        levels.append(lk_pad(np.ascontiguousarray(cur), winW, winH, "reflect")[1])
        h, w = cur.shape[:2]
        if (w + 1) // 2 <= winW or (h + 1) // 2 <= winH:
            break
    return levels


def lk_drive(prev, nxt, pts, winSize, maxLevel, crit, flags, minEig, nextPts, pyrdown, scharr, tracker):
    """the level loop of SparsePyrLKOpticalFlowImpl::calc (:1397-1424) + LKTrackerInvoker's per-level point scaling (:215-231), with the
    three kernels supplied by the caller (oracle here, the MI355X hooks in the GPU tests)"""
    winW, winH = winSize
    maxCount, eps2 = lk_criteria(*crit)
    pp = lk_pyramid(prev, winW, winH, maxLevel, pyrdown)
    npyr = lk_pyramid(nxt, winW, winH, maxLevel, pyrdown)
    maxLevel = min(len(pp), len(npyr)) - 1
    n = len(pts)
    pts = np.ascontiguousarray(pts, np.float32).reshape(n, 2)
    out = np.ascontiguousarray(nextPts, np.float32).reshape(n, 2).copy() if flags & OPTFLOW_USE_INITIAL_FLOW else np.empty((n, 2), np.float32)
    status = np.ones(n, np.uint8)
    err = np.zeros(n, np.float32)
    for level in range(maxLevel, -1, -1):
        I, J = pp[level], npyr[level]
        d = scharr(I)
        dpad, dint = lk_pad(d, winW, winH, "constant")
        scale = np.float32(1.0 / (1 << level))
        prevScaled = pts * scale
        if level == maxLevel:
            out = out * scale if flags & OPTFLOW_USE_INITIAL_FLOW else prevScaled.copy()
        else:
            out = out * np.float32(2)
        out = np.ascontiguousarray(out, np.float32)
        tracker(I, dint, J, np.ascontiguousarray(prevScaled), out, status if level == 0 else None, err, winW, winH, maxCount, eps2,
                bool(flags & OPTFLOW_LK_GET_MIN_EIGENVALS), np.float32(minEig))
    return out, status, err


def orc_ScharrDeriv(img):
    o = oracle()
    h, w = img.shape[:2]
    cn = cn_of(img)
    dst = np.empty((h, w, 2 * cn), np.int16)
    o.orc_ScharrDeriv(P(img), step(img), P(dst), step(dst), w, h, cn)
    return dst


def orc_LKLevel(I, dI, J, prevPts, nextPts, status, err, winW, winH, maxCount, eps2, getMinEig, minEig):
    o = oracle()
    h, w = I.shape[:2]
    rc = o.orc_LKOpticalFlowLevel(P(I), step(I), P(dI), step(dI), P(J), step(J), w, h, cn_of(I), P(prevPts), P(nextPts), ctypes.c_size_t(len(prevPts)),
                                  P(status) if status is not None else None, P(err), winW, winH, maxCount, ctypes.c_double(eps2), int(getMinEig),
                                  ctypes.c_float(minEig))
    assert rc == 0


def orc_calcOpticalFlowPyrLK(prev, nxt, pts, winSize=(21, 21), maxLevel=3, crit=(3, 30, 0.01), flags=0, minEig=1e-4, nextPts=None):
    return lk_drive(prev, nxt, pts, winSize, maxLevel, crit, flags, minEig, nextPts, orc_pyrDown, orc_ScharrDeriv, orc_LKLevel)


def ref_calcOpticalFlowPyrLK(prev, nxt, pts, winSize=(21, 21), maxLevel=3, crit=(3, 30, 0.01), flags=0, minEig=1e-4, nextPts=None):
    r = load_ref()
    n = len(pts)
    pts = np.ascontiguousarray(pts, np.float32).reshape(n, 2)
    out = np.ascontiguousarray(nextPts, np.float32).reshape(n, 2).copy() if nextPts is not None else np.zeros((n, 2), np.float32)
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    h, w = prev.shape[:2]
    rc = r.ref_calcOpticalFlowPyrLK(P(prev), step(prev), P(nxt), step(nxt), w, h, cvtype(prev), P(pts), P(out), n, P(status), P(err), winSize[0], winSize[1],
                                    maxLevel, crit[0], crit[1], ctypes.c_double(crit[2]), flags, ctypes.c_double(minEig))
    assert rc == 0, rc
    return out, status, err


def ref_cvtColorApprox(src, code, dst):
    """cv::cvtColor(..., ALGO_HINT_APPROX) into a preallocated dst of the right shape"""
    r = load_ref()
    rc = r.ref_cvtColorApprox(P(src), step(src), src.shape[1], src.shape[0], cvtype(src), P(dst), step(dst), dst.shape[1], dst.shape[0], cvtype(dst), code)
    assert rc == 0, rc
    return dst


_HSV_INV = {54: (0, 0), 55: (1, 0), 70: (0, 1), 71: (1, 1)}           # HSV2BGR, HSV2RGB, HSV2BGR_FULL, HSV2RGB_FULL -> (swapBlue, fullRange)


def orc_cvtHSVtoBGR(src, code, dcn=3, lanes=8):
    """lanes = floats per vector of the reference build that runs (8: the AVX2 dispatch of oracle/ref and of any x86 host with AVX2)"""
    o = oracle()
    h, w = src.shape[:2]
    swap, full = _HSV_INV[code]
    dst = np.empty((h, w, dcn), np.uint8)
    o.orc_cvtHSVtoBGR8u(P(src), step(src), P(dst), step(dst), w, h, dcn, swap, full, lanes)
    return dst


# ---------------------------------------------------------------------------------- features2d FAST
def orc_FAST(src, threshold, nonmax=True, ftype=2, cap=200000):
    """keypoints of cv::FAST as an (n, 3) array of (x, y, response) in raster order"""
    o = oracle()
    h, w = src.shape
    out = np.zeros((cap, 3), np.float32)
    n = o.orc_FAST(P(src), step(src), w, h, threshold, 1 if nonmax else 0, ftype, P(out), cap)
    assert 0 <= n <= cap, n
    return out[:n].copy()


def ref_FAST(src, threshold, nonmax=True, ftype=2, cap=200000):
    r = load_ref()
    h, w = src.shape
    out = np.zeros((cap, 3), np.float32)
    n = r.ref_FAST(P(src), step(src), w, h, threshold, 1 if nonmax else 0, ftype, P(out), cap)
    assert 0 <= n <= cap, n
    return out[:n].copy()


def orc_FAST_dense(src, ftype=2):
    o = oracle()
    h, w = src.shape
    dst = np.empty_like(src)
    assert o.orc_FAST_dense(P(src), step(src), P(dst), step(dst), w, h, ftype) == 0
    return dst


def orc_FAST_nms(scores):
    o = oracle()
    h, w = scores.shape
    dst = np.empty_like(scores)
    o.orc_FAST_nms(P(scores), step(scores), P(dst), step(dst), w, h)
    return dst


def orc_cvtColorYUVwide(src, code, dcn=3):
    """BGR <-> YUV / YCrCb on CV_16U / CV_32F images (codes 82, 83, 36, 37 forward; 84, 85, 38, 39 inverse)"""
    o = oracle()
    h, w = src.shape[:2]
    fwd = {82: (0, 0), 83: (1, 0), 36: (0, 1), 37: (1, 1)}
    inv = {84: (0, 0), 85: (1, 0), 38: (0, 1), 39: (1, 1)}
    if code in fwd:
        swap, cb = fwd[code]
        dst = np.empty((h, w, 3), src.dtype)
        (o.orc_cvtBGRtoYUV16u if src.dtype == np.uint16 else o.orc_cvtBGRtoYUV32f)(P(src), step(src), P(dst), step(dst), w, h, src.shape[2], swap, cb)
    else:
        swap, cb = inv[code]
        dst = np.empty((h, w, dcn), src.dtype)
        (o.orc_cvtYUVtoBGR16u if src.dtype == np.uint16 else o.orc_cvtYUVtoBGR32f)(P(src), step(src), P(dst), step(dst), w, h, dcn, swap, cb)
    return dst


# ---------------------------------------------------------------------------------- features2d ORB
KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32), ("response", np.float32),
                     ("octave", np.int32), ("class_id", np.int32)])          # cv::KeyPoint, 28 bytes
ORB_DEFAULTS = dict(nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2, scoreType=0, patchSize=31, fastThreshold=20)


def _orb_call(fn, is_ref, src, keypoints, descriptors, cap, p, mask=None):
    h, w = src.shape[:2]
    if mask is not None:
        assert mask.shape == (h, w) and mask.dtype == np.uint8
    kps = np.zeros(cap, KP_DTYPE)
    n_in = 0
    if keypoints is not None:
        n_in = len(keypoints)
        kps = np.zeros(max(cap, n_in), KP_DTYPE)
        kps[:n_in] = keypoints
    desc = np.zeros((len(kps), 32), np.uint8)
    c_f = ctypes.c_float
    # ORB::create takes the scale factor as a float, setScaleFactor (key "setScaleFactor") as a double; the member is a double either way
    set_scale = float(p.get("setScaleFactor", 0.0))
    eff_scale = set_scale if set_scale > 0 else float(np.float32(p["scaleFactor"]))
    rest = [p["nlevels"], p["edgeThreshold"], p["firstLevel"], p["WTA_K"], p["scoreType"], p["patchSize"],
            p["fastThreshold"], 1 if keypoints is not None else 0, P(kps), n_in, len(kps), P(desc), 1 if descriptors else 0]
    tail = [p["nfeatures"], c_f(p["scaleFactor"])] + rest + [c_dbl(set_scale)] if is_ref else [p["nfeatures"], c_dbl(eff_scale)] + rest
    m_args = [P(mask), step(mask)] if mask is not None else [None, c_sz(0)]
    if is_ref:
        n = fn(P(src), step(src), w, h, cvtype(src), *m_args, *tail)
    else:
        n = fn(P(src), step(src), w, h, *m_args, *tail)
    assert 0 <= n <= len(kps), n
    return kps[:n].copy(), (desc[:n].copy() if descriptors else None)


def orc_ORB(src, keypoints=None, descriptors=True, cap=20000, mask=None, **kw):
    """cv::ORB::detectAndCompute by the restatement: (keypoints as a KP_DTYPE array in the reference's order, n x 32 descriptors)"""
    return _orb_call(oracle().orc_ORBmask, False, src, keypoints, descriptors, cap, dict(ORB_DEFAULTS, **kw), mask)


def ref_ORB(src, keypoints=None, descriptors=True, cap=20000, mask=None, **kw):
    return _orb_call(load_ref().ref_ORB, True, src, keypoints, descriptors, cap, dict(ORB_DEFAULTS, **kw), mask)


def orb_mask(w, h, seed=0):
    """a mask with hard edges, a soft (1..254) ramp that the level thresholds eat differently, and isolated holes"""
    rng = np.random.default_rng(1000 + seed)
    m = np.full((h, w), 255, np.uint8)
    m[:, : w // 5] = 0
    m[h // 3: h // 2, w // 2: w // 2 + w // 6] = np.linspace(1, 254, w // 6).astype(np.uint8)[None, :]
    m[rng.integers(0, h, 300), rng.integers(0, w, 300)] = 0
    yy, xx = np.mgrid[0:h, 0:w]
    m[(xx - 3 * w // 4) ** 2 + (yy - 2 * h // 3) ** 2 < (min(w, h) // 7) ** 2] = 0
    return m


def orb_scene(w, h, seed=0, texture=1.0):
    """a synthetic scene with corners at many scales: smoothed noise plus rectangles and discs of random grey levels"""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w)).astype(np.float32)
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16
    for _ in range(2):
        img = np.apply_along_axis(lambda r: np.convolve(np.pad(r, 2, mode="reflect"), k, mode="valid"), 1, img)
        img = np.apply_along_axis(lambda c: np.convolve(np.pad(c, 2, mode="reflect"), k, mode="valid"), 0, img)
    img = 128 + (img - 128) * texture
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(max(6, w * h // 12000)):
        cx, cy = rng.integers(0, w), rng.integers(0, h)
        s = int(rng.integers(4, max(6, min(w, h) // 5)))
        g = float(rng.integers(0, 256))
        if rng.integers(0, 2):
            a = rng.uniform(0, np.pi)
            u = (xx - cx) * np.cos(a) + (yy - cy) * np.sin(a); v = -(xx - cx) * np.sin(a) + (yy - cy) * np.cos(a)
            m = (np.abs(u) < s) & (np.abs(v) < s * rng.uniform(0.3, 1.0))
        else:
            m = (xx - cx) ** 2 + (yy - cy) ** 2 < s * s
        img[m] = 0.6 * g + 0.4 * img[m]
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


# ---------------------------------------------------------------------------------- HLS (CV_8U, CV_32F) and HSV (CV_32F)
# cv::ColorConversionCodes: code -> (swapBlue, fullRange)
_HLS_FWD = {52: (0, 0), 53: (1, 0), 68: (0, 1), 69: (1, 1)}           # BGR2HLS, RGB2HLS, BGR2HLS_FULL, RGB2HLS_FULL
_HLS_INV = {60: (0, 0), 61: (1, 0), 72: (0, 1), 73: (1, 1)}           # HLS2BGR, HLS2RGB, HLS2BGR_FULL, HLS2RGB_FULL


def orc_cvtColorHxx(src, code, dcn=3, lanes=8):
    """BGR/RGB(A) <-> HLS for CV_8U / CV_32F and <-> HSV for CV_32F (oracle/color_hls.c)"""
    o = oracle()
    h, w = src.shape[:2]
    f32 = src.dtype == np.float32
    fwd = code in _HLS_FWD or code in _HSV
    hls = code in _HLS_FWD or code in _HLS_INV
    swap, full = (_HLS_FWD.get(code) or _HLS_INV.get(code) or _HSV.get(code) or _HSV_INV[code])
    dst = np.empty((h, w, 3 if fwd else dcn), src.dtype)
    if f32:
        (o.orc_cvtBGRtoHxx32f if fwd else o.orc_cvtHxxtoBGR32f)(P(src), step(src), P(dst), step(dst), w, h, src.shape[2] if fwd else dcn, swap, int(hls))
    else:
        assert hls
        (o.orc_cvtBGRtoHLS8u if fwd else o.orc_cvtHLStoBGR8u)(P(src), step(src), P(dst), step(dst), w, h, src.shape[2] if fwd else dcn, swap, full, lanes)
    return dst
