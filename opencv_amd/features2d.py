"""Host-side mirror of the reference's FAST corner detector (modules/features2d/src/fast.cpp; SURVEY §8 f3 "features2d detectors").

cv::FAST -> the whole detector on the GPU in one call (mi355cv_FAST); FAST_dense / FAST_NMS are the two features2d HAL hooks
(modules/features2d/src/hal_replacement.hpp:75, :87) the reference's own hal_FAST (fast.cpp:438-493) is assembled from."""
import ctypes

import numpy as np

from . import _lib
from .core import Img, bind_stream, empty_like_kind, CV_8U

__all__ = ["FAST", "FAST_dense", "FAST_NMS", "FAST_hooks", "FAST_TYPE_5_8", "FAST_TYPE_7_12", "FAST_TYPE_9_16"]

L = _lib.lib
FAST_TYPE_5_8, FAST_TYPE_7_12, FAST_TYPE_9_16 = 0, 1, 2


def _vp(p):
    return ctypes.c_void_p(p)


def FAST(image, threshold, nonmaxSuppression=True, type=FAST_TYPE_9_16):
    """cv::FAST (fast.cpp:496): keypoints as an (n, 3) float32 array of (x, y, response) in the reference's order.  The other fields of
    cv::KeyPoint are constants there (size 7, angle -1, octave 0, class_id -1)."""
    s = Img(image)
    if s.depth != CV_8U or s.cn != 1:
        raise ValueError("FAST: CV_8UC1 image expected")
    bind_stream(s)
    cap = 1 << 14
    while True:
        out = np.empty((cap, 3), np.float32)
        n = L.mi355cv_FAST(_vp(s.ptr), s.step, s.w, s.h, int(threshold), 1 if nonmaxSuppression else 0, int(type), out.ctypes.data, cap)
        if n == -1:
            raise NotImplementedError("mi355cv_FAST: NOT_IMPLEMENTED for these arguments (%s); no CPU fallback in opencv_amd" % L.mi355cv_lastError().decode())
        if n < 0:
            raise _lib.Mi355cvError("mi355cv_FAST failed: " + L.mi355cv_lastError().decode())
        if n <= cap:
            return out[:n].copy()
        cap = n


def FAST_dense(image, type=FAST_TYPE_9_16, dst=None):
    """cv_hal_FAST_dense: the score image (largest t + 1 for which the pixel is a corner at threshold t)"""
    s = Img(image)
    out = dst if dst is not None else empty_like_kind(image, s.h, s.w, 1, CV_8U)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_FAST_dense(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, int(type)), "FAST_dense")
    return out


def FAST_NMS(scores, dst=None):
    """cv_hal_FAST_NMS: 3x3 non-maximum suppression of a score image"""
    s = Img(scores)
    out = dst if dst is not None else empty_like_kind(scores, s.h, s.w, 1, CV_8U)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_FAST_NMS(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h), "FAST_NMS")
    return out


def FAST_hooks(image, threshold, nonmaxSuppression=True, type=FAST_TYPE_9_16):
    """cv::FAST the way the reference assembles it from the two hooks (hal_FAST, fast.cpp:438-493): dense scores, optional suppression, then the
    raster scan of the score image on the host"""
    if threshold > 20:
        raise NotImplementedError("hal_FAST serves thresholds <= 20 only (fast.cpp:440)")
    sc = FAST_dense(image, type)
    fin = FAST_NMS(sc) if nonmaxSuppression else sc
    a = fin.cpu().numpy() if hasattr(fin, "cpu") else np.asarray(fin)
    thr = 1 if (threshold == 0 and nonmaxSuppression) else threshold
    h, w = a.shape
    if h <= 6 or w <= 6:
        return np.zeros((0, 3), np.float32)
    inner = a[3:h - 3, 3:w - 3]
    ys, xs = np.nonzero(inner > thr)
    resp = (inner[ys, xs].astype(np.float32) - 1) if nonmaxSuppression else np.zeros(len(xs), np.float32)
    return np.stack([xs.astype(np.float32) + 3, ys.astype(np.float32) + 3, resp], axis=1)
