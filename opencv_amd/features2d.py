"""Host-side mirror of the reference's FAST corner detector (modules/features2d/src/fast.cpp) and of cv::ORB (modules/features2d/src/orb.cpp;
SURVEY §8 f3 "features2d detectors").

cv::FAST -> the whole detector on the GPU in one call (mi355cv_FAST); FAST_dense / FAST_NMS are the two features2d HAL hooks
(modules/features2d/src/hal_replacement.hpp:75, :87) the reference's own hal_FAST (fast.cpp:438-493) is assembled from."""
import ctypes

import numpy as np

from . import _lib
from .core import Img, bind_stream, empty_like_kind, CV_8U

__all__ = ["FAST", "FAST_dense", "FAST_NMS", "FAST_hooks", "FAST_TYPE_5_8", "FAST_TYPE_7_12", "FAST_TYPE_9_16",
           "ORB", "ORB_create", "ORB_HARRIS_SCORE", "ORB_FAST_SCORE", "KEYPOINT_DTYPE"]

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
    cap = max(1 << 14, (s.w * s.h) >> 6)               # one keypoint per 64 pixels fits without a second call on ordinary frames
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


# ---------------------------------------------------------------------------------------------------- cv::ORB (features2d.hpp:425-520, orb.cpp)
ORB_HARRIS_SCORE, ORB_FAST_SCORE = 0, 1
# cv::KeyPoint (core/types.hpp:777): pt.x, pt.y, size, angle, response, octave, class_id -- 28 bytes, the record mi355cv_KeyPoint declares
KEYPOINT_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32), ("response", np.float32),
                           ("octave", np.int32), ("class_id", np.int32)])


class _OrbParams(ctypes.Structure):
    _fields_ = [("nfeatures", ctypes.c_int), ("scaleFactor", ctypes.c_double), ("nlevels", ctypes.c_int), ("edgeThreshold", ctypes.c_int),
                ("firstLevel", ctypes.c_int), ("WTA_K", ctypes.c_int), ("scoreType", ctypes.c_int), ("patchSize", ctypes.c_int), ("fastThreshold", ctypes.c_int)]


class ORB:
    """cv::ORB (ORB_Impl, orb.cpp:655-760): the parameter set of ORB::create with its getters / setters, detect / compute / detectAndCompute.
    Keypoints are numpy records of KEYPOINT_DTYPE in the reference's order, descriptors an (n, 32) uint8 array (descriptorSize() = 32, CV_8U,
    NORM_HAMMING for WTA_K = 2, NORM_HAMMING2 otherwise).  The whole call runs on the GPU but for the two culls of the candidate lists."""

    kBytes = 32

    def __init__(self, nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2, scoreType=ORB_HARRIS_SCORE, patchSize=31, fastThreshold=20):
        if firstLevel < 0:
            raise ValueError("ORB: firstLevel >= 0 (orb.cpp:1261 CV_Assert)")
        self._p = dict(nfeatures=int(nfeatures), scaleFactor=float(np.float32(scaleFactor)), nlevels=int(nlevels), edgeThreshold=int(edgeThreshold), firstLevel=int(firstLevel),
                       WTA_K=int(WTA_K), scoreType=int(scoreType), patchSize=int(patchSize), fastThreshold=int(fastThreshold))

    # getters / setters under the reference's names (features2d.hpp:470-510)
    def descriptorSize(self): return self.kBytes
    def descriptorType(self): return CV_8U
    def defaultNorm(self): return 6 if self._p["WTA_K"] == 2 else 7                    # NORM_HAMMING / NORM_HAMMING2 (orb.cpp:700-709)
    def getDefaultName(self): return "Feature2D.ORB"

    def __getattr__(self, name):
        if name.startswith("get") and name[3:4].isupper():
            key = {"MaxFeatures": "nfeatures", "NLevels": "nlevels"}.get(name[3:], name[3].lower() + name[4:] if name[3:] != "WTA_K" else "WTA_K")
            if key in self._p:
                return lambda: self._p[key]
        if name.startswith("set") and name[3:4].isupper():
            key = {"MaxFeatures": "nfeatures", "NLevels": "nlevels"}.get(name[3:], name[3].lower() + name[4:] if name[3:] != "WTA_K" else "WTA_K")
            if key in self._p:
                def setter(v):
                    if key == "firstLevel" and v < 0:
                        raise ValueError("ORB: firstLevel >= 0")
                    self._p[key] = float(v) if key == "scaleFactor" else int(v)         # setScaleFactor(double) keeps the double; ORB::create rounds to float (orb.cpp:660, :1262)
                return setter
        raise AttributeError(name)

    def _gray(self, image):
        s = Img(image)
        if s.depth == CV_8U and s.cn in (3, 4):                                       # orb.cpp:1040-1041
            from .imgproc import cvtColor, COLOR_BGR2GRAY, COLOR_BGRA2GRAY
            image = cvtColor(image, COLOR_BGR2GRAY if s.cn == 3 else COLOR_BGRA2GRAY)
            s = Img(image)
        if s.depth != CV_8U or s.cn != 1:
            raise ValueError("ORB: CV_8UC1 (or 8-bit BGR / BGRA) image expected")
        return image, s

    def detectAndCompute(self, image, mask=None, keypoints=None, useProvidedKeypoints=False, descriptors=True):
        """ORB_Impl::detectAndCompute (orb.cpp:1012): -> (keypoints, descriptors or None)"""
        if self._p["patchSize"] < 2:
            raise ValueError("ORB: patchSize >= 2 (orb.cpp:1018 CV_Assert)")
        if (useProvidedKeypoints and not descriptors) or image is None or np.prod(np.shape(image)) == 0:      # orb.cpp:1023
            return (np.zeros(0, KEYPOINT_DTYPE) if keypoints is None else np.asarray(keypoints, KEYPOINT_DTYPE)), None
        image, s = self._gray(image)
        m = None
        if mask is not None:
            m = Img(mask)
            if m.depth != CV_8U or m.cn != 1 or (m.w, m.h) != (s.w, s.h):
                raise ValueError("ORB: the mask is a CV_8UC1 image of the image's size")
            if m.device != s.device:
                raise ValueError("ORB: image and mask live in the same kind of memory")
            bind_stream(s, m)
        else:
            bind_stream(s)
        prm = _OrbParams(**self._p)
        n_in = 0
        if useProvidedKeypoints:
            kin = np.ascontiguousarray(np.asarray(keypoints, KEYPOINT_DTYPE))
            n_in = len(kin)
        cap = max(n_in, self._p["nfeatures"] + 64, 1024)
        while True:
            kps = np.zeros(cap, KEYPOINT_DTYPE)
            if n_in:
                kps[:n_in] = kin
            desc = np.zeros((cap, 32), np.uint8) if descriptors else None
            n = L.mi355cv_ORB_detectAndCompute(_vp(s.ptr), s.step, s.w, s.h, _vp(m.ptr) if m is not None else None, m.step if m is not None else 0, ctypes.byref(prm),
                                                1 if useProvidedKeypoints else 0, kps.ctypes.data, n_in, cap, desc.ctypes.data if descriptors else None, 32)
            if n == -1:
                L.mi355cv_noteDecline(b"ORB_detectAndCompute")
                raise NotImplementedError("mi355cv_ORB_detectAndCompute: NOT_IMPLEMENTED for these arguments (%s); no CPU fallback in opencv_amd" % L.mi355cv_lastError().decode())
            if n < 0:
                raise _lib.Mi355cvError("mi355cv_ORB_detectAndCompute failed: " + L.mi355cv_lastError().decode())
            if n <= cap:
                return kps[:n].copy(), (desc[:n].copy() if descriptors else None)
            cap = n

    def detect(self, image, mask=None):
        """Feature2D::detect"""
        return self.detectAndCompute(image, mask, descriptors=False)[0]

    def compute(self, image, keypoints):
        """Feature2D::compute: keypoints too close to the border are dropped, the rest regrouped by octave; -> (keypoints, descriptors)"""
        return self.detectAndCompute(image, None, keypoints, useProvidedKeypoints=True)


def ORB_create(nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2, scoreType=ORB_HARRIS_SCORE, patchSize=31, fastThreshold=20):
    """cv::ORB::create (orb.cpp:1258-1265)"""
    return ORB(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize, fastThreshold)
