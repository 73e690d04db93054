"""Host-side mirror of the reference's sparse pyramidal Lucas-Kanade tracker (modules/video/src/lkpyramid.cpp).

The control flow is the reference's -- buildOpticalFlowPyramid (:747-843), the level loop of SparsePyrLKOpticalFlowImpl::calc (:1259-1425)
and the per-level point scaling of LKTrackerInvoker (:215-231); the work is done by the hooks the video module calls itself:
cv_hal_pyrdown, cv_hal_ScharrDeriv and cv_hal_LKOpticalFlowLevel (modules/video/src/hal_replacement.hpp), plus mi355cv_copyMakeBorder for
the padding.  With CUDA(ROCm) tensors nothing leaves HBM between the two input frames and the three result vectors.
"""
import ctypes

import numpy as np

from . import _lib
from .core import Img, bind_stream, torch, CV_8U, BORDER_CONSTANT, BORDER_REFLECT_101, BORDER_ISOLATED
from .imgproc import pyrDown

__all__ = ["ScharrDeriv", "copyMakeBorder", "buildOpticalFlowPyramid", "LKOpticalFlowLevel", "calcOpticalFlowPyrLK", "calcOpticalFlowPyrLK_hooks",
           "OPTFLOW_USE_INITIAL_FLOW", "OPTFLOW_LK_GET_MIN_EIGENVALS", "TERM_COUNT", "TERM_EPS"]

L = _lib.lib
OPTFLOW_USE_INITIAL_FLOW, OPTFLOW_LK_GET_MIN_EIGENVALS = 4, 8
TERM_COUNT, TERM_EPS = 1, 2


def _vp(p):
    return ctypes.c_void_p(p)


def _is_dev(a):
    return torch is not None and isinstance(a, torch.Tensor) and a.is_cuda


def _empty(ref, shape, dtype_np):
    if torch is not None and isinstance(ref, torch.Tensor):
        return torch.empty(shape, dtype={np.uint8: torch.uint8, np.int16: torch.int16, np.float32: torch.float32}[dtype_np], device=ref.device)
    return np.empty(shape, dtype_np)


def _zeros(ref, shape, dtype_np):
    out = _empty(ref, shape, dtype_np)
    out[...] = 0
    return out


def ScharrDeriv(src, dst=None):
    """calcScharrDeriv (lkpyramid.cpp:59-71) through cv_hal_ScharrDeriv: CV_8U, cn channels -> CV_16S, 2*cn channels, (dI/dx, dI/dy) interleaved."""
    s = Img(src)
    if s.depth != CV_8U:
        raise ValueError("ScharrDeriv: CV_8U only")                                            # CV_Assert(depth == CV_8U), :64
    out = dst if dst is not None else _empty(src, (s.h, s.w, 2 * s.cn), np.int16)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_ScharrDeriv(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.cn), "ScharrDeriv")
    return out


def copyMakeBorder(src, top, bottom, left, right, borderType, dst=None):
    """cv::copyMakeBorder (core/src/copy.cpp:1183) for device-resident images, BORDER_CONSTANT value 0.  `dst` may be the array `src` is the
    interior of (then only the frame is written, as buildOpticalFlowPyramid does)."""
    s = Img(src)
    shape = (s.h + top + bottom, s.w + left + right) + ((s.cn,) if getattr(src, "ndim", 2) == 3 else ())
    out = dst if dst is not None else torch.empty(shape, dtype=src.dtype, device=src.device)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_copyMakeBorder(_vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, top, bottom, left, right, s.cn * s.esz, int(borderType)),
               "copyMakeBorder")
    return out


def _padded_level(img, winW, winH):
    """a fresh array with a (winH, winW) frame around room for `img`; returns (whole, interior view)"""
    s = Img(img)
    shape = (s.h + 2 * winH, s.w + 2 * winW) + ((s.cn,) if getattr(img, "ndim", 2) == 3 else ())
    whole = _empty(img, shape, np.uint8)
    return whole, whole[winH:winH + s.h, winW:winW + s.w]


def _frame(whole, interior, winW, winH, borderType):
    if _is_dev(whole):
        copyMakeBorder(interior, winH, winH, winW, winW, borderType | BORDER_ISOLATED, dst=whole)
    else:                                                                                      # host arrays: the CPU's job (the hook declines host pointers)
        pad = ((winH, winH), (winW, winW)) + (((0, 0),) if interior.ndim == 3 else ())
        src = interior.numpy() if torch is not None and isinstance(interior, torch.Tensor) else interior
        whole[...] = torch.from_numpy(np.pad(src, pad, mode="reflect")) if torch is not None and isinstance(whole, torch.Tensor) else np.pad(src, pad, mode="reflect")


def buildOpticalFlowPyramid(img, winSize, maxLevel):
    """cv::buildOpticalFlowPyramid (lkpyramid.cpp:747-843) with withDerivatives = false, pyrBorder = BORDER_REFLECT_101: the list of
    level images, each one the interior view of an array padded by winSize on every side.  Stops early like the reference (:836-840)."""
    winW, winH = winSize
    if Img(img).depth != CV_8U or winW <= 2 or winH <= 2:
        raise ValueError("buildOpticalFlowPyramid: CV_8U image, winSize > 2")                  # CV_Assert, :753
    levels = []
    for level in range(maxLevel + 1):
        if level == 0:
            whole, inner = _padded_level(img, winW, winH)
            inner[...] = img
        else:
            prev = levels[-1]
            ph, pw = int(prev.shape[0]), int(prev.shape[1])
            sz = ((pw + 1) // 2, (ph + 1) // 2)
            shape = (sz[1] + 2 * winH, sz[0] + 2 * winW) + ((int(prev.shape[2]),) if prev.ndim == 3 else ())
            whole = _empty(prev, shape, np.uint8)
            inner = whole[winH:winH + sz[1], winW:winW + sz[0]]
            pyrDown(prev, sz, dst=inner)
        _frame(whole, inner, winW, winH, BORDER_REFLECT_101)
        levels.append(inner)
        h, w = int(inner.shape[0]), int(inner.shape[1])
        if (w + 1) // 2 <= winW or (h + 1) // 2 <= winH:
            break
    return levels


def LKOpticalFlowLevel(prevImg, prevDeriv, nextImg, prevPts, nextPts, status, err, winSize, maxCount, epsilon2, getMinEig, minEigThreshold):
    """cv_hal_LKOpticalFlowLevel: one pyramid level, in place on nextPts / status / err (status None above level 0).  The three images must be
    interior views of arrays padded by winSize."""
    I, D, J = Img(prevImg), Img(prevDeriv), Img(nextImg)
    n = int(prevPts.shape[0])
    bind_stream(I, J)
    ptr = lambda a: _vp(a.data_ptr() if torch is not None and isinstance(a, torch.Tensor) else a.ctypes.data) if a is not None else None
    rc = L.mi355cv_LKOpticalFlowLevel(_vp(I.ptr), I.step, _vp(D.ptr), D.step, _vp(J.ptr), J.step, I.w, I.h, I.cn, ptr(prevPts), ptr(nextPts), n,
                                      ptr(status), ptr(err), int(winSize[0]), int(winSize[1]), int(maxCount), float(epsilon2), bool(getMinEig),
                                      float(minEigThreshold))
    _lib.check(rc, "LKOpticalFlowLevel")


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts=None, winSize=(21, 21), maxLevel=3, criteria=(TERM_COUNT | TERM_EPS, 30, 0.01), flags=0,
                         minEigThreshold=1e-4):
    """cv::calcOpticalFlowPyrLK (lkpyramid.cpp:1432; SparsePyrLKOpticalFlowImpl::calc :1259) through the one-call entry point (pyramids,
    derivatives and every level on the device).  prevPts: n x 2 float32 (same kind as the images).  Returns (nextPts n x 2 float32,
    status n uint8, err n float32).  The same computation assembled from the video module's hooks: calcOpticalFlowPyrLK_hooks."""
    winW, winH = int(winSize[0]), int(winSize[1])
    if maxLevel < 0 or winW <= 2 or winH <= 2:
        raise ValueError("calcOpticalFlowPyrLK: maxLevel >= 0 and winSize > 2")                # CV_Assert, :1277
    n = int(prevPts.shape[0])
    dev = _is_dev(prevImg)
    pts = prevPts.reshape(n, 2)
    if dev:
        pts = pts.to(device=prevImg.device, dtype=torch.float32).contiguous()
    else:
        pts = np.ascontiguousarray(np.asarray(pts), np.float32)
    if flags & OPTFLOW_USE_INITIAL_FLOW:
        if nextPts is None or int(nextPts.shape[0]) != n:
            raise ValueError("calcOpticalFlowPyrLK: OPTFLOW_USE_INITIAL_FLOW needs nextPts of the same length")      # :1297
        out = nextPts.reshape(n, 2).to(device=prevImg.device, dtype=torch.float32).clone() if dev else np.array(np.asarray(nextPts).reshape(n, 2), np.float32)
    else:
        out = None
    if n == 0:
        return _empty(pts, (0, 2), np.float32), _empty(pts, (0,), np.uint8), _empty(pts, (0,), np.float32)
    status = _empty(pts, (n,), np.uint8)
    err = _empty(pts, (n,), np.float32)
    if out is None:
        out = _empty(pts, (n, 2), np.float32)
    I, J = Img(prevImg), Img(nextImg)
    if (I.w, I.h, I.cn, I.depth) != (J.w, J.h, J.cn, J.depth) or I.depth != CV_8U:
        raise ValueError("calcOpticalFlowPyrLK: two CV_8U images of the same size and type")   # CV_Assert, :1416-1417, :753
    bind_stream(I, J)
    ptr = lambda a: _vp(a.data_ptr() if torch is not None and isinstance(a, torch.Tensor) else a.ctypes.data)
    ctype, maxCount, eps = criteria
    rc = L.mi355cv_calcOpticalFlowPyrLK(_vp(I.ptr), I.step, _vp(J.ptr), J.step, I.w, I.h, I.cn, ptr(pts), ptr(out), n, ptr(status), ptr(err), winW, winH,
                                        int(maxLevel), int(ctype), int(maxCount), float(eps), int(flags), float(minEigThreshold))
    _lib.check(rc, "calcOpticalFlowPyrLK")
    return out, status, err


def calcOpticalFlowPyrLK_hooks(prevImg, nextImg, prevPts, nextPts=None, winSize=(21, 21), maxLevel=3, criteria=(TERM_COUNT | TERM_EPS, 30, 0.01), flags=0,
                         minEigThreshold=1e-4):
    """The level loop of SparsePyrLKOpticalFlowImpl::calc (lkpyramid.cpp:1397-1424) written out over the hooks the video module calls
    itself -- cv_hal_pyrdown, cv_hal_ScharrDeriv, cv_hal_LKOpticalFlowLevel -- with the per-level point scaling of LKTrackerInvoker
    (:215-231) done here.  Same arguments and results as calcOpticalFlowPyrLK."""
    winW, winH = int(winSize[0]), int(winSize[1])
    if maxLevel < 0 or winW <= 2 or winH <= 2:
        raise ValueError("calcOpticalFlowPyrLK_hooks: maxLevel >= 0 and winSize > 2")                # CV_Assert, :1277
    n = int(prevPts.shape[0])
    dev = _is_dev(prevImg)
    pts = prevPts.reshape(n, 2)
    if dev:
        pts = pts.to(device=prevImg.device, dtype=torch.float32).contiguous()
    else:
        pts = np.ascontiguousarray(np.asarray(pts), np.float32)
    if flags & OPTFLOW_USE_INITIAL_FLOW:
        if nextPts is None or int(nextPts.shape[0]) != n:
            raise ValueError("calcOpticalFlowPyrLK_hooks: OPTFLOW_USE_INITIAL_FLOW needs nextPts of the same length")      # :1297
        out = nextPts.reshape(n, 2).to(device=prevImg.device, dtype=torch.float32).clone() if dev else np.array(np.asarray(nextPts).reshape(n, 2), np.float32)
    else:
        out = None
    if n == 0:
        return _empty(pts, (0, 2), np.float32), _empty(pts, (0,), np.uint8), _empty(pts, (0,), np.float32)
    status = _empty(pts, (n,), np.uint8)
    status[...] = 1
    err = _zeros(pts, (n,), np.float32)
    ctype, maxCount, eps = criteria
    maxCount = 30 if (ctype & TERM_COUNT) == 0 else min(max(int(maxCount), 0), 100)            # :1386-1395
    eps = 0.01 if (ctype & TERM_EPS) == 0 else min(max(float(eps), 0.0), 10.0)
    eps *= eps
    prevPyr = buildOpticalFlowPyramid(prevImg, (winW, winH), maxLevel)
    nextPyr = buildOpticalFlowPyramid(nextImg, (winW, winH), maxLevel)
    maxLevel = min(len(prevPyr), len(nextPyr)) - 1
    for level in range(maxLevel, -1, -1):
        I, J = prevPyr[level], nextPyr[level]
        h, w = int(I.shape[0]), int(I.shape[1])
        cn = int(I.shape[2]) if I.ndim == 3 else 1
        dwhole = _zeros(I, (h + 2 * winH, w + 2 * winW, 2 * cn), np.int16)                     # BORDER_CONSTANT frame of the derivative image (:1409)
        dI = dwhole[winH:winH + h, winW:winW + w]
        ScharrDeriv(I, dst=dI)
        scale = 1.0 / (1 << level)
        prevScaled = pts * scale if dev else pts * np.float32(scale)
        if level == maxLevel:
            out = (out * scale if dev else out * np.float32(scale)) if flags & OPTFLOW_USE_INITIAL_FLOW else (prevScaled.clone() if dev else prevScaled.copy())
        else:
            out = out * 2.0 if dev else out * np.float32(2)
        out = out.contiguous() if dev else np.ascontiguousarray(out, np.float32)
        LKOpticalFlowLevel(I, dI, J, prevScaled.contiguous() if dev else np.ascontiguousarray(prevScaled), out, status if level == 0 else None, err,
                           (winW, winH), maxCount, eps, bool(flags & OPTFLOW_LK_GET_MIN_EIGENVALS), minEigThreshold)
    return out, status, err
