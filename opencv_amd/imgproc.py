"""Host-side mirror of the reference's imgproc dispatchers for the hot path.

Each function follows the argument handling of the cv:: function it is named
after (file:line cited) and then calls the matching cv_hal_* replacement in
libmi355cv.so.  No pixels are computed here.
"""
import ctypes
import numpy as np

from . import _lib
from .core import (Img, empty_like_kind, bind_stream, torch, CV_8U, CV_32F,  # noqa: F401
                   BORDER_CONSTANT, BORDER_ISOLATED, BORDER_DEFAULT)

L = _lib.lib
_vp = ctypes.c_void_p

__all__ = ["GaussianBlur", "GaussianBlurBatch", "sepSmoothFixedU8", "getGaussianKernelQ8_binomial",
           "getGaussianKernel", "getGaussianKernelQ"]

_BINOM = {1: [256], 3: [64, 128, 64], 5: [16, 64, 96, 64, 16], 7: [8, 28, 56, 72, 56, 28, 8],
          9: [4, 13, 30, 51, 60, 51, 30, 13, 4]}


def getGaussianKernelQ8_binomial(ksize):
    """Q8.8 taps of getGaussianKernelBitExact for sigma<=0 (smooth.dispatch.cpp:89-145)."""
    return np.array(_BINOM[ksize], dtype=np.uint16)


def getGaussianKernel(ksize, sigma, ktype=CV_32F + 1):
    """cv::getGaussianKernel (smooth.dispatch.cpp:200-222): CV_64F (default) or CV_32F column of taps."""
    buf = (ctypes.c_double * ksize)()
    _lib.check(L.mi355cv_getGaussianKernel(ksize, float(sigma), buf), "getGaussianKernel")
    k = np.array(buf[:], dtype=np.float64)
    return k.astype(np.float32) if ktype == CV_32F else k


def getGaussianKernelQ(ksize, sigma, fractionBits=8):
    """Fixed-point taps cv::GaussianBlur uses for CV_8U (Q8.8) / CV_16U (Q16.16), smooth.dispatch.cpp:224-258."""
    buf = (ctypes.c_int64 * ksize)()
    _lib.check(L.mi355cv_getGaussianKernelQ(ksize, float(sigma), fractionBits, buf), "getGaussianKernelQ")
    return np.array(buf[:], dtype=np.int64)


def _cvRound(v):
    return int(np.rint(v))


def _ksize(ksize):
    if isinstance(ksize, int):
        return ksize, ksize
    return int(ksize[0]), int(ksize[1])


def _copy_like(src):
    return src.clone() if torch is not None and isinstance(src, torch.Tensor) else np.array(src, copy=True)


def GaussianBlur(src, ksize, sigmaX=0.0, sigmaY=0.0, borderType=BORDER_DEFAULT, dst=None):
    """cv::GaussianBlur (smooth.dispatch.cpp:609-826), CV_8U sigma==0 square kernels for now.

    Mirrors :620-637 (1-pixel dimension clamps the kernel; 1x1 kernel is a copy), :639 (sigma2
    defaults to sigma1) and the hook selection :688-699 (cv_hal_gaussianBlurBinomial).
    """
    s = Img(src)
    kw, kh = _ksize(ksize)
    if (borderType & ~BORDER_ISOLATED) != BORDER_CONSTANT:      # :623-630 (src is never a submatrix here)
        if s.h == 1:
            kh = 1
        if s.w == 1:
            kw = 1
    if kw == 1 and kh == 1:                                     # :632-636
        out = _copy_like(src)
        if dst is not None:
            dst[...] = out
            return dst
        return out
    if sigmaY <= 0:
        sigmaY = sigmaX
    # createGaussianKernels :280-304: kernel size from sigma (3 sigma each side for 8U, 4 otherwise)
    if kw <= 0 and sigmaX > 0:
        kw = _cvRound(sigmaX * (3 if s.depth == CV_8U else 4) * 2 + 1) | 1
    if kh <= 0 and sigmaY > 0:
        kh = _cvRound(sigmaY * (3 if s.depth == CV_8U else 4) * 2 + 1) | 1
    if kw <= 0 or kh <= 0 or kw % 2 == 0 or kh % 2 == 0:
        raise ValueError("ksize must be positive and odd")       # :293-294
    sigmaX, sigmaY = max(sigmaX, 0.0), max(sigmaY, 0.0)
    if s.depth != CV_8U:
        raise NotImplementedError("GaussianBlur: only CV_8U so far")
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, s.cn, s.depth)
    d = Img(out)
    if (d.h, d.w, d.cn, d.depth) != (s.h, s.w, s.cn, s.depth):
        raise ValueError("dst geometry mismatch")
    if d.ptr == s.ptr:                                          # :685-686 in-place -> clone the source
        src = _copy_like(src)
        s = Img(src)
    bind_stream(s, d)
    if sigmaX == 0.0 and sigmaY == 0.0 and kw == kh:            # :688-699
        rc = L.mi355cv_gaussianBlurBinomial(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn,
                                            0, 0, 0, 0, kw, borderType & ~BORDER_ISOLATED)
        _lib.check(rc, "gaussianBlurBinomial")
        return out
    # everything else the reference runs through GaussianBlurFixedPoint (:720); cv_hal_gaussianBlur has the
    # matching signature (:708) and is bit-exact here, so it serves both ALGO_HINT modes.
    rc = L.mi355cv_gaussianBlur(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn, 0, 0, 0, 0,
                                kw, kh, float(sigmaX), float(sigmaY), borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "gaussianBlur")
    return out


def sepSmoothFixedU8(src, kx, ky, borderType=BORDER_DEFAULT, dst=None, margins=(0, 0, 0, 0)):
    """GaussianBlurFixedPoint<uint16_t> (smooth.simd.hpp:2219) with explicit Q8.8 taps.

    `margins` = (left, top, right, bottom) real pixels around `src` in memory (src is then a view
    into a larger image), as cv::GaussianBlur passes them for non-isolated borders.
    """
    s = Img(src)
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, s.cn, s.depth)
    d = Img(out)
    kx = np.ascontiguousarray(kx, dtype=np.uint16)
    ky = np.ascontiguousarray(ky, dtype=np.uint16)
    bind_stream(s, d)
    rc = L.mi355cv_sepSmoothFixedU8(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.cn,
                                    margins[0], margins[1], margins[2], margins[3],
                                    kx.ctypes.data, len(kx), ky.ctypes.data, len(ky), borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "sepSmoothFixedU8")
    return out


def GaussianBlurBatch(frames, ksize, borderType=BORDER_DEFAULT, dst=None):
    """N independent frames [N,H,W(,C)] resident in HBM, one launch (frames shard, never split)."""
    if torch is None or not isinstance(frames, torch.Tensor) or not frames.is_cuda:
        raise ValueError("GaussianBlurBatch needs a CUDA(ROCm) tensor [N,H,W(,C)]")
    if frames.dim() not in (3, 4) or frames.dtype != torch.uint8:
        raise ValueError("frames must be uint8 [N,H,W] or [N,H,W,C]")
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    cn = int(frames.shape[3]) if frames.dim() == 4 else 1
    if not frames[0].is_contiguous():
        raise ValueError("each frame must be contiguous")
    out = dst if dst is not None else torch.empty_like(frames)
    k = ksize if isinstance(ksize, int) else ksize[0]
    s0, d0 = Img(frames[0]), Img(out[0])
    bind_stream(s0, d0)
    rc = L.mi355cv_gaussianBlurBinomialBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)), _vp(d0.ptr), d0.step,
                                             int(out.stride(0)), n, w, h, CV_8U, cn, k, borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "gaussianBlurBinomialBatch")
    return out
