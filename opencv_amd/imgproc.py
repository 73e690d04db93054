"""Host-side mirror of the reference's imgproc dispatchers for the hot path.

Each function follows the argument handling of the cv:: function it is named
after (file:line cited) and then calls the matching cv_hal_* replacement in
libmi355cv.so.  No pixels are computed here.
"""
import ctypes
import numpy as np

from . import _lib
from .core import (Img, empty_like_kind, bind_stream, torch, CV_8U, CV_16U, CV_16S, CV_32F, CV_64F, _DEPTH_T,  # noqa: F401
                   BORDER_CONSTANT, BORDER_ISOLATED, BORDER_DEFAULT)

L = _lib.lib
_vp = ctypes.c_void_p

__all__ = ["cvtColor", "cvtColorBatch", "COLOR_BGR2YCrCb", "COLOR_RGB2YCrCb", "COLOR_YCrCb2BGR", "COLOR_YCrCb2RGB", "COLOR_BGR2YUV", "COLOR_RGB2YUV",
           "COLOR_YUV2BGR", "COLOR_YUV2RGB", "COLOR_YUV2RGB_NV12", "COLOR_YUV2BGR_NV12", "COLOR_YUV2RGB_NV21", "COLOR_YUV2BGR_NV21",
           "COLOR_YUV2RGBA_NV12", "COLOR_YUV2BGRA_NV12", "COLOR_YUV2RGBA_NV21", "COLOR_YUV2BGRA_NV21",
           "COLOR_BGR2Lab", "COLOR_RGB2Lab", "COLOR_LBGR2Lab", "COLOR_LRGB2Lab", "COLOR_Lab2BGR", "COLOR_Lab2RGB", "COLOR_Lab2LBGR", "COLOR_Lab2LRGB",
           "COLOR_BGR2Luv", "COLOR_RGB2Luv", "COLOR_LBGR2Luv", "COLOR_LRGB2Luv", "COLOR_Luv2BGR", "COLOR_Luv2RGB", "COLOR_Luv2LBGR", "COLOR_Luv2LRGB",
           "COLOR_BGR2HSV", "COLOR_RGB2HSV", "COLOR_BGR2HSV_FULL", "COLOR_RGB2HSV_FULL", "COLOR_YUV2RGB_YV12", "COLOR_YUV2BGR_YV12", "COLOR_YUV2RGB_IYUV", "COLOR_YUV2BGR_IYUV", "COLOR_YUV2RGB_I420", "COLOR_YUV2BGR_I420",
           "COLOR_YUV2RGBA_YV12", "COLOR_YUV2BGRA_YV12", "COLOR_YUV2RGBA_IYUV", "COLOR_YUV2BGRA_IYUV", "COLOR_BGR2BGRA", "COLOR_RGB2RGBA", "COLOR_BGRA2BGR", "COLOR_RGBA2RGB", "COLOR_BGR2RGBA",
           "COLOR_RGB2BGRA", "COLOR_RGBA2BGR", "COLOR_BGRA2RGB", "COLOR_BGR2RGB", "COLOR_RGB2BGR", "COLOR_BGRA2RGBA",
           "COLOR_RGBA2BGRA", "COLOR_BGR2GRAY", "COLOR_RGB2GRAY", "COLOR_GRAY2BGR", "COLOR_GRAY2RGB", "COLOR_GRAY2BGRA",
           "COLOR_GRAY2RGBA", "COLOR_BGRA2GRAY", "COLOR_RGBA2GRAY",
           "matchTemplate", "matchTemplateBatch", "integral", "integralBatch", "TM_SQDIFF", "TM_SQDIFF_NORMED", "TM_CCORR", "TM_CCORR_NORMED",
           "TM_CCOEFF", "TM_CCOEFF_NORMED",
           "pyrDown", "buildPyramid", "buildPyramidBatch", "cornerHarris", "cornerMinEigenVal", "cornerHarrisBatch", "goodFeaturesToTrack",
           "resize", "warpAffine", "warpPerspective", "SobelBatch", "boxFilterBatch", "sepFilter2DBatch", "thresholdBatch", "resizeBatch", "warpAffineBatch", "warpPerspectiveBatch", "pyrDownBatch", "remap", "convertMaps", "warpPolar", "WARP_FILL_OUTLIERS", "WARP_POLAR_LINEAR", "WARP_POLAR_LOG", "getRotationMatrix2D", "invertAffineTransform",
           "Canny", "equalizeHist", "cvtColorBGR2NV", "THRESH_OTSU", "adaptiveThreshold", "ADAPTIVE_THRESH_MEAN_C", "ADAPTIVE_THRESH_GAUSSIAN_C", "medianBlur", "bilateralFilter", "moments", "erode", "dilate", "MORPH_ERODE", "MORPH_DILATE", "threshold", "THRESH_BINARY", "THRESH_BINARY_INV", "THRESH_TRUNC", "THRESH_TOZERO", "THRESH_TOZERO_INV",
           "filter2D", "filter2DBatch", "cvtColorFilter2DBatch", "sepFilter2D", "Sobel", "Scharr", "boxFilter", "blur",
           "GaussianBlur", "GaussianBlurBatch", "sepSmoothFixedU8", "getGaussianKernelQ8_binomial",
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
        # smooth.dispatch.cpp:800-826: every depth without a fixed-point path ends in sepFilter2D with the taps of createGaussianKernels (:280-304:
        # getGaussianKernel(ksize, sigma, max(depth, CV_32F))); cv_hal_gaussianBlur declines these depths, the sepFilter hook serves them
        if s.depth == CV_16U:
            # the reference's Q16.16 fixed-point path (smooth.dispatch.cpp:726-760): its only hook is cv_hal_gaussianBlurBinomial (sigma 0, square kernel)
            if not (sigmaX == 0.0 and sigmaY == 0.0 and kw == kh and kw in (3, 5, 7, 9)):
                raise NotImplementedError("GaussianBlur: CV_16U beyond the sigma-0 square 3 / 5 / 7 / 9-tap case runs the reference's Q16.16 path, which has no hook: not served")
            out = dst if dst is not None else empty_like_kind(src, s.h, s.w, s.cn, s.depth)
            d = Img(out)
            if d.ptr == s.ptr:
                src = _copy_like(src); s = Img(src)
            bind_stream(s, d)
            rc = L.mi355cv_gaussianBlurBinomial(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn, 0, 0, 0, 0, kw, borderType & ~BORDER_ISOLATED)
            _lib.check(rc, "gaussianBlurBinomial")
            return out
        kx = getGaussianKernel(kw, sigmaX, CV_32F)
        ky = kx if (kh == kw and abs(sigmaY - sigmaX) < 2.220446049250313e-16) else getGaussianKernel(kh, sigmaY, CV_32F)
        return sepFilter2D(src, -1, kx, ky, borderType=borderType, dst=dst)
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


def GaussianBlurBatch(frames, ksize, borderType=BORDER_DEFAULT, dst=None, sigmaX=0.0, sigmaY=0.0):
    """N independent frames [N,H,W(,C)] resident in HBM, one launch (frames shard, never split).  Frames in HOST memory (a CPU tensor,
    ideally page-locked) take the library's pipelined path: chunks cross PCIe through two sets of device buffers, upload / filter / download overlapped."""
    if torch is None or not isinstance(frames, torch.Tensor):
        raise ValueError("GaussianBlurBatch needs a tensor [N,H,W(,C)] (CUDA(ROCm) resident, or a CPU tensor for the pipelined host path)")
    if frames.dim() not in (3, 4) or frames.dtype != torch.uint8:
        raise ValueError("frames must be uint8 [N,H,W] or [N,H,W,C]")
    if not frames.is_cuda:
        n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        cn = int(frames.shape[3]) if frames.dim() == 4 else 1
        if not frames[0].is_contiguous():
            raise ValueError("each frame must be contiguous")
        out = dst if dst is not None else torch.empty_like(frames, pin_memory=frames.is_pinned())
        k = ksize if isinstance(ksize, int) else ksize[0]
        bind_stream()                                         # host frames: the library's own streams
        rc = L.mi355cv_gaussianBlurBinomialBatch(_vp(frames.data_ptr()), w * cn, int(frames.stride(0)), _vp(out.data_ptr()), w * cn, int(out.stride(0)), n, w, h, CV_8U, cn, k,
                                                 borderType & ~BORDER_ISOLATED)
        _lib.check(rc, "gaussianBlurBinomialBatch")
        return out
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    cn = int(frames.shape[3]) if frames.dim() == 4 else 1
    if not frames[0].is_contiguous():
        raise ValueError("each frame must be contiguous")
    out = dst if dst is not None else torch.empty_like(frames)
    k = ksize if isinstance(ksize, int) else ksize[0]
    s0, d0 = Img(frames[0]), Img(out[0])
    bind_stream(s0, d0)
    if sigmaX > 0 or sigmaY > 0 or not isinstance(ksize, int):
        # any sigma: cv::GaussianBlur's kernel-size rule for CV_8U (createGaussianKernels, smooth.dispatch.cpp:270-276: cvRound(sigma * 3 * 2 + 1) | 1) and its Q8.8 taps
        kw, kh = (ksize, ksize) if isinstance(ksize, int) else ksize
        sy = sigmaY if sigmaY > 0 else sigmaX
        if kw <= 0 and sigmaX > 0: kw = int(np.rint(sigmaX * 6 + 1)) | 1
        if kh <= 0 and sy > 0: kh = int(np.rint(sy * 6 + 1)) | 1
        rc = L.mi355cv_gaussianBlurBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)), _vp(d0.ptr), d0.step, int(out.stride(0)), n, w, h, CV_8U, cn, kw, kh,
                                         float(sigmaX), float(sigmaY), borderType & ~BORDER_ISOLATED)
        _lib.check(rc, "gaussianBlurBatch")
        return out
    rc = L.mi355cv_gaussianBlurBinomialBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)), _vp(d0.ptr), d0.step,
                                             int(out.stride(0)), n, w, h, CV_8U, cn, k, borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "gaussianBlurBinomialBatch")
    return out


# ----------------------------------------------------------------------------- cvtColor (a6)
# ColorConversionCodes (imgproc.hpp:537-563)
COLOR_BGR2BGRA = COLOR_RGB2RGBA = 0
COLOR_BGRA2BGR = COLOR_RGBA2RGB = 1
COLOR_BGR2RGBA = COLOR_RGB2BGRA = 2
COLOR_RGBA2BGR = COLOR_BGRA2RGB = 3
COLOR_BGR2RGB = COLOR_RGB2BGR = 4
COLOR_BGRA2RGBA = COLOR_RGBA2BGRA = 5
COLOR_BGR2GRAY, COLOR_RGB2GRAY = 6, 7
COLOR_GRAY2BGR = COLOR_GRAY2RGB = 8
COLOR_GRAY2BGRA = COLOR_GRAY2RGBA = 9
COLOR_BGRA2GRAY, COLOR_RGBA2GRAY = 10, 11

# code -> (scn, dcn, swapBlue), as the switch in cv::cvtColor does (color.cpp:192-260 / color.hpp dcn helpers)
_RGB2RGB = {COLOR_BGR2BGRA: (3, 4, False), COLOR_BGRA2BGR: (4, 3, False), COLOR_BGR2RGBA: (3, 4, True),
            COLOR_RGBA2BGR: (4, 3, True), COLOR_BGR2RGB: (3, 3, True), COLOR_BGRA2RGBA: (4, 4, True)}
_RGB2GRAY = {COLOR_BGR2GRAY: (3, False), COLOR_RGB2GRAY: (3, True), COLOR_BGRA2GRAY: (4, False), COLOR_RGBA2GRAY: (4, True)}
_GRAY2RGB = {COLOR_GRAY2BGR: 3, COLOR_GRAY2BGRA: 4}
# cv::ColorConversionCodes of the YUV family (imgproc.hpp:560-700): code -> (swapBlue, isCbCr) / (dcn, swapBlue, uIdx)
COLOR_BGR2YCrCb, COLOR_RGB2YCrCb, COLOR_YCrCb2BGR, COLOR_YCrCb2RGB = 36, 37, 38, 39
COLOR_BGR2YUV, COLOR_RGB2YUV, COLOR_YUV2BGR, COLOR_YUV2RGB = 82, 83, 84, 85
COLOR_YUV2RGB_NV12, COLOR_YUV2BGR_NV12, COLOR_YUV2RGB_NV21, COLOR_YUV2BGR_NV21 = 90, 91, 92, 93
COLOR_YUV2RGBA_NV12, COLOR_YUV2BGRA_NV12, COLOR_YUV2RGBA_NV21, COLOR_YUV2BGRA_NV21 = 94, 95, 96, 97
COLOR_BGR2HSV, COLOR_RGB2HSV, COLOR_BGR2HSV_FULL, COLOR_RGB2HSV_FULL = 40, 41, 66, 67
COLOR_BGR2Lab, COLOR_RGB2Lab, COLOR_LBGR2Lab, COLOR_LRGB2Lab = 44, 45, 74, 75                  # imgproc.hpp:602-631
COLOR_Lab2BGR, COLOR_Lab2RGB, COLOR_Lab2LBGR, COLOR_Lab2LRGB = 56, 57, 78, 79
COLOR_BGR2Luv, COLOR_RGB2Luv, COLOR_LBGR2Luv, COLOR_LRGB2Luv = 50, 51, 76, 77
COLOR_Luv2BGR, COLOR_Luv2RGB, COLOR_Luv2LBGR, COLOR_Luv2LRGB = 58, 59, 80, 81
_HSV = {40: (0, 0), 41: (1, 0), 66: (0, 1), 67: (1, 1)}
_YUV_FWD = {82: (0, 0), 83: (1, 0), 36: (0, 1), 37: (1, 1)}
_YUV_INV = {84: (0, 0), 85: (1, 0), 38: (0, 1), 39: (1, 1)}
COLOR_YUV2RGB_YV12, COLOR_YUV2BGR_YV12, COLOR_YUV2RGB_IYUV, COLOR_YUV2BGR_IYUV = 98, 99, 100, 101
COLOR_YUV2RGB_I420, COLOR_YUV2BGR_I420 = 100, 101
COLOR_YUV2RGBA_YV12, COLOR_YUV2BGRA_YV12, COLOR_YUV2RGBA_IYUV, COLOR_YUV2BGRA_IYUV = 102, 103, 104, 105
_YUV_3P = {98: (3, 1, 1), 99: (3, 0, 1), 100: (3, 1, 0), 101: (3, 0, 0), 102: (4, 1, 1), 103: (4, 0, 1), 104: (4, 1, 0), 105: (4, 0, 0)}
_YUV_NV = {90: (3, 1, 0), 91: (3, 0, 0), 92: (3, 1, 1), 93: (3, 0, 1), 94: (4, 1, 0), 95: (4, 0, 0), 96: (4, 1, 1), 97: (4, 0, 1)}


def cvtColor(src, code, dst=None, dstCn=0):
    """cv::cvtColor (color.cpp:192-) for the RGB<->RGB / RGB<->gray families; depth 8U, 16U, 32F."""
    s = Img(src)
    if code in _RGB2GRAY:
        scn, swap = _RGB2GRAY[code]
        if s.cn != scn:
            raise ValueError(f"cvtColor: source must have {scn} channels")     # CvtHelper asserts, color.hpp:140
        out = dst if dst is not None else empty_like_kind(src[..., 0], s.h, s.w, 1, s.depth)
        d = Img(out)
        bind_stream(s, d)
        _lib.check(L.mi355cv_cvtBGRtoGray(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, scn, swap), "cvtBGRtoGray")
        return out
    if code in _GRAY2RGB:
        dcn = _GRAY2RGB[code]
        if s.cn != 1:
            raise ValueError("cvtColor: source must be single-channel")
        ref3 = src[..., None] if getattr(src, "ndim", 2) == 2 else src
        out = dst if dst is not None else empty_like_kind(ref3, s.h, s.w, dcn, s.depth)
        d = Img(out)
        bind_stream(s, d)
        _lib.check(L.mi355cv_cvtGraytoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, dcn), "cvtGraytoBGR")
        return out
    if code in _RGB2RGB:
        scn, dcn, swap = _RGB2RGB[code]
        if s.cn != scn:
            raise ValueError(f"cvtColor: source must have {scn} channels")
        out = dst if dst is not None else empty_like_kind(src, s.h, s.w, dcn, s.depth)
        d = Img(out)
        bind_stream(s, d)
        _lib.check(L.mi355cv_cvtBGRtoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, scn, dcn, swap), "cvtBGRtoBGR")
        return out
    if code in _HSV or code in _HLS:
        swap, full = _HSV[code] if code in _HSV else _HLS[code]
        if s.cn not in (3, 4):
            raise ValueError("cvtColor: source must have 3 or 4 channels")
        ref3 = src[..., :3] if s.cn == 4 else src
        out = dst if dst is not None else empty_like_kind(ref3, s.h, s.w, 3, s.depth)
        d = Img(out)
        bind_stream(s, d)
        _lib.check(L.mi355cv_cvtBGRtoHSV(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn, bool(swap), bool(full), code in _HSV), "cvtBGRtoHSV")
        return out
    if code in _YUV_FWD:
        swap, cbcr = _YUV_FWD[code]
        if s.cn not in (3, 4):
            raise ValueError("cvtColor: source must have 3 or 4 channels")
        ref3 = src[..., :3] if s.cn == 4 else src
        out = dst if dst is not None else empty_like_kind(ref3, s.h, s.w, 3, s.depth)
        d = Img(out)
        bind_stream(s, d)
        _lib.check(L.mi355cv_cvtBGRtoYUV(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn, bool(swap), bool(cbcr)), "cvtBGRtoYUV")
        return out
    if code in _YUV_INV:
        swap, cbcr = _YUV_INV[code]
        dcn = dstCn if dstCn in (3, 4) else 3
        if s.cn != 3:
            raise ValueError("cvtColor: source must have 3 channels")
        out = dst if dst is not None else empty_like_kind(src, s.h, s.w, dcn, s.depth)
        d = Img(out)
        bind_stream(s, d)
        _lib.check(L.mi355cv_cvtYUVtoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, dcn, bool(swap), bool(cbcr)), "cvtYUVtoBGR")
        return out
    if code in _YUV_NV or code in _YUV_3P:
        dcn, swap, uidx = _YUV_NV[code] if code in _YUV_NV else _YUV_3P[code]
        if s.cn != 1 or s.depth != CV_8U or s.h % 3 or s.w % 2:
            raise ValueError("cvtColor: NV12/NV21 needs a CV_8UC1 image of (3/2 * height) x width, width and height even")    # color.cpp cvtColorTwoPlane
        dh = s.h * 2 // 3
        out = dst if dst is not None else empty_like_kind(src[..., None] if getattr(src, "ndim", 2) == 2 else src, dh, s.w, dcn, s.depth)
        d = Img(out)
        bind_stream(s, d)
        if code in _YUV_NV:
            _lib.check(L.mi355cv_cvtTwoPlaneYUVtoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, dh, dcn, bool(swap), uidx), "cvtTwoPlaneYUVtoBGR")
        else:
            _lib.check(L.mi355cv_cvtThreePlaneYUVtoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, dh, dcn, bool(swap), uidx), "cvtThreePlaneYUVtoBGR")
        return out
    if code in _HSV_INV or code in _HLS_INV:                                # COLOR_HSV2BGR / RGB (54, 55), COLOR_HLS2BGR / RGB (60, 61) and their *_FULL forms
        swap, full = _HSV_INV[code] if code in _HSV_INV else _HLS_INV[code]
        dcn = dstCn if dstCn in (3, 4) else 3
        if s.cn != 3 or s.depth not in (CV_8U, CV_32F):
            raise ValueError("cvtColor: HSV / HLS -> BGR needs a CV_8UC3 or CV_32FC3 source")
        out = dst if dst is not None else empty_like_kind(src, s.h, s.w, dcn, s.depth)
        d = Img(out)
        bind_stream(s, d)
        _lib.check(L.mi355cv_cvtHSVtoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, dcn, bool(swap), bool(full), code in _HSV_INV), "cvtHSVtoBGR")
        return out
    if code in _MISC:
        return _cvt_misc(src, s, code, dst, dstCn)
    raise NotImplementedError(f"cvtColor: conversion code {code} is outside the hot path built so far")


_HSV_INV = {54: (0, 0), 55: (1, 0), 70: (0, 1), 71: (1, 1)}
_HLS = {52: (0, 0), 53: (1, 0), 68: (0, 1), 69: (1, 1)}                          # COLOR_BGR2HLS, RGB2HLS, BGR2HLS_FULL, RGB2HLS_FULL (imgproc.hpp:606-640)
_HLS_INV = {60: (0, 0), 61: (1, 0), 72: (0, 1), 73: (1, 1)}                      # COLOR_HLS2BGR, HLS2RGB, HLS2BGR_FULL, HLS2RGB_FULL
COLOR_BGR2HLS, COLOR_RGB2HLS, COLOR_HLS2BGR, COLOR_HLS2RGB = 52, 53, 60, 61
COLOR_BGR2HLS_FULL, COLOR_RGB2HLS_FULL, COLOR_HLS2BGR_FULL, COLOR_HLS2RGB_FULL = 68, 69, 72, 73


def _misc_table():
    """cvtColor codes (imgproc.hpp:565-864) of the remaining integer conversions -> (kind, parameters as color.cpp:200-372 derives them)"""
    t = {}
    for code, swap in ((32, 0), (33, 1)): t[code] = ("to_xyz", swap)
    for code, swap in ((34, 0), (35, 1)): t[code] = ("from_xyz", swap)
    for code, swap, srgb in ((44, 0, 1), (45, 1, 1), (74, 0, 0), (75, 1, 0)): t[code] = ("to_lab", swap, srgb)           # BGR2Lab, RGB2Lab, LBGR2Lab, LRGB2Lab
    for code, swap, srgb in ((56, 0, 1), (57, 1, 1), (78, 0, 0), (79, 1, 0)): t[code] = ("from_lab", swap, srgb)         # Lab2BGR, Lab2RGB, Lab2LBGR, Lab2LRGB
    for code, swap, srgb in ((50, 0, 1), (51, 1, 1), (76, 0, 0), (77, 1, 0)): t[code] = ("to_luv", swap, srgb)           # BGR2Luv, RGB2Luv, LBGR2Luv, LRGB2Luv
    for code, swap, srgb in ((58, 0, 1), (59, 1, 1), (80, 0, 0), (81, 1, 0)): t[code] = ("from_luv", swap, srgb)         # Luv2BGR, Luv2RGB, Luv2LBGR, Luv2LRGB
    for code, cn, swap, gb in ((12, 3, 0, 6), (13, 3, 1, 6), (16, 4, 0, 6), (17, 4, 1, 6), (22, 3, 0, 5), (23, 3, 1, 5), (26, 4, 0, 5), (27, 4, 1, 5)):
        t[code] = ("to_5x5", cn, swap, gb)
    for code, cn, swap, gb in ((14, 3, 0, 6), (15, 3, 1, 6), (18, 4, 0, 6), (19, 4, 1, 6), (24, 3, 0, 5), (25, 3, 1, 5), (28, 4, 0, 5), (29, 4, 1, 5)):
        t[code] = ("from_5x5", cn, swap, gb)
    t[20] = ("gray_to_5x5", 6); t[30] = ("gray_to_5x5", 5); t[21] = ("5x5_to_gray", 6); t[31] = ("5x5_to_gray", 5)
    for code, dcn, swap, uidx, ycn in ((107, 3, 1, 0, 1), (108, 3, 0, 0, 1), (111, 4, 1, 0, 1), (112, 4, 0, 0, 1), (115, 3, 1, 0, 0), (116, 3, 0, 0, 0),
                                       (117, 3, 1, 1, 0), (118, 3, 0, 1, 0), (119, 4, 1, 0, 0), (120, 4, 0, 0, 0), (121, 4, 1, 1, 0), (122, 4, 0, 1, 0)):
        t[code] = ("dec422", dcn, swap, uidx, ycn)
    for code, scn, swap, uidx, ycn in ((143, 3, 1, 0, 1), (144, 3, 0, 0, 1), (145, 4, 1, 0, 1), (146, 4, 0, 0, 1), (147, 3, 1, 0, 0), (148, 3, 0, 0, 0),
                                       (149, 3, 1, 1, 0), (150, 3, 0, 1, 0), (151, 4, 1, 0, 0), (152, 4, 0, 0, 0), (153, 4, 1, 1, 0), (154, 4, 0, 1, 0)):
        t[code] = ("enc422", scn, swap, uidx, ycn)
    for code, scn, swap, uidx in ((127, 3, 1, 1), (128, 3, 0, 1), (129, 4, 1, 1), (130, 4, 0, 1), (131, 3, 1, 2), (132, 3, 0, 2), (133, 4, 1, 2), (134, 4, 0, 2)):
        t[code] = ("enc420p", scn, swap, uidx)
    t[125] = ("premul",); t[126] = ("unpremul",)
    return t


_MISC = _misc_table()


def _like(src, h, w, cn, depth):
    """a fresh array of the same kind (numpy / torch, device) as `src`"""
    base = src
    while getattr(base, "ndim", 2) > 2:
        base = base[..., 0]
    return empty_like_kind(base if cn == 1 else base[..., None], h, w, cn, depth)


def _cvt_misc(src, s, code, dst, dstCn):
    k = _MISC[code]
    kind = k[0]

    def need(cond, what):
        if not cond:
            raise ValueError("cvtColor: " + what)                                               # the CvtHelper assertions, color.hpp:140-170

    a = None
    if kind == "to_xyz":
        need(s.cn in (3, 4) and s.depth in (CV_8U, CV_16U, CV_32F), "BGR2XYZ: 3 or 4 channels, CV_8U / CV_16U / CV_32F")
        out = dst if dst is not None else _like(src, s.h, s.w, 3, s.depth)
        call = lambda d: L.mi355cv_cvtBGRtoXYZ(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn, bool(k[1]))
    elif kind == "from_xyz":
        need(s.cn == 3 and s.depth in (CV_8U, CV_16U, CV_32F), "XYZ2BGR: 3 channels, CV_8U / CV_16U / CV_32F")
        dcn = dstCn if dstCn in (3, 4) else 3
        out = dst if dst is not None else _like(src, s.h, s.w, dcn, s.depth)
        call = lambda d: L.mi355cv_cvtXYZtoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, dcn, bool(k[1]))
    elif kind in ("to_lab", "to_luv"):
        need(s.cn in (3, 4) and s.depth in (CV_8U, CV_32F), "BGR2Lab / BGR2Luv: 3 or 4 channels, CV_8U or CV_32F")
        out = dst if dst is not None else _like(src, s.h, s.w, 3, s.depth)
        call = lambda d: L.mi355cv_cvtBGRtoLab(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn, bool(k[1]), kind == "to_lab", bool(k[2]))
    elif kind in ("from_lab", "from_luv"):
        need(s.cn == 3 and s.depth in (CV_8U, CV_32F), "Lab2BGR / Luv2BGR: 3 channels, CV_8U or CV_32F")
        dcn = dstCn if dstCn in (3, 4) else 3
        out = dst if dst is not None else _like(src, s.h, s.w, dcn, s.depth)
        call = lambda d: L.mi355cv_cvtLabtoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, dcn, bool(k[1]), kind == "from_lab", bool(k[2]))
    elif kind == "to_5x5":
        need(s.cn == k[1] and s.depth == CV_8U, f"source must be CV_8UC{k[1]}")
        out = dst if dst is not None else _like(src, s.h, s.w, 2, CV_8U)
        call = lambda d: L.mi355cv_cvtBGRtoBGR5x5(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.cn, bool(k[2]), k[3])
    elif kind == "from_5x5":
        need(s.cn == 2 and s.depth == CV_8U, "source must be CV_8UC2")
        out = dst if dst is not None else _like(src, s.h, s.w, k[1], CV_8U)
        call = lambda d: L.mi355cv_cvtBGR5x5toBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, k[1], bool(k[2]), k[3])
    elif kind == "gray_to_5x5":
        need(s.cn == 1 and s.depth == CV_8U, "source must be CV_8UC1")
        out = dst if dst is not None else _like(src, s.h, s.w, 2, CV_8U)
        call = lambda d: L.mi355cv_cvtGraytoBGR5x5(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, k[1])
    elif kind == "5x5_to_gray":
        need(s.cn == 2 and s.depth == CV_8U, "source must be CV_8UC2")
        out = dst if dst is not None else _like(src, s.h, s.w, 1, CV_8U)
        call = lambda d: L.mi355cv_cvtBGR5x5toGray(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, k[1])
    elif kind == "dec422":
        need(s.cn == 2 and s.depth == CV_8U and s.w % 2 == 0, "YUV 4:2:2 source must be CV_8UC2 with an even width")
        out = dst if dst is not None else _like(src, s.h, s.w, k[1], CV_8U)
        call = lambda d: L.mi355cv_cvtOnePlaneYUVtoBGR(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, k[1], bool(k[2]), k[3], k[4])
    elif kind == "enc422":
        need(s.cn == k[1] and s.depth == CV_8U and s.w % 2 == 0, f"source must be CV_8UC{k[1]} with an even width")
        out = dst if dst is not None else _like(src, s.h, s.w, 2, CV_8U)
        call = lambda d: L.mi355cv_cvtOnePlaneBGRtoYUV(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.cn, bool(k[2]), k[3], k[4])
    elif kind == "enc420p":
        need(s.cn == k[1] and s.depth == CV_8U and s.w % 2 == 0 and s.h % 2 == 0, f"source must be CV_8UC{k[1]} with even width and height")
        out = dst if dst is not None else _like(src, s.h * 3 // 2, s.w, 1, CV_8U)
        call = lambda d: L.mi355cv_cvtBGRtoThreePlaneYUV(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.cn, bool(k[2]), k[3])
    else:
        need(s.cn == 4 and s.depth == CV_8U, "source must be CV_8UC4")
        out = dst if dst is not None else _like(src, s.h, s.w, 4, CV_8U)
        fn = L.mi355cv_cvtRGBAtoMultipliedRGBA if kind == "premul" else L.mi355cv_cvtMultipliedRGBAtoRGBA
        call = lambda d: fn(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(call(d), "cvtColor(" + kind + ")")
    return out


def cvtColorBGR2NV(src, swapBlue=False, nv21=False, dst=None):
    """cv::hal::cvtBGRtoTwoPlaneYUV (color_yuv.dispatch.cpp:231; no cvtColor code reaches it -- videoio's writers call it): BGR/RGB(A) ->
    NV12 (or NV21) as one (3/2 * height) x width CV_8UC1 array, luma rows followed by the interleaved chroma rows."""
    s = Img(src)
    if s.cn not in (3, 4) or s.depth != CV_8U or s.w % 2 or s.h % 2:
        raise ValueError("cvtColorBGR2NV: CV_8UC3 / CV_8UC4 with even width and height")
    out = dst if dst is not None else _like(src, s.h * 3 // 2, s.w, 1, CV_8U)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_cvtBGRtoTwoPlaneYUV(_vp(s.ptr), s.step, _vp(d.ptr), d.step, _vp(d.ptr + d.step * s.h), d.step, s.w, s.h, s.cn, bool(swapBlue),
                                             2 if nv21 else 1), "cvtBGRtoTwoPlaneYUV")
    return out


def cvtColorBatch(frames, code, dst=None):
    """[N,H,W,C] device-resident frames -> gray [N,H,W], one launch."""
    if code not in _RGB2GRAY:
        raise NotImplementedError("cvtColorBatch: only *2GRAY")
    scn, swap = _RGB2GRAY[code]
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    out = dst if dst is not None else _batch_alloc(frames, (n, h, w), frames.dtype)
    s0, d0 = Img(frames[0]), Img(out[0])
    bind_stream(s0, d0)
    rc = L.mi355cv_cvtBGRtoGrayBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, _vp(d0.ptr), d0.step,
                                     int(out.stride(0)) * d0.esz, n, w, h, s0.depth, scn, int(swap))
    _lib.check(rc, "cvtBGRtoGrayBatch")
    return out


# ----------------------------------------------------------------------------- threshold (f1)
THRESH_BINARY, THRESH_BINARY_INV, THRESH_TRUNC, THRESH_TOZERO, THRESH_TOZERO_INV = range(5)
THRESH_OTSU, THRESH_TRIANGLE = 8, 16
_SAT = {CV_8U: (0, 255), CV_16U: (0, 65535), CV_16S: (-32768, 32767)}


def threshold(src, thresh, maxval, type, dst=None):
    """cv::threshold (thresh.cpp:1542) for the fixed-level types: the reference's own preprocessing of (thresh, maxval) and its
    degenerate-threshold shortcuts on the host side, the per-element rule through cv_hal_threshold.  Returns (retval, dst)."""
    s = Img(src)
    if type & THRESH_OTSU and not type & THRESH_TRIANGLE:                      # thresh.cpp:1563-1569 -> cv_hal_threshold_otsu
        if s.cn != 1 or s.depth not in (CV_8U, CV_16U):
            raise ValueError("threshold: THRESH_OTSU needs CV_8UC1 or CV_16UC1")
        out = dst if dst is not None else empty_like_kind(src, s.h, s.w, 1, s.depth)
        d = Img(out)
        bind_stream(s, d)
        level = ctypes.c_double(0)
        _lib.check(L.mi355cv_threshold_otsu(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, float(maxval), int(type & 7), ctypes.byref(level)),
                   "threshold_otsu")
        return level.value, out
    if type < 0 or type > 4:
        raise NotImplementedError("threshold: THRESH_TRIANGLE estimates the level on the CPU in the reference; not on this path")
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, s.cn, s.depth)
    if s.depth in _SAT:
        lo, hi = _SAT[s.depth]
        ithresh = int(np.floor(thresh))
        imaxval = int(np.rint(maxval))
        if type == THRESH_TRUNC:
            imaxval = ithresh
        imaxval = min(max(imaxval, lo), hi)
        if ithresh < lo or ithresh >= hi:                                   # thresh.cpp:1595-1609
            if type in (THRESH_BINARY, THRESH_BINARY_INV) or (type in (THRESH_TRUNC, THRESH_TOZERO_INV) and ithresh < lo) or \
                    (type == THRESH_TOZERO and ithresh >= hi):
                v = (0 if ithresh >= hi else imaxval) if type == THRESH_BINARY else \
                    (imaxval if ithresh >= hi else 0) if type == THRESH_BINARY_INV else 0
                out[...] = v
            elif out is not src:
                out[...] = src
            return float(ithresh), out
        thresh, maxval = float(ithresh), float(imaxval)
    elif s.depth not in (CV_32F, CV_64F):
        raise NotImplementedError("threshold: depth")
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_threshold(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn, float(thresh), float(maxval), int(type)), "threshold")
    return float(thresh), out


def Canny(image, threshold1, threshold2, apertureSize=3, L2gradient=False, dst=None):
    """cv::Canny (canny.cpp:823) through cv_hal_canny: CV_8U, 1-4 channels, aperture 3 or 5."""
    s = Img(image)
    if s.depth != CV_8U:
        raise ValueError("Canny: CV_8U only")                                               # CV_Assert, :829
    if apertureSize % 2 == 0 or apertureSize < 3 or apertureSize > 7:
        raise ValueError("Canny: aperture size should be odd between 3 and 7")              # :845
    lo, hi = float(threshold1), float(threshold2)
    if apertureSize == 7:
        lo, hi = lo / 16.0, hi / 16.0
    if lo > hi:
        lo, hi = hi, lo
    ref2 = image[..., 0] if s.cn > 1 else image
    out = dst if dst is not None else empty_like_kind(ref2, s.h, s.w, 1, CV_8U)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_canny(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.cn, lo, hi, int(apertureSize), bool(L2gradient)), "canny")
    return out


def equalizeHist(src, dst=None):
    """cv::equalizeHist (histogram.cpp:3436) through cv_hal_equalize_hist: CV_8UC1."""
    s = Img(src)
    if s.depth != CV_8U or s.cn != 1:
        raise ValueError("equalizeHist: CV_8UC1 only")                                        # CV_Assert, :3440
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, 1, CV_8U)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_equalize_hist(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h), "equalize_hist")
    return out


ADAPTIVE_THRESH_MEAN_C, ADAPTIVE_THRESH_GAUSSIAN_C = 0, 1


def adaptiveThreshold(src, maxValue, adaptiveMethod, thresholdType, blockSize, C, dst=None):
    """cv::adaptiveThreshold (thresh.cpp:1693) through cv_hal_adaptiveThreshold: CV_8UC1, ADAPTIVE_THRESH_MEAN_C (any odd block the box hook
    serves) and ADAPTIVE_THRESH_GAUSSIAN_C (blockSize <= 33, the separable hook's tap limit)."""
    s = Img(src)
    if s.depth != CV_8U or s.cn != 1:
        raise ValueError("adaptiveThreshold: CV_8UC1 only")                                   # CV_Assert, :1699
    if blockSize % 2 != 1 or blockSize <= 1:
        raise ValueError("adaptiveThreshold: blockSize must be odd and > 1")
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, 1, s.depth)
    if maxValue < 0:
        out[...] = 0
        return out
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_adaptiveThreshold(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, float(maxValue), int(adaptiveMethod), int(thresholdType),
                                           int(blockSize), float(C)), "adaptiveThreshold")
    return out


def moments(src, binaryImage=False):
    """cv::moments of a single-channel CV_8U / CV_16U / CV_16S / CV_32F / CV_64F image through cv_hal_imageMoments: the ten spatial moments as a dict (m00 .. m03); the central
    and normalised moments follow from them as in completeMomentState (moments.cpp:33-66)."""
    s = Img(src)
    if s.cn != 1:
        raise ValueError("moments: single-channel images")
    bind_stream(s)
    buf = (ctypes.c_double * 10)()
    _lib.check(L.mi355cv_imageMoments(_vp(s.ptr), s.step, s.type, s.w, s.h, bool(binaryImage), buf), "imageMoments")
    return dict(zip(("m00", "m10", "m01", "m20", "m11", "m02", "m30", "m21", "m12", "m03"), buf[:]))


def bilateralFilter(src, d, sigmaColor, sigmaSpace, borderType=BORDER_DEFAULT, dst=None):
    """cv::bilateralFilter (bilateral_filter.dispatch.cpp:393) through cv_hal_bilateralFilter: CV_8UC1 / CV_8UC3 / CV_32FC1 / CV_32FC3, radius <= 16.  A view with padded rows is
    treated as the image it shows (BORDER_ISOLATED): the mirror has no parent to pad from."""
    s = Img(src)
    if s.depth not in (CV_8U, CV_32F) or s.cn not in (1, 3):
        raise NotImplementedError("bilateralFilter: CV_8UC1 / CV_8UC3 / CV_32FC1 / CV_32FC3")
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, s.cn, s.depth)
    dd = Img(out)
    bind_stream(s, dd)
    _lib.check(L.mi355cv_bilateralFilter(_vp(s.ptr), s.step, _vp(dd.ptr), dd.step, s.w, s.h, s.depth, s.cn, int(d), float(sigmaColor), float(sigmaSpace),
                                         int(borderType) | BORDER_ISOLATED), "bilateralFilter")
    return out


# ----------------------------------------------------------------------------- median (f1)
def medianBlur(src, ksize, dst=None):
    """cv::medianBlur (median_blur.dispatch.cpp:279) through cv_hal_medianBlur: CV_8U, ksize 3 / 5."""
    s = Img(src)
    if ksize % 2 != 1:
        raise ValueError("medianBlur: ksize must be odd")
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, s.cn, s.depth)
    if ksize <= 1:
        out[...] = src
        return out
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_medianBlur(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, s.cn, int(ksize)), "medianBlur")
    return out


# ----------------------------------------------------------------------------- erode / dilate (f1)
MORPH_ERODE, MORPH_DILATE = 0, 1
_DBL_MAX = 1.7976931348623157e308


def _morph(op, src, kernel, anchor, iterations, borderType, borderValue, dst, roi):
    """morphOp (morph.dispatch.cpp:935-1010): default 3x3 rectangle, iterations of a rectangle folded into one bigger rectangle,
    then cv_hal_morphInit / cv_hal_morph / cv_hal_morphFree."""
    view, s, fw, fh, ox, oy = _parent_geometry(src, roi, borderType)
    k = None if kernel is None else np.ascontiguousarray(np.asarray(kernel) != 0, dtype=np.uint8)
    ksz = (3, 3) if k is None else (k.shape[1], k.shape[0])
    ax = ksz[0] // 2 if anchor[0] < 0 else anchor[0]
    ay = ksz[1] // 2 if anchor[1] < 0 else anchor[1]
    out = dst if dst is not None else empty_like_kind(view, s.h, s.w, s.cn, s.depth)
    if iterations == 0 or ksz == (1, 1):
        out[...] = view
        return out
    if k is None:
        k = np.ones((1 + 2 * iterations, 1 + 2 * iterations), np.uint8)
        ax = ay = iterations
        iterations = 1
    elif iterations > 1 and int(k.sum()) == k.size:
        ax, ay = ax * iterations, ay * iterations
        k = np.ones((ksz[1] + (iterations - 1) * (ksz[1] - 1), ksz[0] + (iterations - 1) * (ksz[0] - 1)), np.uint8)
        iterations = 1
    bv = (ctypes.c_double * 4)(*([_DBL_MAX] * 4)) if borderValue is None else (ctypes.c_double * 4)(*_border_value(borderValue))
    d = Img(out)
    bind_stream(s, d)
    ctx = ctypes.c_void_p()
    rc = L.mi355cv_morphInit(ctypes.byref(ctx), op, s.type, d.type, s.w, s.h, 0, k.ctypes.data, k.strides[0], k.shape[1], k.shape[0], ax, ay,
                             borderType & ~BORDER_ISOLATED, bv, iterations, roi is not None, d.ptr == s.ptr)
    _lib.check(rc, "morphInit")
    try:
        rc = L.mi355cv_morph(ctx, _vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, fw, fh, ox, oy, s.w, s.h, 0, 0)
    finally:
        L.mi355cv_morphFree(ctx)
    _lib.check(rc, "morph")
    return out


def erode(src, kernel=None, anchor=(-1, -1), iterations=1, borderType=BORDER_CONSTANT, borderValue=None, dst=None, roi=None):
    """cv::erode (morph.dispatch.cpp:1013).  borderValue=None is morphologyDefaultBorderValue()."""
    return _morph(MORPH_ERODE, src, kernel, anchor, iterations, borderType, borderValue, dst, roi)


def dilate(src, kernel=None, anchor=(-1, -1), iterations=1, borderType=BORDER_CONSTANT, borderValue=None, dst=None, roi=None):
    """cv::dilate (morph.dispatch.cpp:1024)."""
    return _morph(MORPH_DILATE, src, kernel, anchor, iterations, borderType, borderValue, dst, roi)


# ----------------------------------------------------------------------------- linear filters (a3, a4, a5)
def _np_kernel(k):
    k = np.asarray(k)
    if k.dtype not in (np.float32, np.float64, np.int32, np.uint8):
        k = k.astype(np.float64 if k.dtype.kind == "f" else np.float32)
    return np.ascontiguousarray(k)


_K_TYPE = {np.dtype(np.uint8): 0, np.dtype(np.int32): 4, np.dtype(np.float32): 5, np.dtype(np.float64): 6}


def _parent_geometry(src, roi, borderType):
    """(view, full_w, full_h, off_x, off_y): the locateROI() information cv:: passes to the HAL for non-isolated borders."""
    if roi is None:
        s = Img(src)
        return src, s, s.w, s.h, 0, 0
    x0, y0, w, h = roi
    view = src[y0:y0 + h, x0:x0 + w]
    s = Img(view)
    p = Img(src)
    if borderType & BORDER_ISOLATED:
        return view, s, s.w, s.h, 0, 0
    return view, s, p.w, p.h, x0, y0


def filter2D(src, ddepth, kernel, anchor=(-1, -1), delta=0.0, borderType=BORDER_DEFAULT, dst=None, roi=None):
    """cv::filter2D (filter.dispatch.cpp:1521-1553) through cv_hal_filterInit / cv_hal_filter / cv_hal_filterFree.

    `roi=(x, y, w, h)` filters a sub-rectangle of `src` the way a cv::Mat submatrix is filtered: borders read the
    parent's real pixels unless BORDER_ISOLATED is set."""
    view, s, fw, fh, ox, oy = _parent_geometry(src, roi, borderType)
    if ddepth < 0:
        ddepth = s.depth
    k = _np_kernel(kernel)
    if k.ndim == 1:
        k = k[None, :]
    kh, kw = k.shape
    ax, ay = anchor
    if ax < 0:
        ax = kw // 2                                     # normalizeAnchor (filterengine.hpp:352)
    if ay < 0:
        ay = kh // 2
    out = dst if dst is not None else empty_like_kind(view, s.h, s.w, s.cn, ddepth)
    d = Img(out)
    if d.ptr == s.ptr and roi is None:               # in place: FilterEngine's row ring makes that safe on the CPU; here the source is cloned
        view = _copy_like(view); s = Img(view)       # (with a roi the library declines a dst that overlaps the rows it reads)
    bind_stream(s, d)
    ctx = ctypes.c_void_p()
    rc = L.mi355cv_filterInit(ctypes.byref(ctx), k.ctypes.data, k.strides[0], _K_TYPE[k.dtype], kw, kh, s.w, s.h,
                              s.type, d.type, borderType & ~BORDER_ISOLATED, float(delta), ax, ay,
                              roi is not None, d.ptr == s.ptr)
    _lib.check(rc, "filterInit")
    try:
        rc = L.mi355cv_filter(ctx, _vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, fw, fh, ox, oy)
    finally:
        L.mi355cv_filterFree(ctx)
    _lib.check(rc, "filter")
    return out


def filter2DBatch(frames, ddepth, kernel, anchor=(-1, -1), delta=0.0, borderType=BORDER_DEFAULT, dst=None):
    """[N,H,W(,C)] device-resident frames, one launch (isolated borders: every frame is a whole image)."""
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    s0 = Img(frames[0])
    if ddepth < 0:
        ddepth = s0.depth
    if ddepth != s0.depth:
        raise NotImplementedError("filter2DBatch: ddepth must equal the source depth")
    k = _np_kernel(kernel)
    if k.ndim == 1:
        k = k[None, :]
    kh, kw = k.shape
    ax = kw // 2 if anchor[0] < 0 else anchor[0]
    ay = kh // 2 if anchor[1] < 0 else anchor[1]
    out = dst if dst is not None else _batch_alloc(frames, frames.shape, frames.dtype)
    d0 = Img(out[0])
    bind_stream(s0, d0)
    ctx = ctypes.c_void_p()
    _lib.check(L.mi355cv_filterInit(ctypes.byref(ctx), k.ctypes.data, k.strides[0], _K_TYPE[k.dtype], kw, kh, w, h, s0.type, d0.type,
                                    borderType & ~BORDER_ISOLATED, float(delta), ax, ay, False, False), "filterInit")
    try:
        rc = L.mi355cv_filterBatch(ctx, _vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz, n, w, h)
    finally:
        L.mi355cv_filterFree(ctx)
    _lib.check(rc, "filterBatch")
    return out


def cvtColorFilter2DBatch(frames, code, kernel, delta=0.0, borderType=BORDER_DEFAULT, dst=None):
    """filter2D(cvtColor(frame, code), -1, kernel) for [N,H,W,3|4] CV_8U device frames and code in {BGR2GRAY, RGB2GRAY, BGRA2GRAY, RGBA2GRAY}, in ONE
    pass over the colour frames (mi355cv_cvtBGRtoGrayFilterBatch): the gray frames are never written.  3x3 / 5x5 kernels, centred anchor, width a
    multiple of 16; the library declines anything else (make the two calls then)."""
    if code not in (COLOR_BGR2GRAY, COLOR_RGB2GRAY, COLOR_BGRA2GRAY, COLOR_RGBA2GRAY):
        raise NotImplementedError("cvtColorFilter2DBatch: a colour -> gray code")
    n, h, w, scn = (int(v) for v in frames.shape)
    if scn != (4 if code in (COLOR_BGRA2GRAY, COLOR_RGBA2GRAY) else 3):
        raise ValueError("channel count does not match the conversion code")
    k = _np_kernel(kernel)
    kh, kw = k.shape
    out = dst if dst is not None else _batch_alloc(frames, (n, h, w), frames.dtype)
    s0, d0 = Img(frames[0]), Img(out[0])
    bind_stream(s0, d0)
    ctx = ctypes.c_void_p()
    _lib.check(L.mi355cv_filterInit(ctypes.byref(ctx), k.ctypes.data, k.strides[0], _K_TYPE[k.dtype], kw, kh, w, h, d0.type, d0.type,
                                    borderType & ~BORDER_ISOLATED, float(delta), kw // 2, kh // 2, False, False), "filterInit")
    try:
        rc = L.mi355cv_cvtBGRtoGrayFilterBatch(ctx, _vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz,
                                               n, w, h, scn, code in (COLOR_RGB2GRAY, COLOR_RGBA2GRAY))
    finally:
        L.mi355cv_filterFree(ctx)
    _lib.check(rc, "cvtBGRtoGrayFilterBatch")
    return out


def sepFilter2D(src, ddepth, kernelX, kernelY, anchor=(-1, -1), delta=0.0, borderType=BORDER_DEFAULT, dst=None, roi=None):
    """cv::sepFilter2D (filter.dispatch.cpp:1555-1594) through cv_hal_sepFilterInit / sepFilter / sepFilterFree."""
    view, s, fw, fh, ox, oy = _parent_geometry(src, roi, borderType)
    if ddepth < 0:
        ddepth = s.depth
    kx = np.ascontiguousarray(np.asarray(kernelX, dtype=np.float64).ravel())
    ky = np.ascontiguousarray(np.asarray(kernelY, dtype=np.float64).ravel())
    out = dst if dst is not None else empty_like_kind(view, s.h, s.w, s.cn, ddepth)
    d = Img(out)
    if d.ptr == s.ptr and roi is None:
        view = _copy_like(view); s = Img(view)
    bind_stream(s, d)
    ctx = ctypes.c_void_p()
    rc = L.mi355cv_sepFilterInit(ctypes.byref(ctx), s.type, d.type, 6, kx.ctypes.data, len(kx), ky.ctypes.data, len(ky),
                                 anchor[0], anchor[1], float(delta), borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "sepFilterInit")
    try:
        rc = L.mi355cv_sepFilter(ctx, _vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, fw, fh, ox, oy)
    finally:
        L.mi355cv_sepFilterFree(ctx)
    _lib.check(rc, "sepFilter")
    return out


def Sobel(src, ddepth, dx, dy, ksize=3, scale=1.0, delta=0.0, borderType=BORDER_DEFAULT, dst=None, roi=None):
    """cv::Sobel (deriv.cpp:414-466) through cv_hal_sobel; ksize=-1 (FILTER_SCHARR) selects the Scharr taps."""
    view, s, fw, fh, ox, oy = _parent_geometry(src, roi, borderType)
    if ddepth < 0:
        ddepth = s.depth
    out = dst if dst is not None else empty_like_kind(view, s.h, s.w, s.cn, ddepth)
    d = Img(out)
    bind_stream(s, d)
    rc = L.mi355cv_sobel(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, d.depth, s.cn,
                         ox, oy, fw - s.w - ox, fh - s.h - oy, dx, dy, ksize, float(scale), float(delta),
                         borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "sobel")
    return out


def Scharr(src, ddepth, dx, dy, scale=1.0, delta=0.0, borderType=BORDER_DEFAULT, dst=None, roi=None):
    """cv::Scharr (deriv.cpp:468-) through cv_hal_scharr."""
    view, s, fw, fh, ox, oy = _parent_geometry(src, roi, borderType)
    if ddepth < 0:
        ddepth = s.depth
    out = dst if dst is not None else empty_like_kind(view, s.h, s.w, s.cn, ddepth)
    d = Img(out)
    bind_stream(s, d)
    rc = L.mi355cv_scharr(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, d.depth, s.cn,
                          ox, oy, fw - s.w - ox, fh - s.h - oy, dx, dy, float(scale), float(delta),
                          borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "scharr")
    return out


def boxFilter(src, ddepth, ksize, anchor=(-1, -1), normalize=True, borderType=BORDER_DEFAULT, dst=None, roi=None):
    """cv::boxFilter (box_filter.dispatch.cpp:440-489) through cv_hal_boxFilter."""
    view, s, fw, fh, ox, oy = _parent_geometry(src, roi, borderType)
    if ddepth < 0:
        ddepth = s.depth
    kw, kh = _ksize(ksize)
    if borderType != BORDER_CONSTANT and normalize and (borderType & BORDER_ISOLATED) != 0:   # :458-464
        if s.h == 1:
            kh = 1
        if s.w == 1:
            kw = 1
    out = dst if dst is not None else empty_like_kind(view, s.h, s.w, s.cn, ddepth)
    d = Img(out)
    bind_stream(s, d)
    rc = L.mi355cv_boxFilter(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.depth, d.depth, s.cn,
                             ox, oy, fw - s.w - ox, fh - s.h - oy, kw, kh, anchor[0], anchor[1], bool(normalize),
                             borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "boxFilter")
    return out


def blur(src, ksize, anchor=(-1, -1), borderType=BORDER_DEFAULT, dst=None):
    """cv::blur (box_filter.dispatch.cpp:492-499)."""
    return boxFilter(src, -1, ksize, anchor, True, borderType, dst)


# ----------------------------------------------------------------------------- geometric transforms (a7, a8, a9)
from .core import INTER_NEAREST, INTER_LINEAR, INTER_AREA, WARP_INVERSE_MAP, CV_MAKETYPE  # noqa: E402

INTER_MAX = 7


def _sat_int(v):
    return int(np.rint(v))            # saturate_cast<int>(double) == cvRound


def resize(src, dsize, fx=0.0, fy=0.0, interpolation=INTER_LINEAR, dst=None):
    """cv::resize (resize.cpp:4201-4246): dsize or (fx, fy); same-size -> copy (:4238); then cv_hal_resize."""
    s = Img(src)
    if dsize is None or dsize[0] == 0 or dsize[1] == 0:
        if not (fx > 0 and fy > 0):
            raise ValueError("resize: dsize or fx/fy required")
        dsize = (_sat_int(s.w * fx), _sat_int(s.h * fy))                 # :4218-4220
        if dsize[0] <= 0 or dsize[1] <= 0:
            raise ValueError("resize: empty destination")
    else:
        fx, fy = dsize[0] / s.w, dsize[1] / s.h                           # :4224-4225
    out = dst if dst is not None else empty_like_kind(src, dsize[1], dsize[0], s.cn, s.depth)
    d = Img(out)
    if (d.w, d.h) == (s.w, s.h):                                          # :4236-4241 plain copy
        out[...] = src
        return out
    bind_stream(s, d)
    rc = L.mi355cv_resize(s.type, _vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, d.w, d.h, float(fx), float(fy), interpolation)
    _lib.check(rc, "resize")
    return out


def getRotationMatrix2D(center, angle, scale):
    """cv::getRotationMatrix2D (imgwarp.cpp: getRotationMatrix2D_): center is a Point2f, math in double."""
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))
    a = angle * np.pi / 180.0
    alpha, beta = np.cos(a) * scale, np.sin(a) * scale
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def invertAffineTransform(M):
    """the inversion cv::warpAffine applies when WARP_INVERSE_MAP is not set (imgwarp.cpp:2824-2834)"""
    M = np.array(M, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11; M[0, 1] *= -D
    M[1, 0] *= -D; M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2] = b1; M[1, 2] = b2
    return M


def _border_value(v):
    bv = np.zeros(4, np.float64)
    if np.ndim(v) == 0:
        bv[0] = v                      # cv::Scalar(v) = (v, 0, 0, 0)
    else:
        bv[:len(v)] = v
    return bv


def warpAffine(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0.0, dst=None):
    """cv::warpAffine (imgwarp.cpp:2788-2903) -> cv_hal_warpAffine."""
    s = Img(src)
    interpolation = flags & INTER_MAX
    if interpolation == INTER_AREA:
        interpolation = INTER_LINEAR
    dw, dh = (s.w, s.h) if dsize is None or dsize[0] == 0 else dsize
    out = dst if dst is not None else empty_like_kind(src, dh, dw, s.cn, s.depth)
    d = Img(out)
    if d.ptr == s.ptr:
        src = _copy_like(src); s = Img(src)
    Mm = np.array(M, np.float64).reshape(2, 3)
    if not (flags & WARP_INVERSE_MAP):
        Mm = invertAffineTransform(Mm)
    Mm = np.ascontiguousarray(Mm)
    bv = _border_value(borderValue)
    bind_stream(s, d)
    rc = L.mi355cv_warpAffine(s.type, _vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, d.w, d.h, Mm.ctypes.data, interpolation,
                              borderMode, bv.ctypes.data)
    _lib.check(rc, "warpAffine")
    return out


def warpPerspective(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0.0, dst=None):
    """cv::warpPerspective (imgwarp.cpp:3370-3466) -> cv_hal_warpPerspective."""
    s = Img(src)
    interpolation = flags & INTER_MAX
    if interpolation == INTER_AREA:
        interpolation = INTER_LINEAR
    dw, dh = (s.w, s.h) if dsize is None or dsize[0] == 0 else dsize
    out = dst if dst is not None else empty_like_kind(src, dh, dw, s.cn, s.depth)
    d = Img(out)
    if d.ptr == s.ptr:
        src = _copy_like(src); s = Img(src)
    Mm = np.array(M, np.float64).reshape(3, 3)
    if not (flags & WARP_INVERSE_MAP):
        Mm = np.linalg.inv(Mm)         # cv::invert(matM, matM) :3407 (LU); parity tests pass WARP_INVERSE_MAP matrices
    Mm = np.ascontiguousarray(Mm)
    bv = _border_value(borderValue)
    bind_stream(s, d)
    rc = L.mi355cv_warpPerspective(s.type, _vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, d.w, d.h, Mm.ctypes.data, interpolation,
                                   borderMode, bv.ctypes.data)
    _lib.check(rc, "warpPerspective")
    return out


def remap(src, map1, map2=None, interpolation=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0.0, dst=None):
    """cv::remap (imgwarp.cpp:1718-1921).  A pair of CV_32FC1 maps goes through cv_hal_remap32f (:1820); the other representations -- one
    CV_32FC2 map, or the fixed-point maps of convertMaps (CV_16SC2 + CV_16UC1, CV_16SC2 alone for nearest) -- through mi355cv_remap."""
    s = Img(src)
    m1 = Img(map1)
    m2 = Img(map2) if map2 is not None else None
    if m2 is not None and (m1.w, m1.h) != (m2.w, m2.h):
        raise ValueError("remap: map sizes differ")
    out = dst if dst is not None else empty_like_kind(src, m1.h, m1.w, s.cn, s.depth)
    d = Img(out)
    if d.ptr == s.ptr:
        src = _copy_like(src); s = Img(src)
    bv = _border_value(borderValue)
    bind_stream(s, d)
    if m2 is not None and m1.type == CV_MAKETYPE(CV_32F, 1) and m2.type == CV_MAKETYPE(CV_32F, 1):
        rc = L.mi355cv_remap32f(s.type, _vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, d.w, d.h, _vp(m1.ptr), m1.step, _vp(m2.ptr), m2.step,
                                interpolation, borderMode, bv.ctypes.data)
        _lib.check(rc, "remap32f")
        return out
    rc = L.mi355cv_remap(s.type, _vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, d.w, d.h, _vp(m1.ptr), m1.step, m1.type,
                         _vp(m2.ptr) if m2 is not None else None, m2.step if m2 is not None else 0, m2.type if m2 is not None else 0,
                         interpolation, borderMode, bv.ctypes.data)
    _lib.check(rc, "remap")
    return out


def convertMaps(map1, map2, dstmap1type, nninterpolation=False):
    """cv::convertMaps (imgwarp.cpp:1925): (CV_32FC1, CV_32FC1) / CV_32FC2 -> (CV_16SC2, CV_16UC1) [CV_16SC2 alone with nninterpolation], and
    CV_16SC2 (+ CV_16UC1) -> (CV_32FC1, CV_32FC1) / CV_32FC2.  Returns (dstmap1, dstmap2 or None)."""
    m1 = Img(map1)
    m2 = Img(map2) if map2 is not None else None
    T16SC2, T16UC1, T32FC1, T32FC2 = CV_MAKETYPE(CV_16S, 2), CV_MAKETYPE(CV_16U, 1), CV_MAKETYPE(CV_32F, 1), CV_MAKETYPE(CV_32F, 2)
    if dstmap1type <= 0:
        dstmap1type = T32FC2 if m1.type == T16SC2 else T16SC2
    plane = map1[..., 0] if m1.cn > 1 else map1                  # a 2-D array of map1's kind: single-channel outputs are H x W, not H x W x 1
    if dstmap1type == T16SC2:
        o1 = empty_like_kind(map1, m1.h, m1.w, 2, CV_16S)
        o2 = None if nninterpolation else empty_like_kind(plane, m1.h, m1.w, 1, CV_16U)
    elif dstmap1type == T32FC1:
        o1, o2 = empty_like_kind(plane, m1.h, m1.w, 1, CV_32F), empty_like_kind(plane, m1.h, m1.w, 1, CV_32F)
    elif dstmap1type == T32FC2:
        o1, o2 = empty_like_kind(map1, m1.h, m1.w, 2, CV_32F), None
    else:
        raise ValueError("convertMaps: dstmap1type must be CV_16SC2, CV_32FC1 or CV_32FC2")
    a, b = Img(o1), (Img(o2) if o2 is not None else None)
    bind_stream(m1, a)
    rc = L.mi355cv_convertMaps(_vp(m1.ptr), m1.step, m1.type, _vp(m2.ptr) if m2 is not None else None, m2.step if m2 is not None else 0,
                               m2.type if m2 is not None else 0, _vp(a.ptr), a.step, dstmap1type, _vp(b.ptr) if b is not None else None,
                               b.step if b is not None else 0, m1.w, m1.h, 1 if nninterpolation else 0)
    _lib.check(rc, "convertMaps")
    return o1, o2


WARP_FILL_OUTLIERS, WARP_POLAR_LINEAR, WARP_POLAR_LOG = 8, 0, 256


def warpPolar(src, dsize, center, maxRadius, flags, dst=None):
    """cv::warpPolar (imgwarp.cpp:3731): Cartesian -> polar / semi-log polar, and with WARP_INVERSE_MAP the way back (the map is evaluated in the kernel
    with the reference's float approximations of cartToPolar / log)."""
    s = Img(src)
    dw, dh = dsize
    if dw <= 0 and dh <= 0:                                       # :3737-3745
        dw, dh = _cvRound(maxRadius), _cvRound(maxRadius * np.pi)
    elif dh <= 0:
        dh = _cvRound(dw * np.pi)
    if dst is not None:
        out = dst
    else:
        out = empty_like_kind(src, dh, dw, s.cn, s.depth)
        out[...] = 0                                              # Mat::create leaves new memory as it is; BORDER_TRANSPARENT keeps it -- start from zeros
    d = Img(out)
    bind_stream(s, d)
    rc = L.mi355cv_warpPolar(s.type, _vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, d.w, d.h, float(center[0]), float(center[1]), float(maxRadius), flags)
    _lib.check(rc, "warpPolar")
    return out


# ----------------------------------------------------------------------------- frame batches of the single-image hooks
def _batch_geom(frames):
    """frames: a tensor [N,H,W(,C)] -- resident in HBM (one launch), or a CPU tensor (ideally page-locked): the library then moves the batch across
    PCIe in chunks through two sets of device buffers, upload / kernels / download overlapped (rt.h runHostBatch)"""
    if torch is None or not isinstance(frames, torch.Tensor) or frames.dim() not in (3, 4):
        raise ValueError("batch entries take a tensor [N,H,W(,C)] (CUDA(ROCm) resident, or a CPU tensor for the pipelined host path)")
    return int(frames.shape[0]), Img(frames[0])


def _batch_alloc(frames, shape, dtype):
    if frames.is_cuda:
        return torch.empty(shape, dtype=dtype, device=frames.device)
    return torch.empty(shape, dtype=dtype, pin_memory=frames.is_pinned())


def _batch_out(frames, dst, shape, dtype):
    if dst is not None:
        if tuple(dst.shape) != tuple(shape) or dst.dtype != dtype or dst.is_cuda != frames.is_cuda:
            raise ValueError("dst geometry mismatch")
        return dst
    return _batch_alloc(frames, shape, dtype)


def SobelBatch(frames, ddepth, dx, dy, ksize=3, scale=1.0, delta=0.0, borderType=BORDER_DEFAULT, dst=None):
    """cv::Sobel over [N,H,W(,C)] device frames, one launch (ksize=-1: Scharr taps)"""
    n, s0 = _batch_geom(frames)
    if ddepth < 0:
        ddepth = s0.depth
    out = _batch_out(frames, dst, frames.shape, _DEPTH_T[ddepth])
    d0 = Img(out[0])
    bind_stream(s0, d0)
    rc = L.mi355cv_sobelBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz, n, s0.w, s0.h, s0.depth, d0.depth,
                              s0.cn, dx, dy, ksize, float(scale), float(delta), borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "sobelBatch")
    return out


def boxFilterBatch(frames, ddepth, ksize, anchor=(-1, -1), normalize=True, borderType=BORDER_DEFAULT, dst=None):
    n, s0 = _batch_geom(frames)
    if ddepth < 0:
        ddepth = s0.depth
    out = _batch_out(frames, dst, frames.shape, _DEPTH_T[ddepth])
    d0 = Img(out[0])
    bind_stream(s0, d0)
    rc = L.mi355cv_boxFilterBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz, n, s0.w, s0.h, s0.depth,
                                  d0.depth, s0.cn, ksize[0], ksize[1], anchor[0], anchor[1], bool(normalize), borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "boxFilterBatch")
    return out


def sepFilter2DBatch(frames, ddepth, kernelX, kernelY, anchor=(-1, -1), delta=0.0, borderType=BORDER_DEFAULT, dst=None):
    n, s0 = _batch_geom(frames)
    if ddepth < 0:
        ddepth = s0.depth
    out = _batch_out(frames, dst, frames.shape, _DEPTH_T[ddepth])
    d0 = Img(out[0])
    kx = np.ascontiguousarray(np.asarray(kernelX, dtype=np.float64).ravel())
    ky = np.ascontiguousarray(np.asarray(kernelY, dtype=np.float64).ravel())
    bind_stream(s0, d0)
    ctx = ctypes.c_void_p()
    _lib.check(L.mi355cv_sepFilterInit(ctypes.byref(ctx), s0.type, d0.type, 6, kx.ctypes.data, len(kx), ky.ctypes.data, len(ky), anchor[0], anchor[1], float(delta),
                                       borderType & ~BORDER_ISOLATED), "sepFilterInit")
    try:
        rc = L.mi355cv_sepFilterBatch(ctx, _vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz, n, s0.w, s0.h)
    finally:
        L.mi355cv_sepFilterFree(ctx)
    _lib.check(rc, "sepFilterBatch")
    return out


def thresholdBatch(frames, thresh, maxval, type, dst=None):
    """cv::threshold with a fixed level over [N,H,W(,C)] device frames (the automatic levels are per image: use threshold)"""
    n, s0 = _batch_geom(frames)
    if type & ~7:
        raise NotImplementedError("thresholdBatch: fixed levels only")
    out = _batch_out(frames, dst, frames.shape, frames.dtype)
    d0 = Img(out[0])
    if s0.depth != CV_8U:
        raise NotImplementedError("thresholdBatch: CV_8U frames")
    ith = int(np.floor(thresh))                              # thresh.cpp:1583-1610: integer levels for integer images
    if ith < 0 or ith >= 255:
        raise NotImplementedError("thresholdBatch: degenerate levels are constant fills (use threshold)")
    th = float(ith)
    mv = float(ith if type == THRESH_TRUNC else min(max(int(np.rint(maxval)), 0), 255))
    bind_stream(s0, d0)
    rc = L.mi355cv_thresholdBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz, n, s0.w, s0.h, s0.depth, s0.cn,
                                  th, mv, type)
    _lib.check(rc, "thresholdBatch")
    return out


def resizeBatch(frames, dsize, fx=0.0, fy=0.0, interpolation=INTER_LINEAR, dst=None):
    n, s0 = _batch_geom(frames)
    if dsize is None or dsize[0] == 0:
        dsize = (_cvRound(s0.w * fx), _cvRound(s0.h * fy))    # saturate_cast<int>(src.cols * fx), resize.cpp:4217
    else:
        fx, fy = dsize[0] / s0.w, dsize[1] / s0.h
    shape = (n, dsize[1], dsize[0]) + tuple(frames.shape[3:])
    out = _batch_out(frames, dst, shape, frames.dtype)
    d0 = Img(out[0])
    bind_stream(s0, d0)
    rc = L.mi355cv_resizeBatch(s0.type, _vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, s0.w, s0.h, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz, d0.w, d0.h, n,
                               float(fx), float(fy), interpolation)
    _lib.check(rc, "resizeBatch")
    return out


def _warpBatch(fn, name, nM, frames, M, dsize, flags, borderMode, borderValue, dst, invert):
    n, s0 = _batch_geom(frames)
    interpolation = flags & INTER_MAX
    dw, dh = (s0.w, s0.h) if dsize is None or dsize[0] == 0 else dsize
    shape = (n, dh, dw) + tuple(frames.shape[3:])
    out = _batch_out(frames, dst, shape, frames.dtype)
    d0 = Img(out[0])
    Mm = np.array(M, np.float64).reshape(nM // 3, 3)
    if not (flags & WARP_INVERSE_MAP):
        Mm = invert(Mm)
    Mm = np.ascontiguousarray(Mm)
    bv = _border_value(borderValue)
    bind_stream(s0, d0)
    rc = fn(s0.type, _vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, s0.w, s0.h, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz, d0.w, d0.h, n,
            Mm.ctypes.data, interpolation, borderMode, bv.ctypes.data)
    _lib.check(rc, name)
    return out


def warpAffineBatch(frames, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0.0, dst=None):
    """cv::warpAffine with one matrix over [N,H,W(,C)] device frames, one launch"""
    return _warpBatch(L.mi355cv_warpAffineBatch, "warpAffineBatch", 6, frames, M, dsize, flags, borderMode, borderValue, dst, invertAffineTransform)


def warpPerspectiveBatch(frames, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0.0, dst=None):
    return _warpBatch(L.mi355cv_warpPerspectiveBatch, "warpPerspectiveBatch", 9, frames, M, dsize, flags, borderMode, borderValue, dst, lambda m: np.linalg.inv(m))


# ----------------------------------------------------------------------------- pyramids and corners (a10, a11, a12)
def pyrDown(src, dstsize=None, borderType=BORDER_DEFAULT, dst=None, margins=None):
    """cv::pyrDown (pyramids.cpp:1348-1395) -> cv_hal_pyrdown / cv_hal_pyrdown_offset."""
    if (borderType & ~BORDER_ISOLATED) == BORDER_CONSTANT:
        raise ValueError("pyrDown: BORDER_CONSTANT is not allowed")               # CV_Assert :1352
    s = Img(src)
    dw, dh = ((s.w + 1) // 2, (s.h + 1) // 2) if dstsize is None or dstsize[0] == 0 else dstsize
    out = dst if dst is not None else empty_like_kind(src, dh, dw, s.cn, s.depth)
    d = Img(out)
    bind_stream(s, d)
    if margins is not None and not (borderType & BORDER_ISOLATED):
        rc = L.mi355cv_pyrdown_offset(_vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, d.w, d.h, s.depth, s.cn,
                                      margins[0], margins[1], margins[2], margins[3], borderType & ~BORDER_ISOLATED)
    else:
        rc = L.mi355cv_pyrdown(_vp(s.ptr), s.step, s.w, s.h, _vp(d.ptr), d.step, d.w, d.h, s.depth, s.cn, borderType)
    _lib.check(rc, "pyrdown")
    return out


def pyrDownBatch(frames, borderType=BORDER_DEFAULT, dst=None):
    """cv::pyrDown over [N,H,W(,C)] frames -> [N,(H+1)/2,(W+1)/2(,C)], one launch (host-resident batches: the pipelined path)"""
    if (borderType & ~BORDER_ISOLATED) == BORDER_CONSTANT:
        raise ValueError("pyrDown: BORDER_CONSTANT is not allowed")
    n, s0 = _batch_geom(frames)
    shape = (n, (s0.h + 1) // 2, (s0.w + 1) // 2) + tuple(frames.shape[3:])
    out = _batch_out(frames, dst, shape, frames.dtype)
    d0 = Img(out[0])
    bind_stream(s0, d0)
    rc = L.mi355cv_pyrdownBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, s0.w, s0.h, _vp(d0.ptr), d0.step, int(out.stride(0)) * d0.esz, d0.w, d0.h, n,
                                s0.depth, s0.cn, borderType & ~BORDER_ISOLATED)
    _lib.check(rc, "pyrdownBatch")
    return out


def buildPyramid(src, maxlevel, borderType=BORDER_DEFAULT):
    """cv::buildPyramid (pyramids.cpp:1616-1643): [src, level1, ..., level maxlevel]; one C-ABI call for all levels."""
    if (borderType & ~BORDER_ISOLATED) == BORDER_CONSTANT:
        raise ValueError("buildPyramid: BORDER_CONSTANT is not allowed")
    s = Img(src)
    levels, w, h = [src], s.w, s.h
    for _ in range(maxlevel):
        w, h = (w + 1) // 2, (h + 1) // 2
        levels.append(empty_like_kind(src, h, w, s.cn, s.depth))
    if maxlevel > 0:
        imgs = [Img(l) for l in levels[1:]]
        ptrs = (ctypes.c_void_p * maxlevel)(*[i.ptr for i in imgs])
        steps = (ctypes.c_size_t * maxlevel)(*[i.step for i in imgs])
        bind_stream(s, imgs[0])
        _lib.check(L.mi355cv_buildPyramid(_vp(s.ptr), s.step, s.w, s.h, s.depth, s.cn, ptrs, steps, maxlevel, borderType), "buildPyramid")
    return levels


def buildPyramidBatch(frames, maxlevel, borderType=BORDER_DEFAULT, dst=None):
    """[N,H,W(,C)] frames -> list of per-level batches [frames, level 1, ...]; every level of every frame is enqueued by one call
    (mi355cv_buildPyramidBatch).  `dst`: the list a previous call returned, to reuse its level arrays.  Frames in HBM: levels in HBM; a CPU tensor
    (ideally page-locked): the batch crosses PCIe in chunks, the upload of the next chunk under the kernels and the downloads of this one's levels
    (rt.h runHostBatchN)"""
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    cn = int(frames.shape[3]) if frames.dim() == 4 else 1
    out = [frames]
    for l in range(maxlevel):
        w, h = (w + 1) // 2, (h + 1) // 2
        shape = (n, h, w) + ((cn,) if frames.dim() == 4 else ())
        lvl = dst[l + 1] if dst is not None else _batch_alloc(frames, shape, frames.dtype)
        if tuple(lvl.shape) != shape or lvl.dtype != frames.dtype or lvl.is_cuda != frames.is_cuda:
            raise ValueError("dst level geometry mismatch")
        out.append(lvl)
    if maxlevel < 1:
        return out
    s0 = Img(frames[0])
    imgs = [Img(l[0]) for l in out[1:]]
    ptrs = (ctypes.c_void_p * maxlevel)(*[i.ptr for i in imgs])
    steps = (ctypes.c_size_t * maxlevel)(*[i.step for i in imgs])
    strides = (ctypes.c_size_t * maxlevel)(*[int(l.stride(0)) * s0.esz for l in out[1:]])
    bind_stream(s0, imgs[0])
    rc = L.mi355cv_buildPyramidBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, s0.w, s0.h, s0.depth, cn, ptrs, steps, strides, maxlevel, n, borderType)
    _lib.check(rc, "buildPyramidBatch")
    return out


def cornerHarris(src, blockSize, ksize, k, borderType=BORDER_DEFAULT, dst=None):
    """cv::cornerHarris (corner.cpp:634-653): CV_8UC1 / CV_32FC1 -> CV_32FC1, fused on the GPU."""
    s = Img(src)
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, 1, CV_32F)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_cornerHarris(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.type, blockSize, ksize, float(k), borderType), "cornerHarris")
    return out


def cornerMinEigenVal(src, blockSize, ksize=3, borderType=BORDER_DEFAULT, dst=None):
    """cv::cornerMinEigenVal (corner.cpp:604-631)."""
    s = Img(src)
    out = dst if dst is not None else empty_like_kind(src, s.h, s.w, 1, CV_32F)
    d = Img(out)
    bind_stream(s, d)
    _lib.check(L.mi355cv_cornerMinEigenVal(_vp(s.ptr), s.step, _vp(d.ptr), d.step, s.w, s.h, s.type, blockSize, ksize, borderType), "cornerMinEigenVal")
    return out


def cornerHarrisBatch(frames, blockSize, ksize, k, borderType=BORDER_DEFAULT, dst=None):
    """[N,H,W] device frames (uint8 or float32) -> [N,H,W] float32 responses, one launch."""
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    out = dst if dst is not None else _batch_alloc(frames, (n, h, w), torch.float32)
    s0, d0 = Img(frames[0]), Img(out[0])
    bind_stream(s0, d0)
    rc = L.mi355cv_cornerHarrisBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, _vp(d0.ptr), d0.step, int(out.stride(0)) * 4, n, w, h,
                                     s0.type, blockSize, ksize, float(k), borderType)
    _lib.check(rc, "cornerHarrisBatch")
    return out


def goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, mask=None, blockSize=3, gradientSize=3,
                        useHarrisDetector=False, k=0.04, returnQuality=False):
    """cv::goodFeaturesToTrack (featureselect.cpp:382-548): Nx2 float32 corner array (x, y)."""
    s = Img(image)
    cap = maxCorners if maxCorners > 0 else s.w * s.h
    corners = np.zeros((cap, 2), np.float32)
    quality = np.zeros(cap, np.float32)
    m = Img(mask) if mask is not None else None
    bind_stream(s)
    n = L.mi355cv_goodFeaturesToTrack(_vp(s.ptr), s.step, s.w, s.h, s.type, corners.ctypes.data, quality.ctypes.data, maxCorners,
                                      float(qualityLevel), float(minDistance), _vp(m.ptr) if m else None, m.step if m else 0,
                                      blockSize, gradientSize, int(useHarrisDetector), float(k))
    if n < 0:
        _lib.check(1 if n == -1 else -1, "goodFeaturesToTrack")
    return (corners[:n].copy(), quality[:n].copy()) if returnQuality else corners[:n].copy()


# ----------------------------------------------------------------------------- template matching (a13)
TM_SQDIFF, TM_SQDIFF_NORMED, TM_CCORR, TM_CCORR_NORMED, TM_CCOEFF, TM_CCOEFF_NORMED = range(6)


def matchTemplate(image, templ, method, result=None, mask=None):
    """cv::matchTemplate (templmatch.cpp:1158-1194): CV_8U / CV_32F, 1..4 channels, all six methods.  With a mask (CV_8U or CV_32F, the template's size, one
    channel or the template's) the reference takes matchTemplateMask (:762-904): mi355cv_matchTemplateMask."""
    s, t = Img(image), Img(templ)
    if mask is not None:
        m = Img(mask)
        if (s.depth, s.cn) != (t.depth, t.cn) or (m.w, m.h) != (t.w, t.h) or m.depth not in (CV_8U, CV_32F) or m.cn not in (1, t.cn):
            raise ValueError("matchTemplate: image / template types differ, or the mask is not CV_8U / CV_32F of the template's size")     # CV_Assert :764-766, :1164
        if s.w < t.w or s.h < t.h:
            raise ValueError("matchTemplate: with a mask the template may not be larger than the image")                                     # CV_Assert :767
        out = result if result is not None else empty_like_kind(image if s.cn == 1 else image[..., 0], s.h - t.h + 1, s.w - t.w + 1, 1, CV_32F)
        d = Img(out)
        bind_stream(s, d)
        rc = L.mi355cv_matchTemplateMask(_vp(s.ptr), s.step, s.w, s.h, _vp(t.ptr), t.step, t.w, t.h, s.type, _vp(m.ptr), m.step, m.type, _vp(d.ptr), d.step, method)
        _lib.check(rc, "matchTemplateMask")
        return out
    if (s.depth, s.cn) != (t.depth, t.cn):
        raise ValueError("matchTemplate: image and template must have the same type")        # CV_Assert :1164
    if s.w < t.w or s.h < t.h:
        raise NotImplementedError("matchTemplate: template larger than image (the reference swaps them)")
    out = result if result is not None else empty_like_kind(image if s.cn == 1 else image[..., 0], s.h - t.h + 1, s.w - t.w + 1, 1, CV_32F)
    d = Img(out)
    bind_stream(s, d)
    rc = L.mi355cv_matchTemplate(_vp(s.ptr), s.step, s.w, s.h, _vp(t.ptr), t.step, t.w, t.h, s.type, _vp(d.ptr), d.step, method)
    _lib.check(rc, "matchTemplate")
    return out


def matchTemplateBatch(frames, templ, method, result=None):
    """[N,H,W(,C)] frames x one template -> [N,H-h+1,W-w+1] float32.  Frames (and results) in HBM, or both in host memory (a CPU tensor, ideally
    page-locked: chunks through two sets of device buffers, rt.h runHostBatch); the template may live on either side."""
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    t = Img(templ)
    out = result if result is not None else _batch_alloc(frames, (n, h - t.h + 1, w - t.w + 1), torch.float32)
    if out.is_cuda != frames.is_cuda:
        raise ValueError("frames and results live in the same kind of memory")
    s0, d0 = Img(frames[0]), Img(out[0])
    bind_stream(s0, d0)
    rc = L.mi355cv_matchTemplateBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)) * s0.esz, n, w, h, _vp(t.ptr), t.step, t.w, t.h, s0.type,
                                      _vp(d0.ptr), d0.step, int(out.stride(0)) * 4, method)
    _lib.check(rc, "matchTemplateBatch")
    return out


def integral(src, sqsum=False, sdepth=-1, sqdepth=-1, tilted=False):
    """cv::integral: (H+1)x(W+1)[xC] sum and, optionally, squared and tilted sums.  sdepth / sqdepth -1 = the reference's defaults (CV_32S sums for 8-bit sources,
    CV_64F otherwise; CV_64F squared sums, sumpixels.dispatch.cpp:417-424); any row of its type table (:383-406).  Returns sum, (sum, sqsum), (sum, tilted) or
    (sum, sqsum, tilted)."""
    s = Img(src)
    CV_32S, CV_32F = 4, 5
    if sdepth <= 0:
        sdepth = CV_32S if s.depth == CV_8U else CV_64F
    if sqdepth <= 0:
        sqdepth = CV_64F
    if sdepth not in (CV_32S, CV_32F, CV_64F) or sqdepth not in (CV_32S, CV_32F, CV_64F):
        raise NotImplementedError("integral: sdepth / sqdepth")
    shape = (s.h + 1, s.w + 1) if s.cn == 1 else (s.h + 1, s.w + 1, s.cn)
    on_dev = torch is not None and isinstance(src, torch.Tensor)

    def new(depth):
        if on_dev:
            return torch.empty(shape, dtype={CV_32S: torch.int32, CV_32F: torch.float32, CV_64F: torch.float64}[depth], device=src.device)
        return np.empty(shape, {CV_32S: np.int32, CV_32F: np.float32, CV_64F: np.float64}[depth])
    sm, sq, tl = new(sdepth), (new(sqdepth) if sqsum else None), (new(sdepth) if tilted else None)
    a, b, c = Img(sm), (Img(sq) if sqsum else None), (Img(tl) if tilted else None)
    bind_stream(s, a)
    rc = L.mi355cv_integral(s.depth, sdepth, sqdepth, _vp(s.ptr), s.step, _vp(a.ptr), a.step, _vp(b.ptr) if b else None, b.step if b else 0,
                            _vp(c.ptr) if c else None, c.step if c else 0, s.w, s.h, s.cn)
    _lib.check(rc, "integral")
    out = tuple(x for x in (sm, sq, tl) if x is not None)
    return out if len(out) > 1 else sm


def integralBatch(frames, sqsum=False, sdepth=-1, dst=None):
    """cv::integral over [N,H,W] device-resident CV_8U frames -> [N,H+1,W+1] int32 (or float64) sums [, float64 squared sums]: one set of launches
    for the batch.  `dst`: the array (or pair) a previous call returned."""
    if torch is None or not isinstance(frames, torch.Tensor) or not frames.is_cuda or frames.dim() != 3 or frames.dtype != torch.uint8:
        raise ValueError("integralBatch takes a CUDA(ROCm) uint8 tensor [N,H,W]")
    CV_32S, CV_64F = 4, 6
    if sdepth <= 0:
        sdepth = CV_32S
    if sdepth not in (CV_32S, CV_64F):
        raise NotImplementedError("integralBatch: sdepth")
    n, h, w = (int(v) for v in frames.shape)
    if dst is not None:
        sm, sq = dst if sqsum else (dst, None)
    else:
        sm = torch.empty((n, h + 1, w + 1), dtype=torch.int32 if sdepth == CV_32S else torch.float64, device=frames.device)
        sq = torch.empty((n, h + 1, w + 1), dtype=torch.float64, device=frames.device) if sqsum else None
    s0, a0 = Img(frames[0]), Img(sm[0])
    b0 = Img(sq[0]) if sqsum else None
    bind_stream(s0, a0)
    rc = L.mi355cv_integralBatch(_vp(s0.ptr), s0.step, int(frames.stride(0)), _vp(a0.ptr), a0.step, int(sm.stride(0)) * a0.esz,
                                 _vp(b0.ptr) if b0 else None, b0.step if b0 else 0, int(sq.stride(0)) * 8 if sqsum else 0, n, w, h, sdepth)
    _lib.check(rc, "integralBatch")
    return (sm, sq) if sqsum else sm
