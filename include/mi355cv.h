/*
 * mi355cv.h -- C ABI of libmi355cv.so: the MI355X (gfx950) implementation of
 * OpenCV's imgproc hot path, shaped as a drop-in for the reference's imgproc
 * HAL replacement interface (reference: modules/imgproc/src/hal_replacement.hpp).
 *
 * Every `mi355cv_<hook>` below has EXACTLY the parameter list of the
 * `hal_ni_<hook>` stub it replaces (file:line cited per function), so a HAL
 * header only has to `#undef cv_hal_<hook>` / `#define cv_hal_<hook> mi355cv_<hook>`
 * (see include/mi355cv_hal.hpp and INTEGRATION.md).
 *
 * Contract (hal_replacement.hpp:1342-1357, core/hal/interface.h:9-11):
 *   return 0  (CV_HAL_ERROR_OK)              -> dst holds the result
 *   return 1  (CV_HAL_ERROR_NOT_IMPLEMENTED) -> nothing was written; caller falls back to its CPU path
 *   return <0 (CV_HAL_ERROR_UNKNOWN)         -> device failure after dst may have been touched
 * Hooks never throw.  Image pointers may be plain host memory (staged through
 * a pinned bounce buffer + H2D/D2H, synchronous) or device / managed memory
 * (launched in place on the calling thread's stream; synchronous unless
 * mi355cv_setAsync(1)).  There is NO CPU fallback inside this library.
 *
 * Plain C: pointers, sizes, ints and doubles only.
 */
#ifndef MI355CV_H
#define MI355CV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef MI355CV_API
#define MI355CV_API __attribute__((visibility("default")))
#endif

typedef unsigned char mi355cv_uchar;

/* return codes == CV_HAL_ERROR_* (modules/core/include/opencv2/core/hal/interface.h:9-11) */
#define MI355CV_OK               0
#define MI355CV_NOT_IMPLEMENTED  1
#define MI355CV_ERROR_UNKNOWN   -1

/* depth / border / interpolation constants, numerically equal to the reference's
 * (core/hal/interface.h:66-80, core/base.hpp:332-345, imgproc.hpp:248-294) */
#define MI355CV_8U 0
#define MI355CV_8S 1
#define MI355CV_16U 2
#define MI355CV_16S 3
#define MI355CV_32S 4
#define MI355CV_32F 5
#define MI355CV_64F 6
#define MI355CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define MI355CV_MAT_DEPTH(type) ((type) & 7)
#define MI355CV_MAT_CN(type) ((((type) >> 3) & 511) + 1)

#define MI355CV_BORDER_CONSTANT    0
#define MI355CV_BORDER_REPLICATE   1
#define MI355CV_BORDER_REFLECT     2
#define MI355CV_BORDER_WRAP        3
#define MI355CV_BORDER_REFLECT_101 4
#define MI355CV_BORDER_TRANSPARENT 5
#define MI355CV_BORDER_ISOLATED    16

#define MI355CV_INTER_NEAREST 0
#define MI355CV_INTER_LINEAR  1
#define MI355CV_INTER_CUBIC   2
#define MI355CV_INTER_AREA    3
#define MI355CV_INTER_LINEAR_EXACT 5
#define MI355CV_WARP_INVERSE_MAP 16

/* ------------------------------------------------------------------ runtime */

/* library / device bring-up; returns 0 when a gfx950 device is usable */
MI355CV_API int  mi355cv_init(int device);
MI355CV_API const char* mi355cv_version(void);
MI355CV_API const char* mi355cv_lastError(void);
/* stream used for launches made by the calling thread: exactly this hipStream_t (NULL = HIP's null
 * stream).  Until called -- or after mi355cv_resetStream() -- a library-owned per-thread stream is used. */
MI355CV_API int  mi355cv_setStream(void* hipStream);
MI355CV_API int  mi355cv_resetStream(void);
/* 1: calls on device-resident images return after enqueue (caller synchronises); 0 (default): synchronous */
MI355CV_API int  mi355cv_setAsync(int enable);
MI355CV_API int  mi355cv_synchronize(void);
/* number of times the named entry point ran its GPU path to completion in this process
 * (the analogue of the reference's CV_IMPL_ADD bookkeeping, core/private.hpp) */
MI355CV_API long long mi355cv_callCount(const char* entry);
/* experiment knobs (see tools/tune_gauss.py) and the streaming-copy probe used as the measured-copy
 * roofline denominator */
MI355CV_API int mi355cv_setParam(const char* key, int value);
MI355CV_API int mi355cv_copyProbe(const void* src_dev, void* dst_dev, size_t bytes, int perThread, int nontemporal);
MI355CV_API int mi355cv_copyProbeColwalk(const void* src_dev, void* dst_dev, int width_bytes, int height, int nframes,
                                         int segRows, int unroll);
/* device memory helpers for hosts without a HIP binding (images that live in HBM) */
MI355CV_API void* mi355cv_deviceAlloc(size_t bytes);
MI355CV_API int   mi355cv_deviceFree(void* p);
MI355CV_API int   mi355cv_upload(void* dst_dev, const void* src_host, size_t bytes);
MI355CV_API int   mi355cv_download(void* dst_host, const void* src_dev, size_t bytes);

/* --------------------------------------------------- a1: Gaussian smoothing */

/* replaces hal_ni_gaussianBlurBinomial (hal_replacement.hpp:1169); caller: cv::GaussianBlur
 * smooth.dispatch.cpp:696 (8U) / :767 (16U).  Implemented: depth 8U, cn 1..4, ksize 3 or 5,
 * borders CONSTANT/REPLICATE/REFLECT/WRAP/REFLECT_101; bit-exact with fixedSmoothInvoker
 * (smooth.simd.hpp:1926). */
MI355CV_API int mi355cv_gaussianBlurBinomial(const mi355cv_uchar* src_data, size_t src_step,
        mi355cv_uchar* dst_data, size_t dst_step, int width, int height, int depth, int cn,
        size_t margin_left, size_t margin_top, size_t margin_right, size_t margin_bottom,
        size_t ksize, int border_type);

/* replaces hal_ni_gaussianBlur (hal_replacement.hpp:1146); callers smooth.dispatch.cpp:708,778,813.
 * Implemented: depth 8U (Q8.8 fixed-point path of GaussianBlurFixedPoint, bit-exact for the
 * sigma==0 tables; sigma>0 kernels are generated as getGaussianKernelBitExact does) and 32F. */
MI355CV_API int mi355cv_gaussianBlur(const mi355cv_uchar* src_data, size_t src_step,
        mi355cv_uchar* dst_data, size_t dst_step, int width, int height, int depth, int cn,
        size_t margin_left, size_t margin_top, size_t margin_right, size_t margin_bottom,
        size_t ksize_width, size_t ksize_height, double sigmaX, double sigmaY, int border_type);

/* a2: host-side tap generators.  getGaussianKernelBitExact (smooth.dispatch.cpp:81-198, also behind
 * cv::getGaussianKernel :200) and getGaussianKernelFixedPoint_ED (:224-258; fractionBits 8 -> the Q8.8
 * taps of the CV_8U path, 16 -> Q16.16 of CV_16U).  Bit-identical to the reference's softdouble code. */
MI355CV_API int mi355cv_getGaussianKernel(int n, double sigma, double* taps);
MI355CV_API int mi355cv_getGaussianKernelQ(int n, double sigma, int fractionBits, int64_t* taps);

/* Fixed-point separable smoothing with caller-supplied Q8.8 taps: the body of
 * GaussianBlurFixedPoint<uint16_t> (smooth.simd.hpp:2219; taps as produced by
 * getGaussianKernelFixedPoint_ED, smooth.dispatch.cpp:224).  dst = (sum_j ky[j]*sum_i kx[i]*p + 2^15) >> 16. */
MI355CV_API int mi355cv_sepSmoothFixedU8(const mi355cv_uchar* src_data, size_t src_step,
        mi355cv_uchar* dst_data, size_t dst_step, int width, int height, int cn,
        size_t margin_left, size_t margin_top, size_t margin_right, size_t margin_bottom,
        const uint16_t* kx, int kxlen, const uint16_t* ky, int kylen, int border_type);

/* Batched form (frames are independent units): `nframes` images of identical
 * geometry, frame f at src_data + f*src_frame_stride; one launch, grid-z = frame. */
MI355CV_API int mi355cv_gaussianBlurBinomialBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes,
        int width, int height, int depth, int cn, size_t ksize, int border_type);

#ifdef __cplusplus
}
#endif
#endif /* MI355CV_H */
