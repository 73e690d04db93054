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
#ifndef __cplusplus
#include <stdbool.h>
#endif

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
#define MI355CV_INTER_LANCZOS4 4
#define MI355CV_INTER_AREA    3
#define MI355CV_INTER_LINEAR_EXACT 5
#define MI355CV_INTER_NEAREST_EXACT 6
#define MI355CV_WARP_INVERSE_MAP 16

/* ------------------------------------------------------------------ runtime */

/* library / device bring-up; returns 0 when a gfx950 device is usable.  `device` >= 0 names the process default device (first call wins;
 * otherwise MI355CV_DEVICE, or the device the host program had current at first use). */
MI355CV_API int  mi355cv_init(int device);
/* multi-GPU in one process (one host thread + its streams per device, frames sharded by index, no data-path collective):
 * mi355cv_setDevice binds the CALLING THREAD's hooks to a device ordinal (-1: back to the process default).  Streams, events and the scratch
 * pool are kept per thread AND device; the host program's own current HIP device is put back when a hook returns; an image that lives in
 * another GPU's memory makes the hook answer NOT_IMPLEMENTED (mi355cv_lastError names both devices).  Returns 0, or -1 for an ordinal that
 * does not exist / is not gfx950. */
MI355CV_API int  mi355cv_deviceCount(void);
MI355CV_API int  mi355cv_setDevice(int device);
MI355CV_API int  mi355cv_getDevice(void);
/* Batches across the GPUs of one node from a C / C++ host (csrc/shard.hip): device slot g of ndev takes the g-th of ndev contiguous blocks of frames whose sizes
 * differ by at most one ([g*B/ndev, (g+1)*B/ndev) when ndev divides B; the partition bench.py --gpus N and opencv_amd/shard.py use) -- and mi355cv_runSharded runs fn(user, slot, device, first, count) for every non-empty slot on its own host thread,
 * bound to devices[slot] (NULL: ordinals 0 .. ndev-1; an ordinal may repeat) when bind != 0; fn calls the ordinary mi355cv_* entry points on its frames, which must live
 * on that device (or in host / managed memory).  Returns 0, or the first non-zero code in slot order (mi355cv_lastError names slot and device).  There is no data-path
 * collective: parameters given as host arguments are uploaded by each device's own hooks; a device-resident parameter image (a matchTemplate template) is copied to every
 * device with mi355cv_replicate (one upload + an RCCL ncclBroadcast over xGMI when the list names more than one device; one hipMemcpy per device otherwise or with MI355CV_REPLICATE=copy; the Python layer broadcasts through torch.distributed). */
MI355CV_API void mi355cv_shardRange(int nframes, int ndev, int slot, int* first, int* count);
MI355CV_API int  mi355cv_runSharded(int ndev, const int* devices, int nframes, int (*fn)(void* user, int slot, int device, int first, int count), void* user, int bind);
MI355CV_API int  mi355cv_replicate(const void* src, size_t bytes, int ndev, const int* devices, void** out);
/* how the last mi355cv_replicate of this process moved its data: 0 = one hipMemcpy per device (a single device, MI355CV_REPLICATE=copy, or no usable librccl), 1 = one upload to the first device + an RCCL ncclBroadcast from
 * there over xGMI (the default for more than one device since round 6; MI355CV_REPLICATE=rccl also for one; librccl.so is resolved with dlopen; a device list RCCL cannot take -- a device twice, no library -- uses the copies) */
MI355CV_API int  mi355cv_replicateMode(void);
MI355CV_API const char* mi355cv_version(void);
/* Which host-resident images the hooks stage through HBM (device / managed pointers are always served).  0 = "auto" (the default: hooks whose CPU path is
 * multi-threaded and bandwidth-bound -- 8-bit Gaussian, colour conversions, threshold, pyrDown, morphology, integral, bilinear resize -- answer NOT_IMPLEMENTED for a plain host
 * image, because two PCIe crossings cost more than the reference's CPU path, and the caller's CPU path runs), 1 = "always" (every hook stages: for callers with no CPU path
 * to fall back to -- opencv_amd sets it when it loads the library), -1 = back to the MI355CV_HOST_POLICY environment variable.  Process-wide. */
MI355CV_API int mi355cv_setHostPolicy(int policy);
MI355CV_API int mi355cv_hostPolicy(void);          /* the policy in force: 0 auto, 1 always */
/* capacity bound of a served path by name (-1: no such key) -- "how far the GPU path was built", not the cases the reference itself has no engine for:
 * "sep_max_taps", "sep_max_taps_64f", "gauss8u_max_ksize", "gauss_float_max_ksize", "adaptive_gaussian_max_block", "adaptive_mean_max_block", "box_max_ksize",
 * "median8u_max_ksize", "bilateral_max_d", "orb_max_levels", "filter2d_dft_taps".  Needs no device.  A value above a bound is answered MI355CV_NOT_IMPLEMENTED by its hook. */
MI355CV_API int mi355cv_limit(const char* key);
MI355CV_API const char* mi355cv_lastError(void);
/* template instance + launch geometry of the dominant kernel the calling thread launched last (bench.py reports it beside the roofline) */
MI355CV_API const char* mi355cv_lastKernel(void);
/* MI355CV_TRACE=1: every entry point runs inside a roctx range named after it (roctxRangePushA / roctxRangePop, resolved with dlopen; rocprofv3 --marker-trace shows the
 * hooks around their kernels -- the role of CV_INSTRUMENT_REGION, core/private.hpp:794).  Returns 1 while ranges are emitted, 0 when MI355CV_TRACE is not set, -1 when no
 * roctx library could be loaded. */
MI355CV_API int mi355cv_traceState(void);
/* stream used for launches made by the calling thread on its current device (mi355cv_setDevice): exactly this hipStream_t (NULL = HIP's
 * null stream).  Until called -- or after mi355cv_resetStream() -- a library-owned per-thread, per-device stream is used. */
MI355CV_API int  mi355cv_setStream(void* hipStream);
MI355CV_API int  mi355cv_resetStream(void);
/* 1: calls on device-resident images return after enqueue (caller synchronises); 0 (default): synchronous */
MI355CV_API int  mi355cv_setAsync(int enable);
MI355CV_API int  mi355cv_synchronize(void);
/* number of times the named entry point ran its GPU path to completion in this process
 * (the analogue of the reference's CV_IMPL_ADD bookkeeping, core/private.hpp) */
MI355CV_API long long mi355cv_callCount(const char* entry);
/* the other side of the ledger: calls a hook DECLINED (answered MI355CV_NOT_IMPLEMENTED, so that the caller ran its own CPU path --
 * hal_replacement.hpp:1351-1357 makes that silent).  The HAL header (mi355cv_hal.hpp) and the Python mirror report every declined call
 * through mi355cv_noteDecline under the reference's hook name ("warpAffine", "sepFilter", ...); mi355cv_declineCount(hook) reads one
 * tally, mi355cv_declineCount(NULL) the total; MI355CV_PRINT_COUNTS=1 prints both ledgers at exit with the last reason per hook. */
MI355CV_API void      mi355cv_noteDecline(const char* hook);
MI355CV_API long long mi355cv_declineCount(const char* hook);
/* image bytes the hooks of this process have moved over PCIe so far (host-resident images staged into HBM + results staged back); stays
 * constant across calls on device-resident / managed images -- how a test shows that a pipeline ran in place in HBM */
MI355CV_API long long mi355cv_stagedBytes(void);
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
/* frame ingest / egress (the storage behind a cv::MatAllocator, core/mat.hpp:496-524): kind 0 = page-locked host memory
 * (a host frame is then staged by one DMA at PCIe rate), kind 1 = managed memory (the hooks run on it in place).  NULL without a device. */
MI355CV_API void* mi355cv_hostAlloc(size_t bytes, int kind);
MI355CV_API int   mi355cv_hostFree(void* p, int kind);

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
 * Implemented: depth 8U with zero margins = the Q8.8 fixed-point path of GaussianBlurFixedPoint (:720), bit-exact (sigma>0 kernels are
 * generated as getGaussianKernelBitExact does).  Declined: other depths (the reference then runs sepFilter2D, whose hooks serve it) and an
 * 8U submatrix with real margins (call site :813 -- there the CPU result is sepFilter2D with float taps, not the fixed-point one). */
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
 * geometry, frame f at src_data + f*src_frame_stride; one launch, grid-z = frame.  Host-resident batches: the chunked two-buffer pipeline (see the
 * batch section below). */
/* the same for any sigma (the Q8.8 taps cv::GaussianBlur computes for CV_8U, smooth.dispatch.cpp:658-724), device-resident frames only, kernel sizes odd and at most
 * mi355cv_limit("gauss8u_max_ksize") */
MI355CV_API int mi355cv_gaussianBlurBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes, int width, int height, int depth, int cn,
        size_t ksize_width, size_t ksize_height, double sigmaX, double sigmaY, int border_type);
MI355CV_API int mi355cv_gaussianBlurBinomialBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes,
        int width, int height, int depth, int cn, size_t ksize, int border_type);

/* --------------------------------------------------- a3/a4/a5: linear filters */

struct cvhalFilter2D;   /* opaque context, as in hal_replacement.hpp:87 */

/* replace hal_ni_filterInit / hal_ni_filter / hal_ni_filterFree (hal_replacement.hpp:109,125,131);
 * callers: replacementFilter2D filter.dispatch.cpp:1163-1185 (cv::filter2D).  Depths 8U/16U/16S/32F,
 * any kernel size up to 1024 taps, any anchor, borders CONSTANT(0)/REPLICATE/REFLECT/WRAP/REFLECT_101. */
MI355CV_API int mi355cv_filterInit(struct cvhalFilter2D** context, mi355cv_uchar* kernel_data, size_t kernel_step, int kernel_type,
        int kernel_width, int kernel_height, int max_width, int max_height, int src_type, int dst_type, int borderType,
        double delta, int anchor_x, int anchor_y, bool allowSubmatrix, bool allowInplace);
MI355CV_API int mi355cv_filter(struct cvhalFilter2D* context, mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data,
        size_t dst_step, int width, int height, int full_width, int full_height, int offset_x, int offset_y);
MI355CV_API int mi355cv_filterFree(struct cvhalFilter2D* context);
/* batched form over device-resident whole frames, with a context from mi355cv_filterInit */
MI355CV_API int mi355cv_filterBatch(struct cvhalFilter2D* context, const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes, int width, int height);

/* cv::cvtColor(COLOR_BGR2GRAY | RGB2GRAY | BGRA2GRAY | RGBA2GRAY) (color.cpp:192 -> hal::cvtBGRtoGray, color_rgb.dispatch.cpp:269) followed by
 * cv::filter2D (filter.dispatch.cpp:1521) on device-resident CV_8UC3 / CV_8UC4 frames in one pass: the gray image is never written (SURVEY
 * section 8d: 33.2 MB per 4K frame instead of 49.8 MB).  `context` from mi355cv_filterInit for CV_8UC1 -> CV_8UC1, 3x3 or 5x5, centred anchor;
 * width a multiple of 16 pixels; NOT_IMPLEMENTED otherwise (make the two calls).  Bit-identical to the two-call sequence. */
MI355CV_API int mi355cv_cvtBGRtoGrayFilterBatch(struct cvhalFilter2D* context, const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes, int width, int height, int scn, bool swapBlue);

/* replace hal_ni_sepFilterInit / hal_ni_sepFilter / hal_ni_sepFilterFree (hal_replacement.hpp:155,171,177);
 * callers: replacementSepFilter filter.dispatch.cpp:1362-1383 (cv::sepFilter2D, and through it Sobel/Scharr/
 * GaussianBlur on non-8U depths). */
MI355CV_API int mi355cv_sepFilterInit(struct cvhalFilter2D** context, int src_type, int dst_type, int kernel_type,
        mi355cv_uchar* kernelx_data, int kernelx_length, mi355cv_uchar* kernely_data, int kernely_length,
        int anchor_x, int anchor_y, double delta, int borderType);
MI355CV_API int mi355cv_sepFilter(struct cvhalFilter2D* context, mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data,
        size_t dst_step, int width, int height, int full_width, int full_height, int offset_x, int offset_y);
MI355CV_API int mi355cv_sepFilterFree(struct cvhalFilter2D* context);
/* what mi355cv_sepFilterInit decided (no reference counterpart; needs no device): info[8] = {mode (0 float, 1 CV_8U bit-exact x 2^8, 2 integer CV_8U -> CV_16S), symY,
 * nx, ny, anchor x, anchor y, delta as the integer modes add it, 1 if the destination is CV_64F}; kx / ky (room for mi355cv_limit("sep_max_taps") words each) receive
 * the taps as the kernels read them: float bits in mode 0, int32 otherwise.  tests/test_hostemu.py replays the LDS-ring kernel on the CPU with these. */
MI355CV_API int mi355cv_sepFilterDescribe(struct cvhalFilter2D* context, int* info, float* deltaF, unsigned* kx, unsigned* ky);

/* replaces hal_ni_sobel (hal_replacement.hpp:1197; caller deriv.cpp:456) and hal_ni_scharr (:1224; caller deriv.cpp:511) */
MI355CV_API int mi355cv_sobel(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right,
        int margin_bottom, int dx, int dy, int ksize, double scale, double delta, int border_type);
MI355CV_API int mi355cv_scharr(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right,
        int margin_bottom, int dx, int dy, double scale, double delta, int border_type);

/* replaces hal_ni_boxFilter (hal_replacement.hpp:1105; caller box_filter.dispatch.cpp:474) */
MI355CV_API int mi355cv_boxFilter(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right,
        int margin_bottom, size_t ksize_width, size_t ksize_height, int anchor_x, int anchor_y, bool normalize, int border_type);

/* --------------------------------------------------- a6: colour conversion */

/* replaces hal_ni_cvtBGRtoGray (hal_replacement.hpp:442); caller hal::cvtBGRtoGray color_rgb.dispatch.cpp:276.
 * depth 8U / 16U / 32F, scn 3|4.  8U/16U bit-exact (RGB2Gray<uchar> color_rgb.simd.hpp:660), 32F = the FMA chain
 * of RGB2Gray<float> :608. */
MI355CV_API int mi355cv_cvtBGRtoGray(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int scn, bool swapBlue);
/* replaces hal_ni_cvtGraytoBGR (hal_replacement.hpp:456) */
MI355CV_API int mi355cv_cvtGraytoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int dcn);
/* replaces hal_ni_cvtBGRtoBGR (hal_replacement.hpp:395): reorder / add / drop alpha */
MI355CV_API int mi355cv_cvtBGRtoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int scn, int dcn, bool swapBlue);
/* batched BGR->gray over device-resident frames (grid-z = frame) */
MI355CV_API int mi355cv_cvtBGRtoGrayBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes,
        int width, int height, int depth, int scn, int swapBlue);

/* --------------------------------------------------- a7/a8/a9: geometric transforms */

/* replaces hal_ni_resize (hal_replacement.hpp:257; caller hal::resize resize.cpp:3840).  INTER_NEAREST, INTER_LINEAR
 * (8U fixed point bit-exact; 16U/16S/32F float), INTER_AREA for integer ratios (resizeAreaFast_) and for upscaling. */
MI355CV_API int mi355cv_resize(int src_type, const mi355cv_uchar* src_data, size_t src_step, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, double inv_scale_x, double inv_scale_y,
        int interpolation);
/* replaces hal_ni_warpAffine (hal_replacement.hpp:275; caller imgwarp.cpp:2678).  M maps dst -> src (already inverted) */
MI355CV_API int mi355cv_warpAffine(int src_type, const mi355cv_uchar* src_data, size_t src_step, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, const double M[6], int interpolation,
        int borderType, const double borderValue[4]);
/* replaces hal_ni_warpPerspective (hal_replacement.hpp:316; caller imgwarp.cpp:3290) */
MI355CV_API int mi355cv_warpPerspective(int src_type, const mi355cv_uchar* src_data, size_t src_step, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, const double M[9], int interpolation,
        int borderType, const double borderValue[4]);
/* replaces hal_ni_remap32f (hal_replacement.hpp:371; caller imgwarp.cpp:1820) */
MI355CV_API int mi355cv_remap32f(int src_type, const mi355cv_uchar* src_data, size_t src_step, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, float* mapx, size_t mapx_step,
        float* mapy, size_t mapy_step, int interpolation, int border_type, const double border_value[4]);
/* cv::remap for every map representation it accepts (imgwarp.cpp:1718-1921; the HAL only has remap32f): (CV_32FC1, CV_32FC1), CV_32FC2, the
 * fixed-point form of cv::convertMaps -- CV_16SC2 + CV_16UC1 / CV_16SC1 (bilinear, nearest), CV_16SC2 alone (nearest).  map*_type: cv type codes. */
MI355CV_API int mi355cv_remap(int src_type, const mi355cv_uchar* src_data, size_t src_step, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, const void* map1, size_t map1_step, int map1_type,
        const void* map2, size_t map2_step, int map2_type, int interpolation, int border_type, const double border_value[4]);
/* cv::convertMaps (imgwarp.cpp:1925): float maps <-> CV_16SC2 (+ CV_16UC1), bit-exact with the reference's rounding */
MI355CV_API int mi355cv_convertMaps(const void* map1, size_t map1_step, int map1_type, const void* map2, size_t map2_step, int map2_type,
        void* dstmap1, size_t dstmap1_step, int dstmap1_type, void* dstmap2, size_t dstmap2_step, int width, int height, int nninterpolate);
/* cv::warpPolar (imgwarp.cpp:3731-3845), both directions: the map is evaluated inside the sampling kernel -- forward from the per-row cos / sin and
 * per-column radii, WARP_INVERSE_MAP from the reference's float approximations of cartToPolar and log (source with one wrapped row above and below) */
MI355CV_API int mi355cv_warpPolar(int src_type, const mi355cv_uchar* src_data, size_t src_step, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, float center_x, float center_y, double maxRadius, int flags);

/* --------------------------------------------------- a10/a11/a12: corners and pyramids */

/* replace hal_ni_pyrdown (hal_replacement.hpp:1244) and hal_ni_pyrdown_offset (:1268); caller cv::pyrDown pyramids.cpp:1371,1377.
 * depth 8U/16U/16S/32F, cn 1..4, every border cv::pyrDown accepts. */
MI355CV_API int mi355cv_pyrdown(const mi355cv_uchar* src_data, size_t src_step, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, int depth, int cn, int border_type);
MI355CV_API int mi355cv_pyrdown_offset(const mi355cv_uchar* src_data, size_t src_step, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, int depth, int cn,
        int margin_left, int margin_top, int margin_right, int margin_bottom, int border_type);
MI355CV_API int mi355cv_pyrdownBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int dst_width, int dst_height, int nframes,
        int depth, int cn, int border_type);
/* cv::buildPyramid (pyramids.cpp:1616-1643) has no HAL hook: dst_data[i] / dst_step[i] receive level i+1. */
MI355CV_API int mi355cv_buildPyramid(const mi355cv_uchar* src_data, size_t src_step, int width, int height, int depth, int cn,
        mi355cv_uchar** dst_data, const size_t* dst_step, int maxlevel, int border_type);
/* the same over a batch of frames, all levels of all frames enqueued by one call: dst_data[l-1] / dst_step[l-1] / dst_frame_stride[l-1] describe
 * level l (frame 0's pointer, row pitch, bytes between frames) of pre-allocated arrays.  Frames and levels all in HBM, or all in host memory */
MI355CV_API int mi355cv_buildPyramidBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, int width, int height, int depth, int cn,
        mi355cv_uchar* const* dst_data, const size_t* dst_step, const size_t* dst_frame_stride, int maxlevel, int nframes, int border_type);

/* cv::cornerHarris (corner.cpp:634) / cv::cornerMinEigenVal (:604) have no HAL hook: fused entry points with the cv::
 * argument list.  src_type CV_8UC1 or CV_32FC1, dst CV_32FC1. */
MI355CV_API int mi355cv_cornerHarris(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int src_type, int blockSize, int ksize, double k, int borderType);
MI355CV_API int mi355cv_cornerMinEigenVal(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int src_type, int blockSize, int ksize, int borderType);
MI355CV_API int mi355cv_cornerHarrisBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes, int width, int height, int src_type,
        int blockSize, int ksize, double k, int borderType);
/* cv::goodFeaturesToTrack (featureselect.cpp:382): returns the number of corners written to `corners` (x,y pairs),
 * -1 if the arguments are unsupported, -2 on a device failure.  `quality` and `mask_data` may be NULL. */
MI355CV_API int mi355cv_goodFeaturesToTrack(const mi355cv_uchar* src_data, size_t src_step, int width, int height, int src_type,
        float* corners, float* quality, int maxCorners, double qualityLevel, double minDistance,
        const mi355cv_uchar* mask_data, size_t mask_step, int blockSize, int gradientSize, int useHarrisDetector, double harrisK);

/* --------------------------------------------------- f1/f4: YUV family, CV_8U */

/* replace hal_ni_cvtBGRtoYUV (hal_replacement.hpp:500), hal_ni_cvtYUVtoBGR (:533), hal_ni_cvtTwoPlaneYUVtoBGR (:664) and
 * hal_ni_cvtTwoPlaneYUVtoBGREx (:701); callers color_yuv.dispatch.cpp:33, :86, :166, :144.  depth CV_8U only (other depths answer
 * NOT_IMPLEMENTED).  The two-plane decoders take NV12 (uIdx 0) / NV21 (uIdx 1), even dst_width and dst_height. */
MI355CV_API int mi355cv_cvtBGRtoYUV(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int scn, bool swapBlue, bool isCbCr);
MI355CV_API int mi355cv_cvtYUVtoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int dcn, bool swapBlue, bool isCbCr);
MI355CV_API int mi355cv_cvtTwoPlaneYUVtoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int dst_width, int dst_height, int dcn, bool swapBlue, int uIdx);
/* hal_ni_cvtBGRtoHSV (hal_replacement.hpp:596; caller color_hsv.dispatch.cpp:65): CV_8U, isHSV only (HLS and CV_32F decline) */
MI355CV_API int mi355cv_cvtBGRtoHSV(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int scn, bool swapBlue, bool isFullRange, bool isHSV);
/* hal_ni_cvtThreePlaneYUVtoBGR (hal_replacement.hpp:763): I420 / IYUV (uIdx 0) and YV12 (uIdx 1) in one array */
MI355CV_API int mi355cv_cvtThreePlaneYUVtoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int dst_width, int dst_height, int dcn, bool swapBlue, int uIdx);
MI355CV_API int mi355cv_cvtTwoPlaneYUVtoBGREx(const mi355cv_uchar* y_data, size_t y_step, const mi355cv_uchar* uv_data, size_t uv_step,
        mi355cv_uchar* dst_data, size_t dst_step, int dst_width, int dst_height, int dcn, bool swapBlue, int uIdx);

/* --------------------------------------------------- f1: fixed-level threshold */

/* replaces hal_ni_threshold (hal_replacement.hpp:1058; caller ThresholdRunner thresh.cpp:1365, once per row stripe).
 * thresh / maxValue as cv::threshold hands them over (already floor/round/saturated for integer depths);
 * depth CV_8U / CV_16U / CV_16S / CV_32F, thresholdType THRESH_BINARY .. THRESH_TOZERO_INV (0..4). */
MI355CV_API int mi355cv_threshold(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int cn, double thresh, double maxValue, int thresholdType);

/* replaces hal_ni_imageMoments (hal_replacement.hpp:1309; caller cv::moments moments.cpp:578): the ten spatial moments m00, m10, m01, m20, m11, m02, m30, m21,
 * m12, m03 of a single-channel CV_8U / CV_16U / CV_16S image (or of its non-zero mask, `binary`), bit-identical: exact integer tile moments on the GPU,
 * the reference's double accumulation over the tiles on the host. */
MI355CV_API int mi355cv_imageMoments(const mi355cv_uchar* src_data, size_t src_step, int src_type, int width, int height, bool binary, double m[10]);

/* replaces hal_ni_bilateralFilter (hal_replacement.hpp:1016; caller cv::bilateralFilter bilateral_filter.dispatch.cpp:418): CV_8UC1 / CV_8UC3, radius <= 16,
 * the float sums in the three forms of the reference's AVX2 build (vector body / 4-group tail / scalar rest), bit-exact.  The hook carries no margins: an
 * image with padded rows and no BORDER_ISOLATED is declined (the reference pads a submatrix with its parent's pixels). */
MI355CV_API int mi355cv_bilateralFilter(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
                                        int depth, int cn, int d, double sigma_color, double sigma_space, int border_type);

/* replaces hal_ni_adaptiveThreshold (hal_replacement.hpp:1038; caller cv::adaptiveThreshold thresh.cpp:1711): CV_8UC1,
 * adaptiveMethod ADAPTIVE_THRESH_MEAN_C (0) with blockSize 3..15, thresholdType THRESH_BINARY (0) / THRESH_BINARY_INV (1). */
MI355CV_API int mi355cv_adaptiveThreshold(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, double maxValue, int adaptiveMethod, int thresholdType, int blockSize, double C);

/* --------------------------------------------------- f1 / f4: remaining integer colour conversions (csrc/color_misc.hip) */

/* replaces hal_ni_cvtBGRtoTwoPlaneYUV (hal_replacement.hpp:743; caller color_yuv.dispatch.cpp:238): BGR/RGB(A) CV_8U -> NV12 (uIdx 1) / NV21 (uIdx 2) */
MI355CV_API int mi355cv_cvtBGRtoTwoPlaneYUV(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* y_data, size_t y_step,
        mi355cv_uchar* uv_data, size_t uv_step, int width, int height, int scn, bool swapBlue, int uIdx);
/* replaces hal_ni_cvtBGRtoThreePlaneYUV (:797; caller color_yuv.dispatch.cpp:222): -> I420 / IYUV (uIdx 1) or YV12 (uIdx 2), one (3/2 height) x width array */
MI355CV_API int mi355cv_cvtBGRtoThreePlaneYUV(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int scn, bool swapBlue, int uIdx);
/* replaces hal_ni_cvtOnePlaneYUVtoBGR (:833; caller color_yuv.dispatch.cpp:260): YUY2 / YVYU / UYVY (CV_8UC2) -> BGR/RGB(A) */
MI355CV_API int mi355cv_cvtOnePlaneYUVtoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int dcn, bool swapBlue, int uIdx, int ycn);
/* replaces hal_ni_cvtOnePlaneBGRtoYUV (:866; caller color_yuv.dispatch.cpp:281) */
MI355CV_API int mi355cv_cvtOnePlaneBGRtoYUV(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int scn, bool swapBlue, int uIdx, int ycn);
/* replace hal_ni_cvtBGRtoXYZ (:564; caller color_lab.cpp:4134) and hal_ni_cvtXYZtoBGR (:579): CV_8U, CV_16U, CV_32F (the body / tail association of the SSE-baseline build, see csrc/color_misc.hip) */
MI355CV_API int mi355cv_cvtBGRtoXYZ(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int depth, int scn, bool swapBlue);
MI355CV_API int mi355cv_cvtXYZtoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int depth, int dcn, bool swapBlue);
/* replace hal_ni_cvtBGRtoLab (hal_replacement.hpp:535; caller hal::cvtBGRtoLab color_lab.cpp:4230) and hal_ni_cvtLabtoBGR (:550; caller :4327): CV_8U
 * L*a*b* from / to sRGB (COLOR_BGR2Lab ...) or linear RGB (COLOR_LBGR2Lab ...) -- the reference's bit-exact integer paths RGB2Lab_b (:1573) and
 * Lab2RGBinteger (:2399) -- and CV_8U L*u*v* (isLab == false): sRGB -> Luv through the reference's 33^3 interpolation table (RGB2Luvinterpolate :3276),
 * Luv -> sRGB / linear RGB by Luv2RGBinteger (:3556) -- and CV_32F L*a*b* (RGB2Lab_f :1895, Lab2RGBfloat :2169, vector bodies and scalar row tails as the
 * reference has them), CV_32F L*u*v* and CV_8U L*u*v* from linear RGB (RGB2Luvfloat :2868, Luv2RGBfloat :3057): every (depth, isLab, srgb) case the
 * reference's pair of hooks receives (csrc/color_lab.hip) */
MI355CV_API int mi355cv_cvtBGRtoLab(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int depth, int scn, bool swapBlue, bool isLab, bool srgb);
MI355CV_API int mi355cv_cvtLabtoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int depth, int dcn, bool swapBlue, bool isLab, bool srgb);
/* diagnostics: the host-built tables behind the two hooks (0 gamma 256 x u16, 1 cbrt 3072 x u16, 2 invGamma 4096 x u16, 3 (y | ify << 16) 256 x u32,
 * 4 RGB -> Luv grid 33^3 x 4 x i16, 5 / 6 LuToUp / LvToVp 65536 x i32, 7 RGB -> Lab grid, 8 / 9 the cube-root / inverse-gamma splines 4096 x f32);
 * returns the entry count, needs no GPU */
MI355CV_API int mi355cv_labTable(int which, void* out);
/* replace hal_ni_cvtBGRtoBGR5x5 (:411), cvtBGR5x5toBGR (:427), cvtBGR5x5toGray (:470), cvtGraytoBGR5x5 (:484); callers color_rgb.dispatch.cpp */
MI355CV_API int mi355cv_cvtBGRtoBGR5x5(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int scn, bool swapBlue, int greenBits);
MI355CV_API int mi355cv_cvtBGR5x5toBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int dcn, bool swapBlue, int greenBits);
MI355CV_API int mi355cv_cvtBGR5x5toGray(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int greenBits);
MI355CV_API int mi355cv_cvtGraytoBGR5x5(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int greenBits);
/* replace hal_ni_cvtRGBAtoMultipliedRGBA (:894) and hal_ni_cvtMultipliedRGBAtoRGBA (:907) */
MI355CV_API int mi355cv_cvtRGBAtoMultipliedRGBA(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height);
MI355CV_API int mi355cv_cvtMultipliedRGBAtoRGBA(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height);

/* replaces hal_ni_cvtHSVtoBGR (hal_replacement.hpp:613; caller color_hsv.dispatch.cpp:95): CV_8U, HSV (HLS and CV_32F decline).  The reference's
 * 8-bit result depends on its vector width (truncation in the vector loop, rounding in the scalar tail); this follows the 8-lane AVX2 build,
 * the widest `color_hsv` is dispatched for (modules/imgproc/CMakeLists.txt: SSE2 SSE4_1 AVX2). */
MI355CV_API int mi355cv_cvtHSVtoBGR(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int depth, int dcn, bool swapBlue, bool isFullRange, bool isHSV);

/* --------------------------------------------------- f1: histogram-driven point operations (csrc/hist.hip) */

/* replaces hal_ni_equalize_hist (hal_replacement.hpp:1120; caller histogram.cpp:3455): CV_8UC1 */
MI355CV_API int mi355cv_equalize_hist(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height);
/* replaces hal_ni_threshold_otsu (:1077; caller thresh.cpp:1568): CV_8UC1 / CV_16UC1; *thresh receives the Otsu level */
MI355CV_API int mi355cv_threshold_otsu(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int depth, double maxValue, int thresholdType, double* thresh);

/* --------------------------------------------------- f3: sparse pyramidal Lucas-Kanade (modules/video) */

/* replaces hal_ni_ScharrDeriv (modules/video/src/hal_replacement.hpp:84; caller calcScharrDeriv lkpyramid.cpp:67): CV_8U cn 1..4 ->
 * interleaved (dI/dx, dI/dy) CV_16S, 2*cn channels */
MI355CV_API int mi355cv_ScharrDeriv(const mi355cv_uchar* src_data, size_t src_step, short* dst_data, size_t dst_step, int width, int height, int cn);
/* replaces hal_ni_LKOpticalFlowLevel (modules/video/src/hal_replacement.hpp:54; caller LKTrackerInvoker::operator() lkpyramid.cpp:233): one
 * pyramid level.  The three images must be readable one window beyond every edge (the padded pyramids of buildOpticalFlowPyramid).
 * status is non-NULL at level 0 only; termination_epsilon is the squared threshold the caller prepares. */
MI355CV_API int mi355cv_LKOpticalFlowLevel(const mi355cv_uchar* prev_data, size_t prev_data_step, const short* prev_deriv_data, size_t prev_deriv_step,
        const mi355cv_uchar* next_data, size_t next_step, int width, int height, int cn,
        const float* prev_points, float* next_points, size_t point_count, mi355cv_uchar* status, float* err,
        const int win_width, const int win_height, int termination_count, double termination_epsilon,
        bool get_min_eigen_vals, float min_eigen_vals_threshold);
/* cv::calcOpticalFlowPyrLK (lkpyramid.cpp:1432) in one call: frames staged once, padded pyramids, derivatives and all tracker levels on the
 * device.  cv:: argument meaning (criteria = TermCriteria type / maxCount / epsilon, flags = OPTFLOW_*); err may be NULL.  Results are
 * bit-identical to the reference's.  Returns 0, 1 (declined, nothing written) or < 0. */
MI355CV_API int mi355cv_calcOpticalFlowPyrLK(const mi355cv_uchar* prev_data, size_t prev_step, const mi355cv_uchar* next_data, size_t next_step,
        int width, int height, int cn, const float* prev_points, float* next_points, int point_count, mi355cv_uchar* status, float* err,
        int win_width, int win_height, int max_level, int criteria_type, int criteria_max_count, double criteria_epsilon,
        int flags, double min_eig_threshold);
/* cv::copyMakeBorder (core/src/copy.cpp:1183; no HAL hook) for device-resident images: pixels of elem_size bytes, BORDER_CONSTANT = zeros;
 * src may be the interior of dst (then only the frame is written) -- what buildOpticalFlowPyramid does at lkpyramid.cpp:804 */
MI355CV_API int mi355cv_copyMakeBorder(const mi355cv_uchar* src_data, size_t src_step, int width, int height, mi355cv_uchar* dst_data, size_t dst_step,
        int top, int bottom, int left, int right, int elem_size, int border_type);

/* --------------------------------------------------- f1: Canny */

/* replaces hal_ni_canny (hal_replacement.hpp:1291; caller cv::Canny canny.cpp:864).  CV_8U, 1..4 channels, ksize 3 or 5; thresholds as
 * cv::Canny passes them to the hook (before the L2 squaring and the floor).  dst CV_8UC1, 255 on edges. */
MI355CV_API int mi355cv_canny(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height,
        int cn, double lowThreshold, double highThreshold, int ksize, bool L2gradient);

/* --------------------------------------------------- f1: erode / dilate */

/* replace hal_ni_morphInit / hal_ni_morph / hal_ni_morphFree (hal_replacement.hpp:207-233; caller halMorph
 * morph.dispatch.cpp:190-220).  operation 0 = MORPH_ERODE, 1 = MORPH_DILATE; kernel = CV_8UC1 mask; iterations must be 1 (the
 * reference folds iterated rectangles before the hook); borderValue all DBL_MAX = morphologyDefaultBorderValue(). */
MI355CV_API int mi355cv_morphInit(struct cvhalFilter2D** context, int operation, int src_type, int dst_type, int max_width, int max_height,
        int kernel_type, mi355cv_uchar* kernel_data, size_t kernel_step, int kernel_width, int kernel_height, int anchor_x, int anchor_y,
        int borderType, const double borderValue[4], int iterations, bool allowSubmatrix, bool allowInplace);
MI355CV_API int mi355cv_morph(struct cvhalFilter2D* context, mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int src_full_width, int src_full_height, int src_roi_x, int src_roi_y,
        int dst_full_width, int dst_full_height, int dst_roi_x, int dst_roi_y);
MI355CV_API int mi355cv_morphFree(struct cvhalFilter2D* context);

/* --------------------------------------------------- f1: median filter */

/* replaces hal_ni_medianBlur (hal_replacement.hpp:995; caller cv::medianBlur median_blur.dispatch.cpp:300).  CV_8U, ksize 3 or 5,
 * 1/3/4 channels; everything else answers NOT_IMPLEMENTED (CPU fallback). */
MI355CV_API int mi355cv_medianBlur(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int cn, int ksize);

/* --------------------------------------------------- a13: template matching */

/* cv::matchTemplate (templmatch.cpp:1158) has no HAL hook.  type CV_8UC1..C4 / CV_32FC1..C4, result CV_32FC1
 * (img_width - templ_width + 1) x (img_height - templ_height + 1), method = cv::TemplateMatchModes 0..5. */
MI355CV_API int mi355cv_matchTemplate(const mi355cv_uchar* img_data, size_t img_step, int img_width, int img_height,
        const mi355cv_uchar* templ_data, size_t templ_step, int templ_width, int templ_height, int type,
        mi355cv_uchar* result_data, size_t result_step, int method);
/* the same with a mask (matchTemplateMask, templmatch.cpp:762-904): mask of the template's size, CV_8U (non-zero counts as 1) or CV_32F (weights), one channel
 * or the template's channel count */
MI355CV_API int mi355cv_matchTemplateMask(const mi355cv_uchar* img_data, size_t img_step, int img_width, int img_height,
        const mi355cv_uchar* templ_data, size_t templ_step, int templ_width, int templ_height, int type,
        const mi355cv_uchar* mask_data, size_t mask_step, int mask_type, mi355cv_uchar* result_data, size_t result_step, int method);
/* frames x one shared template (frames shard, the template is replicated) */
MI355CV_API int mi355cv_matchTemplateBatch(const mi355cv_uchar* img_data, size_t img_step, size_t img_frame_stride, int nframes,
        int img_width, int img_height, const mi355cv_uchar* templ_data, size_t templ_step, int templ_width, int templ_height,
        int type, mi355cv_uchar* result_data, size_t result_step, size_t result_frame_stride, int method);
/* replaces hal_ni_integral (hal_replacement.hpp:977; caller sumpixels.dispatch.cpp:415): every (depth, sdepth, sqdepth) row of the reference's table
 * (sumpixels.dispatch.cpp:383-406), sqsum_data / tilted_data may be NULL.  Integer-valued sums are exact; float sums, float / int squared sums and tilted sums are
 * accumulated in the reference's order (bit for bit).  Declined: CV_8U -> CV_32F sums past 2^24 when neither sqsum nor tilted is asked for (the reference's
 * result then depends on the CPU's vector width). */
MI355CV_API int mi355cv_integral(int depth, int sdepth, int sqdepth, const mi355cv_uchar* src_data, size_t src_step,
        mi355cv_uchar* sum_data, size_t sum_step, mi355cv_uchar* sqsum_data, size_t sqsum_step,
        mi355cv_uchar* tilted_data, size_t tilted_step, int width, int height, int cn);

/* cv::integral over `nframes` device-resident CV_8UC1 frames: sums in CV_32S or CV_64F (sdepth), optional CV_64F squared sums (sqsum_data may be
 * NULL), each (height+1) x (width+1); strides in bytes; one set of launches for the whole batch */
MI355CV_API int mi355cv_integralBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, mi355cv_uchar* sum_data, size_t sum_step,
        size_t sum_frame_stride, mi355cv_uchar* sqsum_data, size_t sqsum_step, size_t sqsum_frame_stride, int nframes, int width, int height, int sdepth);

/* --------------------------------------------------- batches of device-resident frames (frames are the unit that shards)
 * The frame-batched forms of the hooks above: `nframes` whole images of identical geometry, `*_frame_stride` bytes apart, borders per frame.
 * One launch where the kernel takes a frame index (the rolling filters, nearest / bilinear / area-fast resize, the warps, threshold on
 * back-to-back frames), otherwise the per-frame kernels enqueued by one call.  Both ends in HBM -- or both in host memory (pageable or page-locked): the
 * batch then crosses PCIe in chunks of <= 16 frames / 64 MB through two sets of device buffers, upload of chunk i+1 overlapped with the kernels and the
 * download of chunk i; mi355cv_buildPyramidBatch (one output per level) and mi355cv_matchTemplateBatch included. */
MI355CV_API int mi355cv_sobelBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, mi355cv_uchar* dst_data, size_t dst_step,
        size_t dst_frame_stride, int nframes, int width, int height, int src_depth, int dst_depth, int cn, int dx, int dy, int ksize, double scale, double delta,
        int border_type);
MI355CV_API int mi355cv_sepFilterBatch(struct cvhalFilter2D* context, const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, mi355cv_uchar* dst_data,
        size_t dst_step, size_t dst_frame_stride, int nframes, int width, int height);
MI355CV_API int mi355cv_boxFilterBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, mi355cv_uchar* dst_data, size_t dst_step,
        size_t dst_frame_stride, int nframes, int width, int height, int src_depth, int dst_depth, int cn, size_t ksize_width, size_t ksize_height,
        int anchor_x, int anchor_y, bool normalize, int border_type);
MI355CV_API int mi355cv_thresholdBatch(const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, mi355cv_uchar* dst_data, size_t dst_step,
        size_t dst_frame_stride, int nframes, int width, int height, int depth, int cn, double thresh, double maxValue, int thresholdType);
MI355CV_API int mi355cv_resizeBatch(int src_type, const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int dst_width, int dst_height, int nframes, double inv_scale_x, double inv_scale_y,
        int interpolation);
MI355CV_API int mi355cv_warpAffineBatch(int src_type, const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int dst_width, int dst_height, int nframes, const double M[6], int interpolation,
        int borderType, const double borderValue[4]);
MI355CV_API int mi355cv_warpPerspectiveBatch(int src_type, const mi355cv_uchar* src_data, size_t src_step, size_t src_frame_stride, int src_width, int src_height,
        mi355cv_uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int dst_width, int dst_height, int nframes, const double M[9], int interpolation,
        int borderType, const double borderValue[4]);

/* --------------------------------------------------- f3: features2d FAST corner detector (csrc/fast.hip) */
/* replace hal_ni_FAST_dense / hal_ni_FAST_NMS (modules/features2d/src/hal_replacement.hpp:75, :87; caller hal_FAST fast.cpp:438-493, which is reached
 * for threshold <= 20): dense score = largest t + 1 for which the pixel is a 9-of-16 corner at threshold t (0 in the 3-pixel frame); the suppression
 * keeps scores strictly greater than their 8 neighbours.  `type`: cv::FastFeatureDetector::DetectorType, TYPE_9_16 (2) only. */
MI355CV_API int mi355cv_FAST_dense(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height, int type);
MI355CV_API int mi355cv_FAST_NMS(const mi355cv_uchar* src_data, size_t src_step, mi355cv_uchar* dst_data, size_t dst_step, int width, int height);
/* cv::FAST (fast.cpp:496) as one call, any threshold: keypoints as (x, y, response) float triples in the reference's (raster) order, at most
 * `capacity` written.  Returns the number found (>= 0; call again with a larger array if it exceeds capacity), -1 unsupported, -2 device failure. */
MI355CV_API int mi355cv_FAST(const mi355cv_uchar* src_data, size_t src_step, int width, int height, int threshold, int nonmax_suppression, int type,
        float* keypoints_xyr, int capacity);

/* --------------------------------------------------- f3: features2d ORB (csrc/orb.hip) */
/* cv::KeyPoint (core/types.hpp:777) and the parameters of cv::ORB (features2d.hpp:449-510; scoreType: ORB::HARRIS_SCORE 0, FAST_SCORE 1).  scaleFactor is the
 * double the reference keeps: ORB::create takes a float (pass (double)(float)f for it), setScaleFactor a double */
typedef struct mi355cv_KeyPoint { float x, y, size, angle, response; int octave, class_id; } mi355cv_KeyPoint;
typedef struct mi355cv_OrbParams { int nfeatures; double scaleFactor; int nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize, fastThreshold; } mi355cv_OrbParams;
/* cv::ORB::detectAndCompute (modules/features2d/src/orb.cpp:1012; the reference has no HAL hook for it) on a CV_8UC1 image in host or device
 * memory: pyramid, FAST, Harris responses, orientation, smoothing and the rBRIEF descriptors on the GPU; the two culls (KeyPointsFilter::retainBest)
 * on the host with the C++ library's nth_element, so keypoints leave in the reference's order.  `mask`: optional CV_8UC1 image of the same size.
 * use_provided_keypoints != 0: describe the nkeypoints_in keypoints passed in (Feature2D::compute).  `descriptors`: rows of 32 bytes, or NULL to
 * detect only.  At most `capacity` keypoints / rows are written.  Returns the keypoint count (>= 0; call again with larger arrays if it exceeds
 * capacity -- nothing was written then), -1 arguments not served, -2 device failure. */
MI355CV_API int mi355cv_ORB_detectAndCompute(const mi355cv_uchar* image, size_t step, int width, int height, const mi355cv_uchar* mask, size_t mask_step,
        const mi355cv_OrbParams* params, int use_provided_keypoints, mi355cv_KeyPoint* keypoints, int nkeypoints_in, int capacity,
        mi355cv_uchar* descriptors, size_t descriptors_step);

#ifdef __cplusplus
}
#endif
#endif /* MI355CV_H */
