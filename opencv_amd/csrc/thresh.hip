// thresh.hip -- SURVEY.md §8 f1: cv_hal_threshold (hal_replacement.hpp:1058; caller ThresholdRunner thresh.cpp:1365, once per
// row stripe of cv::threshold).  thresh / maxval arrive preprocessed by cv::threshold (floor / round / saturate for the
// integer depths, thresh.cpp:1583-1680); per element dst = f(src > thresh) for the five fixed-level types
// (thresh_8u :112, thresh_16u :300, thresh_16s :478, thresh_32f :652).  HBM-bound: 2 * elemSize bytes per element.
#include "rt.h"
#include <vector>
#include <cmath>

using namespace mi355;

namespace {

enum { D8U = MI355CV_8U, D16U = MI355CV_16U, D16S = MI355CV_16S, D32F = MI355CV_32F, D64F = MI355CV_64F };

template <typename T, int TYPE> __device__ __forceinline__ T threshOne(T v, T t, T m)
{
    const bool gt = v > t;
    if (TYPE == 0) return gt ? m : (T)0;
    if (TYPE == 1) return gt ? (T)0 : m;
    if (TYPE == 2) return gt ? t : v;
    if (TYPE == 3) return gt ? v : (T)0;
    return gt ? (T)0 : v;
}

// four 8-bit pixels of a dword at once: G = 0xff in every byte whose pixel is > t (unsigned compare on the two 16-bit planes: saturating subtract,
// min with 1, times 0xff), then the five types are one bit-select each: dst = (G & A) | (~G & B).  The three packed 16-bit steps are written as
// instructions: expressed with vector builtins (min(sub_sat(e, t), 1) * 255) this LLVM folds the whole expression to "all ones" and drops the load.
__device__ __forceinline__ uint32_t gtMask2(uint32_t plane, uint32_t t16, uint32_t one2, uint32_t ff2)
{
    uint32_t d;
    asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3\n\tv_pk_mul_lo_u16 %0, %0, %4" : "=&v"(d) : "v"(plane), "v"(t16), "v"(one2), "v"(ff2));
    return d;                                              // 0x00ff in each 16-bit lane whose value is > t
}
template <int TYPE> __device__ __forceinline__ uint32_t thresh4(uint32_t v, uint32_t t16 /* t in both 16-bit lanes */, uint32_t m4, uint32_t t4)
{
    const uint32_t one2 = 0x00010001u, ff2 = 0x00ff00ffu;
    const uint32_t G = gtMask2(v & 0x00ff00ffu, t16, one2, ff2) | (gtMask2((v >> 8) & 0x00ff00ffu, t16, one2, ff2) << 8);
    if (TYPE == 0) return G & m4;
    if (TYPE == 1) return ~G & m4;
    if (TYPE == 2) return (G & t4) | (~G & v);
    if (TYPE == 3) return G & v;
    return ~G & v;
}

// a thread owns 16 bytes of a row when the geometry allows (one dwordx4 load / non-temporal store), single elements otherwise; the threshold type is
// a template parameter (a run-time switch per element made this a branch-bound kernel: 56 % of HBM where a copy reaches 79 %)
template <typename T, int TYPE>
__global__ __launch_bounds__(256) void k_threshold(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
                                                   int n /* elements per row */, int h, T t, T m, int vec)
{
    constexpr int EPV = 16 / sizeof(T);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const T* s = reinterpret_cast<const T*>(src + (size_t)y * sstep);
    T* d = reinterpret_cast<T*>(dst + (size_t)y * dstep);
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    if (vec) {
        if (i * EPV >= n) return;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        union { u32x4 q; T e[EPV]; } a;
        a.q = *reinterpret_cast<const u32x4*>(s + (size_t)i * EPV);
        if constexpr (sizeof(T) == 1) {
            const uint32_t t16 = (uint32_t)t * 0x00010001u, m4 = (uint32_t)m * 0x01010101u, t4 = (uint32_t)t * 0x01010101u;
#pragma unroll
            for (int k = 0; k < 4; k++) a.q[k] = thresh4<TYPE>(a.q[k], t16, m4, t4);
        } else {
#pragma unroll
            for (int k = 0; k < EPV; k++) a.e[k] = threshOne<T, TYPE>(a.e[k], t, m);
        }
        __builtin_nontemporal_store(a.q, reinterpret_cast<u32x4*>(d + (size_t)i * EPV));
    } else {
#pragma unroll 4
        for (int k = 0; k < EPV; k++) { const int x = i + k * (gridDim.x * 64); if (x < n) d[x] = threshOne<T, TYPE>(s[x], t, m); }
    }
}

// dst = (src - mean > -idelta) ? maxval : 0  (THRESH_BINARY)  /  (src - mean <= -idelta) ? maxval : 0  (THRESH_BINARY_INV): the 768-entry
// table of cv::adaptiveThreshold (thresh.cpp:1736-1745) evaluated directly
__global__ __launch_bounds__(256) void k_adaptive(const uchar* __restrict__ src, size_t sstep, const uchar* __restrict__ mean, size_t mstep,
                                                  uchar* __restrict__ dst, size_t dstep, int W, int H, int idelta, int maxval, int inv)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (x + k < W) {
            const int v = (int)src[(size_t)y * sstep + x + k] - (int)mean[(size_t)y * mstep + x + k];
            const bool on = inv ? v <= -idelta : v > -idelta;
            dst[(size_t)y * dstep + x + k] = (uchar)(on ? maxval : 0);
        }
    }
}

// ADAPTIVE_THRESH_GAUSSIAN_C: the source as float (src.convertTo(CV_32F), thresh.cpp:1722) ...
__global__ __launch_bounds__(256) void k_u8_to_f32(const uchar* __restrict__ src, size_t sstep, float* __restrict__ dst, size_t dstepF, int W, int H)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
#pragma unroll
    for (int k = 0; k < 4; k++) if (x + k < W) dst[(size_t)y * dstepF + x + k] = (float)src[(size_t)y * sstep + x + k];
}

// ... and the comparison against the blurred float image brought back to 8 bits (meanfloat.convertTo(mean, CV_8U): cvRound + saturate)
__global__ __launch_bounds__(256) void k_adaptive_f(const uchar* __restrict__ src, size_t sstep, const float* __restrict__ mean, size_t mstepF,
                                                    uchar* __restrict__ dst, size_t dstep, int W, int H, int idelta, int maxval, int inv)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (x + k < W) {
            const float mf = __builtin_rintf(mean[(size_t)y * mstepF + x + k]);
            const int m = (int)fminf(fmaxf(mf, 0.f), 255.f);
            const int v = (int)src[(size_t)y * sstep + x + k] - m;
            const bool on = inv ? v <= -idelta : v > -idelta;
            dst[(size_t)y * dstep + x + k] = (uchar)(on ? maxval : 0);
        }
    }
}

} // namespace

extern "C" MI355CV_API int mi355cv_boxFilter(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
        int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right, int margin_bottom,
        size_t ksize_width, size_t ksize_height, int anchor_x, int anchor_y, bool normalize, int border_type);

struct cvhalFilter2D;
extern "C" MI355CV_API int mi355cv_sepFilterInit(cvhalFilter2D** context, int src_type, int dst_type, int kernel_type, uchar* kernelx_data, int kernelx_length,
                                                 uchar* kernely_data, int kernely_length, int anchor_x, int anchor_y, double delta, int borderType);
extern "C" MI355CV_API int mi355cv_sepFilter(cvhalFilter2D* context, uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                             int full_width, int full_height, int offset_x, int offset_y);
extern "C" MI355CV_API int mi355cv_sepFilterFree(cvhalFilter2D* context);
extern "C" MI355CV_API int mi355cv_getGaussianKernel(int n, double sigma, double* taps);

// replaces hal_ni_adaptiveThreshold (hal_replacement.hpp:1038; caller cv::adaptiveThreshold thresh.cpp:1711): CV_8UC1, THRESH_BINARY / THRESH_BINARY_INV.
//   ADAPTIVE_THRESH_MEAN_C:     mean = boxFilter(src, blockSize, normalised, BORDER_REPLICATE | BORDER_ISOLATED) (thresh.cpp:1718), any odd blockSize the
//                               box hook serves (u16 sums up to 15 x 15, int32 sums with the reference's float body / double tail beyond);
//   ADAPTIVE_THRESH_GAUSSIAN_C: mean = saturate_cast<uchar>(GaussianBlur(float(src), blockSize, sigma = 0)) (thresh.cpp:1720-1727): the source as
//                               CV_32F, the CV_32F separable path with the taps of getGaussianKernel(blockSize, -1, CV_32F) -- the very hook a CV_32F
//                               cv::GaussianBlur lands in --, cvRound back to 8 bits;
// then the table of :1736-1745 evaluated directly.
extern "C" MI355CV_API int mi355cv_adaptiveThreshold(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                                     double maxValue, int adaptiveMethod, int thresholdType, int blockSize, double C)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0");
    if ((adaptiveMethod != 0 && adaptiveMethod != 1) || (thresholdType != 0 && thresholdType != 1)) return mi355::declined(__func__, __LINE__, "(adaptiveMethod != 0 && adaptiveMethod != 1) || (thresholdType != 0 && thresholdType != 1)");
    if (blockSize < 3 || !(blockSize & 1) || blockSize > (adaptiveMethod == 0 ? lim::ADAPTIVE_MEAN_MAX_BLOCK : lim::SEP_MAX_TAPS)) return mi355::declined(__func__, __LINE__, "blockSize < 3 || even || beyond lim::ADAPTIVE_MEAN_MAX_BLOCK (MEAN_C) / lim::SEP_MAX_TAPS (GAUSSIAN_C)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    double mv = nearbyint(maxValue); mv = mv < 0 ? 0 : mv > 255 ? 255 : mv;                       // saturate_cast<uchar>(maxValue)
    const int idelta = thresholdType == 0 ? (int)ceil(C) : (int)floor(C);
    dim3 grid(divUp(divUp(width, 4), 64), divUp(height, 4));
    if (adaptiveMethod == 0) {
        const size_t mstep = ((size_t)width + 255) & ~(size_t)255;
        uchar* mean = (uchar*)stg.scratch(mstep * height);
        if (!mean) return mi355::declined(__func__, __LINE__, "!mean");
        const int rc = mi355cv_boxFilter(ds, dss, mean, mstep, width, height, D8U, D8U, 1, 0, 0, 0, 0, (size_t)blockSize, (size_t)blockSize, -1, -1, true, B_REPLICATE);
        if (rc != MI355CV_OK) return rc;
        hipLaunchKernelGGL(k_adaptive, grid, dim3(256), 0, stream(), ds, dss, mean, mstep, dd, dds, width, height, idelta, (int)mv, thresholdType);
        return stg.finish("adaptiveThreshold");
    }
    const size_t fstep = (((size_t)width + 63) & ~(size_t)63);                                     // floats per row of the two CV_32F planes
    float* sf = (float*)stg.scratch(fstep * 4 * height);
    float* mf = (float*)stg.scratch(fstep * 4 * height);
    if (!sf || !mf) return mi355::declined(__func__, __LINE__, "!sf || !mf");
    std::vector<double> kd(blockSize);
    if (mi355cv_getGaussianKernel(blockSize, 0.0, kd.data()) != MI355CV_OK) return mi355::declined(__func__, __LINE__, "mi355cv_getGaussianKernel(blockSize, 0.0, kd.data()) != MI355CV_OK");
    std::vector<float> kf(kd.begin(), kd.end());                                                   // getGaussianKernel(n, sigma, CV_32F): the double taps stored as float
    cvhalFilter2D* ctx = nullptr;
    int rc = mi355cv_sepFilterInit(&ctx, MI355CV_MAKETYPE(MI355CV_32F, 1), MI355CV_MAKETYPE(MI355CV_32F, 1), MI355CV_MAKETYPE(MI355CV_32F, 1), (uchar*)kf.data(), blockSize,
                                   (uchar*)kf.data(), blockSize, -1, -1, 0.0, B_REPLICATE);
    if (rc != MI355CV_OK) return rc;
    hipLaunchKernelGGL(k_u8_to_f32, grid, dim3(256), 0, stream(), ds, dss, sf, fstep, width, height);
    rc = mi355cv_sepFilter(ctx, (uchar*)sf, fstep * 4, (uchar*)mf, fstep * 4, width, height, width, height, 0, 0);
    mi355cv_sepFilterFree(ctx);
    if (rc != MI355CV_OK) return rc;
    hipLaunchKernelGGL(k_adaptive_f, grid, dim3(256), 0, stream(), ds, dss, mf, fstep, dd, dds, width, height, idelta, (int)mv, thresholdType);
    return stg.finish("adaptiveThreshold");
}

extern "C" MI355CV_API int mi355cv_threshold(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                             int depth, int cn, double thresh, double maxValue, int thresholdType)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4");
    if (thresholdType < 0 || thresholdType > 4) return mi355::declined(__func__, __LINE__, "thresholdType < 0 || thresholdType > 4");
    // (CV_64F: thresh_64f thresh.cpp:930-1110, the same five rules on doubles -- round 5; CV_32S is not a depth cv::threshold takes, :1677)
    if (depth != D8U && depth != D16U && depth != D16S && depth != D32F && depth != D64F) return mi355::declined(__func__, __LINE__, "depth != D8U && depth != D16U && depth != D16S && depth != D32F && depth != D64F");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    const int e = depth == D8U ? 1 : depth == D32F ? 4 : depth == D64F ? 8 : 2;
    const int n = width * cn;
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)n * e, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)n * e, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    const int epv = 16 / e;
    const int vec = ((((uintptr_t)ds | dss | (uintptr_t)dd | dds) & 15) == 0 && n % epv == 0) ? 1 : 0;
    const int perRow = vec ? n / epv : divUp(n, epv);
    dim3 grid(divUp(perRow, 64), divUp(height, 4));
    hipStream_t st = stream();
#define TH(T_, TY_) hipLaunchKernelGGL((k_threshold<T_, TY_>), grid, dim3(256), 0, st, ds, dss, dd, dds, n, height, (T_)thresh, (T_)maxValue, vec)
#define THT(T_) do { switch (thresholdType) { case 0: TH(T_, 0); break; case 1: TH(T_, 1); break; case 2: TH(T_, 2); break; case 3: TH(T_, 3); break; default: TH(T_, 4); } } while (0)
    switch (depth) {
    case D8U:  THT(uchar); break;
    case D16U: THT(unsigned short); break;
    case D16S: THT(short); break;
    case D64F: THT(double); break;
    default:   THT(float); break;
    }
#undef THT
#undef TH
    return stg.finish("threshold");
}

// batch of device-resident frames: a point operation, so frames that lie back to back are one tall image (one launch); otherwise frame by frame
extern "C" MI355CV_API int mi355cv_thresholdBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride, uchar* dst_data, size_t dst_step,
                                                  size_t dst_frame_stride, int nframes, int width, int height, int depth, int cn, double thresh, double maxValue,
                                                  int thresholdType)
{
    mi355::EntryGuard entry_(__func__);
    if (nframes < 1 || height <= 0) return mi355::declined(__func__, __LINE__, "nframes < 1 || height <= 0");
    if (width > 0 && height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {            // frames in host memory: chunks through two sets of device buffers
        const HostBatch hb = {src_data, src_step, src_frame_stride, (size_t)width * cn * depthBytes(depth), height, dst_data, dst_step, dst_frame_stride, (size_t)width * cn * depthBytes(depth), height, nframes};
        return runHostBatch("thresholdBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_thresholdBatch(s, ss, sf, d, ds, df, nf, width, height, depth, cn, thresh, maxValue, thresholdType); });
    }
    Stager outer;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || !isDevicePtr(src_data) || !isDevicePtr(dst_data)) return setError(MI355CV_NOT_IMPLEMENTED, "thresholdBatch: device-resident frames only");
    if (nframes == 1 || (src_frame_stride == src_step * (size_t)height && dst_frame_stride == dst_step * (size_t)height && (long long)height * nframes < 0x7fffffffLL))
        return mi355cv_threshold(src_data, src_step, dst_data, dst_step, width, height * nframes, depth, cn, thresh, maxValue, thresholdType);
                                               // one synchronisation for the whole batch
    for (int f = 0; f < nframes; f++) {
        const int rc = mi355cv_threshold(src_data + (size_t)f * src_frame_stride, src_step, dst_data + (size_t)f * dst_frame_stride, dst_step, width, height, depth, cn,
                                         thresh, maxValue, thresholdType);
        if (rc != MI355CV_OK) return rc;
    }
    return outer.finish("thresholdBatch");
}
