// color.hip -- row a6 of SURVEY.md §8: cv::cvtColor, RGB <-> gray / channel reorder family.
//
// Reference semantics (modules/imgproc/src/color_rgb.simd.hpp):
//   RGB2Gray<uchar>  :660-748  y = (b*cb + g*cg + r*cr + 2^14) >> 15, {cb,cg,cr} = {3735,19235,9798} (BGR order,
//                              swapped for RGB), color.simd_helpers.hpp:16-24
//   RGB2Gray<ushort> :752-     same integers on 16-bit samples
//   RGB2Gray<float>  :608-657  d = fma(r, cr, fma(g, cg, b*cb)), {.114f,.587f,.299f}  (the AVX2/FMA3 lane formula;
//                              the scalar tail contracts to the same chain under the reference's default flags)
//   Gray2RGB         :386-     b=g=r=gray, alpha = ColorChannel<T>::max() (255 / 65535 / 1.0f)
//   RGB2RGB          :108-     channel reorder / add / drop alpha
// All are HBM-bound element-wise passes (BGR2GRAY 8U: 3 B read + 1 B written per pixel).
#include "rt.h"
#include "pix4.h"

using namespace mi355;

namespace {

constexpr int BY15 = 3735, GY15 = 19235, RY15 = 9798;

template <typename T> __device__ __forceinline__ T grayOf(T c0, T c1, T c2, int k0, int k1, int k2);
template <> __device__ __forceinline__ uchar grayOf<uchar>(uchar c0, uchar c1, uchar c2, int k0, int k1, int k2)
{ return (uchar)((c0 * k0 + c1 * k1 + c2 * k2 + (1 << 14)) >> 15); }
template <> __device__ __forceinline__ unsigned short grayOf<unsigned short>(unsigned short c0, unsigned short c1, unsigned short c2, int k0, int k1, int k2)
{ return (unsigned short)(((unsigned)c0 * k0 + (unsigned)c1 * k1 + (unsigned)c2 * k2 + (1u << 14)) >> 15); }

// generic: one thread per pixel, any alignment / channel count
template <typename T>
__global__ __launch_bounds__(256) void k_bgr2gray_generic(const uchar* __restrict__ src, size_t sstep, size_t sframe,
                                                          uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                          int W, int H, int scn, int k0, int k1, int k2, float f0, float f1, float f2)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const T* s = reinterpret_cast<const T*>(src + (size_t)blockIdx.z * sframe + (size_t)y * sstep) + (size_t)x * scn;
    T* d = reinterpret_cast<T*>(dst + (size_t)blockIdx.z * dframe + (size_t)y * dstep) + x;
    if constexpr (sizeof(T) == 4) {
        const float b = s[0], g = s[1], r = s[2];
        *d = __builtin_fmaf(r, f2, __builtin_fmaf(g, f1, b * f0));
    } else {
        *d = grayOf<T>(s[0], s[1], s[2], k0, k1, k2);
    }
}

// 8U fast path: one lane = 16 output pixels (16*SCN input bytes as SCN dwordx4 loads, one dwordx4 store);
// work item = (row, 1024-pixel strip); needs 16-byte aligned rows and W % 16 == 0.
template <int SCN>
__global__ __launch_bounds__(256) void k_bgr2gray_u8(const uchar* __restrict__ src, size_t sstep, size_t sframe,
                                                     uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                     int nchunks, int H, int k0, int k1, int k2)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);      // 16-pixel chunk
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= nchunks || y >= H) return;
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)blockIdx.z * sframe + (size_t)y * sstep) + (size_t)c * SCN;
    uint32_t w[4 * SCN];
#pragma unroll
    for (int i = 0; i < SCN; i++) { uint4 v = s[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t acc = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int px = 4 * q + p, b0 = px * SCN;             // byte index of channel 0 of pixel px
            const uint32_t c0 = (w[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu;
            const uint32_t c1 = (w[(b0 + 1) >> 2] >> (8 * ((b0 + 1) & 3))) & 0xffu;
            const uint32_t c2 = (w[(b0 + 2) >> 2] >> (8 * ((b0 + 2) & 3))) & 0xffu;
            const uint32_t yv = (c0 * k0 + c1 * k1 + c2 * k2 + (1u << 14)) >> 15;
            acc |= yv << (8 * p);
        }
        o[q] = acc;
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ov = {o[0], o[1], o[2], o[3]};
    __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(dst + (size_t)blockIdx.z * dframe + (size_t)y * dstep) + c);
}

template <typename T>
__global__ __launch_bounds__(256) void k_gray2bgr(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
                                                  int W, int H, int dcn, T alpha)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const T g = reinterpret_cast<const T*>(src + (size_t)y * sstep)[x];
    T* d = reinterpret_cast<T*>(dst + (size_t)y * dstep) + (size_t)x * dcn;
    d[0] = g; d[1] = g; d[2] = g;
    if (dcn == 4) d[3] = alpha;
}

template <typename T>
__global__ __launch_bounds__(256) void k_bgr2bgr(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
                                                 int W, int H, int scn, int dcn, int swapBlue, T alpha)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const T* s = reinterpret_cast<const T*>(src + (size_t)y * sstep) + (size_t)x * scn;
    T* d = reinterpret_cast<T*>(dst + (size_t)y * dstep) + (size_t)x * dcn;
    const T c0 = s[0], c1 = s[1], c2 = s[2];
    const T a = scn == 4 ? s[3] : alpha;
    d[0] = swapBlue ? c2 : c0; d[1] = c1; d[2] = swapBlue ? c0 : c2;
    if (dcn == 4) d[3] = a;
}

// CV_8U channel shuffles on four pixels per lane (pix4.h: whole-dword loads and stores): gray -> BGR(A), BGR(A) <-> BGR(A) / RGB(A)
template <int DCN> struct OpGray2Bgr {
    __device__ __forceinline__ void operator()(const pix4::Px<1>& in, pix4::Px<DCN>& out) const
    {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int g = in.get(p);
            out.put(p * DCN, g); out.put(p * DCN + 1, g); out.put(p * DCN + 2, g);
            if (DCN == 4) out.put(p * DCN + 3, 255);
        }
    }
};
template <int SCN, int DCN> struct OpBgr2Bgr {
    int swapBlue;
    __device__ __forceinline__ void operator()(const pix4::Px<SCN>& in, pix4::Px<DCN>& out) const
    {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int c0 = in.get(p * SCN), c1 = in.get(p * SCN + 1), c2 = in.get(p * SCN + 2);
            out.put(p * DCN, swapBlue ? c2 : c0); out.put(p * DCN + 1, c1); out.put(p * DCN + 2, swapBlue ? c0 : c2);
            if (DCN == 4) out.put(p * DCN + 3, SCN == 4 ? in.get(p * SCN + 3) : 255);
        }
    }
};

int esz(int depth) { return depth == MI355CV_8U ? 1 : depth == MI355CV_16U ? 2 : depth == MI355CV_32F ? 4 : 0; }

int runBgr2Gray(const char* entry, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe,
                int nframes, int W, int H, int depth, int scn, bool swapBlue)
{
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    const int e = esz(depth);
    if (!e || (scn != 3 && scn != 4) || W <= 0 || H <= 0 || nframes <= 0) return mi355::declined(__func__, __LINE__, "!e || (scn != 3 && scn != 4) || W <= 0 || H <= 0 || nframes <= 0");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src, (size_t)W * H, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src, (size_t)W * H, minPixels())");
    size_t dss = sstep, dds = dstep;
    const uchar* ds = src; uchar* dd = dst;
    if (nframes == 1) {
        ds = stg.in(src, sstep, (size_t)W * scn * e, H, &dss);
        dd = stg.out(dst, dstep, (size_t)W * e, H, &dds);
        if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    } else if (!isDevicePtr(src) || !isDevicePtr(dst))
        return setError(MI355CV_NOT_IMPLEMENTED, "%s: batch entry needs device-resident frames", entry);
    // coefficient order follows the source channel order: blueIdx = swapBlue ? 2 : 0 (color_rgb.simd.hpp:668-676)
    int k0 = BY15, k1 = GY15, k2 = RY15;
    float f0 = 0.114f, f1 = 0.587f, f2 = 0.299f;
    if (swapBlue) { int t = k0; k0 = k2; k2 = t; float ft = f0; f0 = f2; f2 = ft; }
    hipStream_t st = stream();
    const bool fast = depth == MI355CV_8U && (W % 16) == 0 && ((((uintptr_t)ds | dss | (uintptr_t)dd | dds | sframe | dframe) & 15) == 0);
    if (fast) {
        dim3 grid(divUp(W / 16, 64), divUp(H, 4), nframes);
        if (scn == 3) hipLaunchKernelGGL((k_bgr2gray_u8<3>), grid, dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W / 16, H, k0, k1, k2);
        else          hipLaunchKernelGGL((k_bgr2gray_u8<4>), grid, dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W / 16, H, k0, k1, k2);
    } else {
        dim3 grid(divUp(W, 64), divUp(H, 4), nframes);
        if (depth == MI355CV_8U)       hipLaunchKernelGGL((k_bgr2gray_generic<uchar>), grid, dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W, H, scn, k0, k1, k2, f0, f1, f2);
        else if (depth == MI355CV_16U) hipLaunchKernelGGL((k_bgr2gray_generic<unsigned short>), grid, dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W, H, scn, k0, k1, k2, f0, f1, f2);
        else                           hipLaunchKernelGGL((k_bgr2gray_generic<float>), grid, dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W, H, scn, k0, k1, k2, f0, f1, f2);
    }
    return stg.finish(entry);
}

} // namespace

extern "C" {

MI355CV_API int mi355cv_cvtBGRtoGray(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step,
                                     int width, int height, int depth, int scn, bool swapBlue)
{
    mi355::EntryGuard entry_(__func__);
    return runBgr2Gray("cvtBGRtoGray", src_data, src_step, 0, dst_data, dst_step, 0, 1, width, height, depth, scn, swapBlue);
}

MI355CV_API int mi355cv_cvtBGRtoGrayBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride,
                                          uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes,
                                          int width, int height, int depth, int scn, int swapBlue)
{
    mi355::EntryGuard entry_(__func__);
    if (width > 0 && height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {            // frames in host memory: chunks through two sets of device buffers
        const HostBatch hb = {src_data, src_step, src_frame_stride, (size_t)width * scn * depthBytes(depth), height, dst_data, dst_step, dst_frame_stride, (size_t)width * depthBytes(depth), height, nframes};
        return runHostBatch("cvtBGRtoGrayBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_cvtBGRtoGrayBatch(s, ss, sf, d, ds, df, nf, width, height, depth, scn, swapBlue); });
    }
    return runBgr2Gray("cvtBGRtoGrayBatch", src_data, src_step, nframes == 1 ? 0 : src_frame_stride, dst_data, dst_step,
                       nframes == 1 ? 0 : dst_frame_stride, nframes, width, height, depth, scn, swapBlue != 0);
}

MI355CV_API int mi355cv_cvtGraytoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step,
                                     int width, int height, int depth, int dcn)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    const int e = esz(depth);
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!e || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0 || !ensureDevice()) return mi355::declined(__func__, __LINE__, "!e || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0 || !ensureDevice()");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * e, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * dcn * e, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (depth == MI355CV_8U) {
        if (dcn == 3) pix4::launch<1, 3>(stream(), ds, dss, dd, dds, width, height, OpGray2Bgr<3>());
        else          pix4::launch<1, 4>(stream(), ds, dss, dd, dds, width, height, OpGray2Bgr<4>());
    }
    else if (depth == MI355CV_16U) hipLaunchKernelGGL((k_gray2bgr<unsigned short>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, dcn, (unsigned short)65535);
    else                           hipLaunchKernelGGL((k_gray2bgr<float>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, dcn, 1.0f);
    return stg.finish("cvtGraytoBGR");
}

MI355CV_API int mi355cv_cvtBGRtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step,
                                    int width, int height, int depth, int scn, int dcn, bool swapBlue)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    const int e = esz(depth);
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!e || (scn != 3 && scn != 4) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0 || !ensureDevice()) return mi355::declined(__func__, __LINE__, "!e || (scn != 3 && scn != 4) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0 || !ensureDevice()");
    if (inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "inPlaceOnDevice(src_data, dst_data)");     // in-place reorder: leave to the caller's path
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn * e, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * dcn * e, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (depth == MI355CV_8U) {
        const int sb = swapBlue ? 1 : 0;
        if (scn == 3 && dcn == 3)      pix4::launch<3, 3>(stream(), ds, dss, dd, dds, width, height, OpBgr2Bgr<3, 3>{sb});
        else if (scn == 3)             pix4::launch<3, 4>(stream(), ds, dss, dd, dds, width, height, OpBgr2Bgr<3, 4>{sb});
        else if (dcn == 3)             pix4::launch<4, 3>(stream(), ds, dss, dd, dds, width, height, OpBgr2Bgr<4, 3>{sb});
        else                           pix4::launch<4, 4>(stream(), ds, dss, dd, dds, width, height, OpBgr2Bgr<4, 4>{sb});
    }
    else if (depth == MI355CV_16U) hipLaunchKernelGGL((k_bgr2bgr<unsigned short>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, scn, dcn, (int)swapBlue, (unsigned short)65535);
    else                           hipLaunchKernelGGL((k_bgr2bgr<float>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, scn, dcn, (int)swapBlue, 1.0f);
    return stg.finish("cvtBGRtoBGR");
}

} // extern "C"
