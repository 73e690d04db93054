// color_misc.hip -- SURVEY.md §8 f1 / f4: the remaining integer colour hooks of the imgproc HAL (CV_8U unless noted).
//   frame egress   cv_hal_cvtBGRtoTwoPlaneYUV   (hal_replacement.hpp:743)  BGR/RGB(A) -> NV12 / NV21
//                  cv_hal_cvtBGRtoThreePlaneYUV (:797)                      BGR/RGB(A) -> I420 / YV12
//                  cv_hal_cvtOnePlaneBGRtoYUV   (:866)                      BGR/RGB(A) -> YUY2 / YVYU / UYVY
//   frame ingest   cv_hal_cvtOnePlaneYUVtoBGR   (:833)                      YUY2 / YVYU / UYVY -> BGR/RGB(A)
//   cv_hal_cvtBGRtoXYZ (:564), cv_hal_cvtXYZtoBGR (:579)                    CV_8U and CV_16U (12-bit fixed point)
//   cv_hal_cvtBGRtoBGR5x5 (:411), cvtBGR5x5toBGR (:427), cvtBGR5x5toGray (:470), cvtGraytoBGR5x5 (:484)
//   cv_hal_cvtRGBAtoMultipliedRGBA (:894), cvtMultipliedRGBAtoRGBA (:907)
// All of it is byte shuffling plus a handful of integer MACs per pixel: HBM-bound, one thread per pixel (pair / 2x2 block for the
// subsampled formats).  Arithmetic follows color_yuv.simd.hpp:1473-1967, color_lab.cpp:251-936, color_rgb.simd.hpp:180-1096.
#include "rt.h"

using namespace mi355;

namespace {

__device__ __forceinline__ int sat8(int v) { return min(max(v, 0), 255); }

#define PIXEL_XY(W_, H_)                                                     \
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);                      \
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);                       \
    if (x >= (W_) || y >= (H_)) return

// ---------------------------------------------------------------- 4:2:0 encoders: one thread per 2x2 block
template <int SCN, bool INTERLEAVE>
__global__ __launch_bounds__(256) void k_enc420(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ yp, size_t ystep,
                                                uchar* __restrict__ uvp, size_t uvstep, int W, int H, int swapBlue, int swapUV)
{
    PIXEL_XY(W / 2, H / 2);
    int r[4], g[4], b[4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uchar* s = src + (size_t)(2 * y + j) * sstep + (size_t)(2 * x + i) * SCN;
            b[2 * j + i] = s[swapBlue ? 2 : 0]; g[2 * j + i] = s[1]; r[2 * j + i] = s[swapBlue ? 0 : 2];
        }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        uchar* yr = yp + (size_t)(2 * y + j) * ystep + 2 * (size_t)x;
#pragma unroll
        for (int i = 0; i < 2; i++)
            yr[i] = (uchar)sat8((269484 * r[2 * j + i] + 528482 * g[2 * j + i] + 102760 * b[2 * j + i] + (1 << 19) + (16 << 20)) >> 20);
    }
    int uu = sat8((-155188 * r[0] - 305135 * g[0] + 460324 * b[0] + (1 << 19) + (128 << 20)) >> 20);
    int vv = sat8((460324 * r[0] - 385875 * g[0] - 74448 * b[0] + (1 << 19) + (128 << 20)) >> 20);
    if (swapUV) { const int t = uu; uu = vv; vv = t; }
    if (INTERLEAVE) {
        uchar* uv = uvp + (size_t)y * uvstep + 2 * (size_t)x;
        uv[0] = (uchar)uu; uv[1] = (uchar)vv;
    } else {
        const int sRow = 2 * y;                                             // RGB8toYUV420pInvoker's packed quarter planes (:1609-1610)
        uvp[uvstep * (sRow / 4) + ((sRow / 2) % 2) * (W / 2) + x] = (uchar)uu;
        uvp[uvstep * ((sRow + H) / 4) + (((sRow + H) / 2) % 2) * (W / 2) + x] = (uchar)vv;
    }
}

// ---------------------------------------------------------------- 4:2:2: one thread per pixel pair (4 source / destination bytes)
template <int DCN>
__global__ __launch_bounds__(256) void k_dec422(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                int bIdx, int uidx, int vidx, int ycn)
{
    PIXEL_XY(W / 2, H);
    const uchar* p = src + (size_t)y * sstep + 4 * (size_t)x;
    const int uu = (int)p[uidx] - 128, vv = (int)p[vidx] - 128;
    const int ruv = (1 << 19) + 1673527 * vv, guv = (1 << 19) - 852492 * vv - 409993 * uu, buv = (1 << 19) + 2116026 * uu;
    uchar* d = dst + (size_t)y * dstep + 2 * (size_t)x * DCN;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int yv = max((int)p[ycn + 2 * k] - 16, 0) * 1220542;
        d[k * DCN + 2 - bIdx] = (uchar)sat8((yv + ruv) >> 20);
        d[k * DCN + 1] = (uchar)sat8((yv + guv) >> 20);
        d[k * DCN + bIdx] = (uchar)sat8((yv + buv) >> 20);
        if (DCN == 4) d[k * DCN + 3] = 255;
    }
}

template <int SCN>
__global__ __launch_bounds__(256) void k_enc422(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                int bIdx, int uidx, int vidx, int ycn)
{
    PIXEL_XY(W / 2, H);
    const uchar* p1 = src + (size_t)y * sstep + 2 * (size_t)x * SCN; const uchar* p2 = p1 + SCN;
    const int r1 = p1[2 - bIdx], g1 = p1[1], b1 = p1[bIdx], r2 = p2[2 - bIdx], g2 = p2[1], b2 = p2[bIdx];
    uchar* row = dst + (size_t)y * dstep + 4 * (size_t)x;
    row[ycn] = (uchar)sat8(((1 << 13) + r1 * 4211 + g1 * 8258 + b1 * 1606 + (1 << 14) * 16) >> 14);
    row[ycn + 2] = (uchar)sat8(((1 << 13) + r2 * 4211 + g2 * 8258 + b2 * 1606 + (1 << 14) * 16) >> 14);
    const int sr = r1 + r2, sg = g1 + g2, sb = b1 + b2;
    row[uidx] = (uchar)sat8(((1 << 13) + sr * -1212 + sg * -2384 + sb * 3596 + (1 << 13) * 256) >> 14);
    row[vidx] = (uchar)sat8(((1 << 13) + sr * 3596 + sg * -3015 + sb * -582 + (1 << 13) * 256) >> 14);
}

// ---------------------------------------------------------------- XYZ
struct Mat3 { int c[9]; };

template <typename T, int SCN, int DCN>
__global__ __launch_bounds__(256) void k_xyz(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, Mat3 m)
{
    PIXEL_XY(W, H);
    const T* s = (const T*)(src + (size_t)y * sstep) + (size_t)x * SCN;
    T* d = (T*)(dst + (size_t)y * dstep) + (size_t)x * DCN;
    const int a = s[0], b = s[1], c = s[2];
    constexpr int hi = sizeof(T) == 1 ? 255 : 65535;
#pragma unroll
    for (int k = 0; k < 3; k++) d[k] = (T)min(max((a * m.c[3 * k] + b * m.c[3 * k + 1] + c * m.c[3 * k + 2] + (1 << 11)) >> 12, 0), hi);
    if (DCN == 4) d[3] = (T)hi;
}

// ---------------------------------------------------------------- 16-bit packed formats
template <int SCN>
__global__ __launch_bounds__(256) void k_to5x5(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int bidx, int gb)
{
    PIXEL_XY(W, H);
    const uchar* s = src + (size_t)y * sstep + (size_t)x * SCN;
    const int r = s[bidx ^ 2], g = s[1], b = s[bidx], a = SCN == 4 ? s[3] : 0;
    ((unsigned short*)(dst + (size_t)y * dstep))[x] = gb == 6 ? (unsigned short)((b >> 3) | ((g & ~3) << 3) | ((r & ~7) << 8))
                                                              : (unsigned short)((b >> 3) | ((g & ~7) << 2) | ((r & ~7) << 7) | (a ? 0x8000 : 0));
}

template <int DCN>
__global__ __launch_bounds__(256) void k_from5x5(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int bidx, int gb)
{
    PIXEL_XY(W, H);
    const unsigned t = ((const unsigned short*)(src + (size_t)y * sstep))[x];
    uchar* d = dst + (size_t)y * dstep + (size_t)x * DCN;
    d[bidx] = (uchar)(t << 3);
    if (gb == 6) { d[1] = (uchar)((t >> 3) & ~3u); d[bidx ^ 2] = (uchar)((t >> 8) & ~7u); if (DCN == 4) d[3] = 255; }
    else { d[1] = (uchar)((t >> 2) & ~7u); d[bidx ^ 2] = (uchar)((t >> 7) & ~7u); if (DCN == 4) d[3] = (uchar)((t >> 15) * 255); }
}

__global__ __launch_bounds__(256) void k_5x5_to_gray(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int gb)
{
    PIXEL_XY(W, H);
    const int t = ((const unsigned short*)(src + (size_t)y * sstep))[x];
    const int b = (t << 3) & 0xf8, g = gb == 6 ? (t >> 3) & 0xfc : (t >> 2) & 0xf8, r = gb == 6 ? (t >> 8) & 0xf8 : (t >> 7) & 0xf8;
    dst[(size_t)y * dstep + x] = (uchar)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
}

__global__ __launch_bounds__(256) void k_gray_to_5x5(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int gb)
{
    PIXEL_XY(W, H);
    const int t = src[(size_t)y * sstep + x], t3 = t >> 3;
    ((unsigned short*)(dst + (size_t)y * dstep))[x] = gb == 6 ? (unsigned short)(t3 | ((t & ~3) << 3) | (t3 << 11)) : (unsigned short)(t3 | (t3 << 5) | (t3 << 10));
}

// ---------------------------------------------------------------- premultiplied alpha
template <bool UNDO>
__global__ __launch_bounds__(256) void k_premul(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H)
{
    PIXEL_XY(W, H);
    const uchar* s = src + (size_t)y * sstep + 4 * (size_t)x;
    uchar* d = dst + (size_t)y * dstep + 4 * (size_t)x;
    const int a = s[3];
#pragma unroll
    for (int k = 0; k < 3; k++) d[k] = UNDO ? (uchar)(a == 0 ? 0 : min((s[k] * 255 + a / 2) / a, 255)) : (uchar)((s[k] * a + 128) / 255);
    d[3] = (uchar)a;
}


// the checks and staging every hook below shares
#define MISC_PROLOGUE(sRowBytes, sRows, dRowBytes, dRows)                                                                        \
    if (!ensureDevice() || src_data == dst_data) return MI355CV_NOT_IMPLEMENTED;                                                  \
    if (!isDevicePtr(src_data) && (size_t)width * height < minPixels()) return MI355CV_NOT_IMPLEMENTED;                           \
    Stager stg; size_t dss, dds;                                                                                                  \
    const uchar* ds = stg.in(src_data, src_step, (size_t)(sRowBytes), (sRows), &dss);                                             \
    uchar* dd = stg.out(dst_data, dst_step, (size_t)(dRowBytes), (dRows), &dds);                                                  \
    if (!ds || !dd) return MI355CV_NOT_IMPLEMENTED;                                                                               \
    hipStream_t st = stream()

} // namespace

extern "C" {

MI355CV_API int mi355cv_cvtBGRtoTwoPlaneYUV(const uchar* src_data, size_t src_step, uchar* y_data, size_t y_step, uchar* uv_data, size_t uv_step,
                                            int width, int height, int scn, bool swapBlue, int uIdx)
{
    if (disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (height & 1)) return MI355CV_NOT_IMPLEMENTED;
    if (!ensureDevice()) return MI355CV_NOT_IMPLEMENTED;
    if (!isDevicePtr(src_data) && (size_t)width * height < minPixels()) return MI355CV_NOT_IMPLEMENTED;
    Stager stg; size_t dss, ys, uvs;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn, height, &dss);
    uchar* dy = stg.out(y_data, y_step, (size_t)width, height, &ys);
    uchar* duv = stg.out(uv_data, uv_step, (size_t)width, height / 2, &uvs);
    if (!ds || !dy || !duv) return MI355CV_NOT_IMPLEMENTED;
    dim3 grid(divUp(width / 2, 64), divUp(height / 2, 4));
    if (scn == 3) hipLaunchKernelGGL((k_enc420<3, true>), grid, dim3(256), 0, stream(), ds, dss, dy, ys, duv, uvs, width, height, swapBlue ? 1 : 0, uIdx == 2 ? 1 : 0);
    else hipLaunchKernelGGL((k_enc420<4, true>), grid, dim3(256), 0, stream(), ds, dss, dy, ys, duv, uvs, width, height, swapBlue ? 1 : 0, uIdx == 2 ? 1 : 0);
    return stg.finish("cvtBGRtoTwoPlaneYUV");
}

MI355CV_API int mi355cv_cvtBGRtoThreePlaneYUV(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                              int scn, bool swapBlue, int uIdx)
{
    if (disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (height & 1)) return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width * scn, height, width, height * 3 / 2);
    dim3 grid(divUp(width / 2, 64), divUp(height / 2, 4));
    uchar* uv = dd + dds * height;
    if (scn == 3) hipLaunchKernelGGL((k_enc420<3, false>), grid, dim3(256), 0, st, ds, dss, dd, dds, uv, dds, width, height, swapBlue ? 1 : 0, uIdx == 2 ? 1 : 0);
    else hipLaunchKernelGGL((k_enc420<4, false>), grid, dim3(256), 0, st, ds, dss, dd, dds, uv, dds, width, height, swapBlue ? 1 : 0, uIdx == 2 ? 1 : 0);
    return stg.finish("cvtBGRtoThreePlaneYUV");
}

MI355CV_API int mi355cv_cvtOnePlaneYUVtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                            int dcn, bool swapBlue, int uIdx, int ycn)
{
    if (disabled() || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0 || (width & 1) || (uIdx != 0 && uIdx != 1) || (ycn != 0 && ycn != 1) || (uIdx == 1 && ycn == 1))
        return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width * 2, height, width * dcn, height);
    const int uidx = 1 - ycn + uIdx * 2, vidx = (2 + uidx) % 4;
    dim3 grid(divUp(width / 2, 64), divUp(height, 4));
    if (dcn == 3) hipLaunchKernelGGL(k_dec422<3>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, uidx, vidx, ycn);
    else hipLaunchKernelGGL(k_dec422<4>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, uidx, vidx, ycn);
    return stg.finish("cvtOnePlaneYUVtoBGR");
}

MI355CV_API int mi355cv_cvtOnePlaneBGRtoYUV(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                            int scn, bool swapBlue, int uIdx, int ycn)
{
    if (disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (uIdx != 0 && uIdx != 1) || (ycn != 0 && ycn != 1) || (uIdx == 1 && ycn == 1))
        return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width * scn, height, width * 2, height);
    const int uidx = 1 - ycn + uIdx * 2, vidx = (2 + uidx) % 4;
    dim3 grid(divUp(width / 2, 64), divUp(height, 4));
    if (scn == 3) hipLaunchKernelGGL(k_enc422<3>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, uidx, vidx, ycn);
    else hipLaunchKernelGGL(k_enc422<4>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, uidx, vidx, ycn);
    return stg.finish("cvtOnePlaneBGRtoYUV");
}

MI355CV_API int mi355cv_cvtBGRtoXYZ(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int scn, bool swapBlue)
{
    if (disabled() || (depth != MI355CV_8U && depth != MI355CV_16U) || (scn != 3 && scn != 4) || width <= 0 || height <= 0) return MI355CV_NOT_IMPLEMENTED;
    const int e = depth == MI355CV_8U ? 1 : 2;
    MISC_PROLOGUE(width * scn * e, height, width * 3 * e, height);
    static const int k[9] = {1689, 1465, 739, 871, 2929, 296, 79, 488, 3892};        // sRGB2XYZ_D65_i, color_lab.cpp:132
    Mat3 m; for (int i = 0; i < 9; i++) m.c[i] = k[i];
    if (!swapBlue) for (int r = 0; r < 3; r++) std::swap(m.c[3 * r], m.c[3 * r + 2]);
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (e == 1) { if (scn == 3) hipLaunchKernelGGL((k_xyz<uchar, 3, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m);
                  else hipLaunchKernelGGL((k_xyz<uchar, 4, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m); }
    else { if (scn == 3) hipLaunchKernelGGL((k_xyz<unsigned short, 3, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m);
           else hipLaunchKernelGGL((k_xyz<unsigned short, 4, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m); }
    return stg.finish("cvtBGRtoXYZ");
}

MI355CV_API int mi355cv_cvtXYZtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int dcn, bool swapBlue)
{
    if (disabled() || (depth != MI355CV_8U && depth != MI355CV_16U) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0) return MI355CV_NOT_IMPLEMENTED;
    const int e = depth == MI355CV_8U ? 1 : 2;
    MISC_PROLOGUE(width * 3 * e, height, width * dcn * e, height);
    static const int k[9] = {13273, -6296, -2042, -3970, 7684, 170, 228, -836, 4331};  // XYZ2sRGB_D65_i, color_lab.cpp:139
    Mat3 m; for (int i = 0; i < 9; i++) m.c[i] = k[i];
    if (!swapBlue) for (int c = 0; c < 3; c++) std::swap(m.c[c], m.c[6 + c]);
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (e == 1) { if (dcn == 3) hipLaunchKernelGGL((k_xyz<uchar, 3, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m);
                  else hipLaunchKernelGGL((k_xyz<uchar, 3, 4>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m); }
    else { if (dcn == 3) hipLaunchKernelGGL((k_xyz<unsigned short, 3, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m);
           else hipLaunchKernelGGL((k_xyz<unsigned short, 3, 4>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m); }
    return stg.finish("cvtXYZtoBGR");
}

MI355CV_API int mi355cv_cvtBGRtoBGR5x5(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                       int scn, bool swapBlue, int greenBits)
{
    if (disabled() || (scn != 3 && scn != 4) || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0) return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width * scn, height, width * 2, height);
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (scn == 3) hipLaunchKernelGGL(k_to5x5<3>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, greenBits);
    else hipLaunchKernelGGL(k_to5x5<4>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, greenBits);
    return stg.finish("cvtBGRtoBGR5x5");
}

MI355CV_API int mi355cv_cvtBGR5x5toBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                       int dcn, bool swapBlue, int greenBits)
{
    if (disabled() || (dcn != 3 && dcn != 4) || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0) return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width * 2, height, width * dcn, height);
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (dcn == 3) hipLaunchKernelGGL(k_from5x5<3>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, greenBits);
    else hipLaunchKernelGGL(k_from5x5<4>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, greenBits);
    return stg.finish("cvtBGR5x5toBGR");
}

MI355CV_API int mi355cv_cvtBGR5x5toGray(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int greenBits)
{
    if (disabled() || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0) return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width * 2, height, width, height);
    hipLaunchKernelGGL(k_5x5_to_gray, dim3(divUp(width, 64), divUp(height, 4)), dim3(256), 0, st, ds, dss, dd, dds, width, height, greenBits);
    return stg.finish("cvtBGR5x5toGray");
}

MI355CV_API int mi355cv_cvtGraytoBGR5x5(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int greenBits)
{
    if (disabled() || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0) return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width, height, width * 2, height);
    hipLaunchKernelGGL(k_gray_to_5x5, dim3(divUp(width, 64), divUp(height, 4)), dim3(256), 0, st, ds, dss, dd, dds, width, height, greenBits);
    return stg.finish("cvtGraytoBGR5x5");
}

MI355CV_API int mi355cv_cvtRGBAtoMultipliedRGBA(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height)
{
    if (disabled() || width <= 0 || height <= 0) return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width * 4, height, width * 4, height);
    hipLaunchKernelGGL(k_premul<false>, dim3(divUp(width, 64), divUp(height, 4)), dim3(256), 0, st, ds, dss, dd, dds, width, height);
    return stg.finish("cvtRGBAtoMultipliedRGBA");
}

MI355CV_API int mi355cv_cvtMultipliedRGBAtoRGBA(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height)
{
    if (disabled() || width <= 0 || height <= 0) return MI355CV_NOT_IMPLEMENTED;
    MISC_PROLOGUE(width * 4, height, width * 4, height);
    hipLaunchKernelGGL(k_premul<true>, dim3(divUp(width, 64), divUp(height, 4)), dim3(256), 0, st, ds, dss, dd, dds, width, height);
    return stg.finish("cvtMultipliedRGBAtoRGBA");
}

} // extern "C"
