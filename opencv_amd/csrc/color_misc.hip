// color_misc.hip -- SURVEY.md §8 f1 / f4: the remaining integer colour hooks of the imgproc HAL (CV_8U unless noted).
//   frame egress   cv_hal_cvtBGRtoTwoPlaneYUV   (hal_replacement.hpp:743)  BGR/RGB(A) -> NV12 / NV21
//                  cv_hal_cvtBGRtoThreePlaneYUV (:797)                      BGR/RGB(A) -> I420 / YV12
//                  cv_hal_cvtOnePlaneBGRtoYUV   (:866)                      BGR/RGB(A) -> YUY2 / YVYU / UYVY
//   frame ingest   cv_hal_cvtOnePlaneYUVtoBGR   (:833)                      YUY2 / YVYU / UYVY -> BGR/RGB(A)
//   cv_hal_cvtBGRtoXYZ (:564), cv_hal_cvtXYZtoBGR (:579)                    CV_8U and CV_16U (12-bit fixed point), CV_32F (round 5)
//   cv_hal_cvtBGRtoBGR5x5 (:411), cvtBGR5x5toBGR (:427), cvtBGR5x5toGray (:470), cvtGraytoBGR5x5 (:484)
//   cv_hal_cvtRGBAtoMultipliedRGBA (:894), cvtMultipliedRGBAtoRGBA (:907)
// All of it is byte shuffling plus a handful of integer MACs per pixel: HBM-bound.  The 8-bit conversions run on the pix4 launch shape
// (four pixels per lane, whole-dword traffic; pix4.h), the 4:2:0 encoders on its two-row variant below; CV_16U XYZ keeps one thread per pixel.  Arithmetic follows color_yuv.simd.hpp:1473-1967, color_lab.cpp:251-936, color_rgb.simd.hpp:180-1096.
#include "rt.h"
#include "pix4.h"

using namespace mi355;

namespace {

__device__ __forceinline__ int sat8(int v) { return min(max(v, 0), 255); }

#define PIXEL_XY(W_, H_)                                                     \
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);                      \
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);                       \
    if (x >= (W_) || y >= (H_)) return

// ---------------------------------------------------------------- 4:2:0 encoders: one lane per 4x2 pixels (two chroma samples)
template <int SCN, bool INTERLEAVE>
__global__ __launch_bounds__(256) void k_enc420(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ yp, size_t ystep,
                                                uchar* __restrict__ uvp, size_t uvstep, int W, int H, int swapBlue, int swapUV, int aligned)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y2 = blockIdx.y * 4 + (threadIdx.x >> 6);                     // row pair
    if (x4 >= W || 2 * y2 >= H) return;
    const int n = min(4, W - x4);                                           // 2 or 4 (W is even)
    const bool fast = n == 4 && aligned;
    pix4::Px<SCN> in[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uchar* s = src + (size_t)(2 * y2 + j) * sstep + (size_t)x4 * SCN;
        if (fast) {
#pragma unroll
            for (int i = 0; i < SCN; i++) in[j].w[i] = ((const unsigned*)s)[i];
        } else {
            in[j].clear();
#pragma unroll
            for (int i = 0; i < 4 * SCN; i++) if (i < n * SCN) in[j].put(i, s[i]);
        }
    }
    unsigned uvw = 0; int uu[2], vv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        unsigned yw = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int c0 = in[j].get(p * SCN), g = in[j].get(p * SCN + 1), c2 = in[j].get(p * SCN + 2);
            const int b = swapBlue ? c2 : c0, r = swapBlue ? c0 : c2;       // (a run-time byte index would turn the registers into an array)
            yw |= (unsigned)sat8((269484 * r + 528482 * g + 102760 * b + (1 << 19) + (16 << 20)) >> 20) << (8 * p);
            if (j == 0 && (p & 1) == 0) {                                   // chroma from the top-left pixel of each 2x2 block
                int u = sat8((-155188 * r - 305135 * g + 460324 * b + (1 << 19) + (128 << 20)) >> 20);
                int v = sat8((460324 * r - 385875 * g - 74448 * b + (1 << 19) + (128 << 20)) >> 20);
                if (swapUV) { const int t = u; u = v; v = t; }
                uu[p >> 1] = u; vv[p >> 1] = v;
            }
        }
        uchar* yr = yp + (size_t)(2 * y2 + j) * ystep + x4;
        if (fast) *(unsigned*)yr = yw;
        else {
#pragma unroll
            for (int p = 0; p < 4; p++) if (p < n) yr[p] = (uchar)(yw >> (8 * p));
        }
    }
    if (INTERLEAVE) {
        uchar* uv = uvp + (size_t)y2 * uvstep + x4;
        uvw = (unsigned)uu[0] | ((unsigned)vv[0] << 8) | ((unsigned)uu[1] << 16) | ((unsigned)vv[1] << 24);
        if (fast) *(unsigned*)uv = uvw;
        else {
#pragma unroll
            for (int p = 0; p < 4; p++) if (p < n) uv[p] = (uchar)(uvw >> (8 * p));
        }
    } else {
        const int sRow = 2 * y2, xc = x4 / 2;                               // RGB8toYUV420pInvoker's packed quarter planes (:1609-1610)
        uchar* ur = uvp + uvstep * (sRow / 4) + ((sRow / 2) % 2) * (W / 2) + xc;
        uchar* vr = uvp + uvstep * ((sRow + H) / 4) + (((sRow + H) / 2) % 2) * (W / 2) + xc;
        ur[0] = (uchar)uu[0]; vr[0] = (uchar)vv[0];
        if (n == 4) { ur[1] = (uchar)uu[1]; vr[1] = (uchar)vv[1]; }
    }
}

// ---------------------------------------------------------------- 4:2:2: two pixel pairs per lane
template <int DCN>
struct OpDec422 {
    int bIdx, uidx, vidx, ycn;
    __device__ __forceinline__ void operator()(const pix4::Px<2>& in, pix4::Px<DCN>& out) const {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const unsigned grp = in.w[q];                                   // byte positions inside the group are run-time values
            const int uu = (int)((grp >> (8 * uidx)) & 255u) - 128, vv = (int)((grp >> (8 * vidx)) & 255u) - 128;
            const int ruv = (1 << 19) + 1673527 * vv, guv = (1 << 19) - 852492 * vv - 409993 * uu, buv = (1 << 19) + 2116026 * uu;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int yv = max((int)((grp >> (8 * (ycn + 2 * k))) & 255u) - 16, 0) * 1220542, o = (2 * q + k) * DCN;
                const int r = sat8((yv + ruv) >> 20), g = sat8((yv + guv) >> 20), b = sat8((yv + buv) >> 20);
                out.put(o + 1, g);
                if (bIdx) { out.put(o, r); out.put(o + 2, b); } else { out.put(o, b); out.put(o + 2, r); }
                if (DCN == 4) out.put(o + 3, 255);
            }
        }
    }
};

template <int SCN>
struct OpEnc422 {
    int bIdx, uidx, vidx, ycn;
    __device__ __forceinline__ void operator()(const pix4::Px<SCN>& in, pix4::Px<2>& out) const {
        // the byte positions inside a 4-byte group are run-time values: build the group with shifts
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int o1 = 2 * q * SCN, o2 = o1 + SCN;
            const int b1 = bIdx ? in.get(o1 + 2) : in.get(o1), r1 = bIdx ? in.get(o1) : in.get(o1 + 2), g1 = in.get(o1 + 1);
            const int b2 = bIdx ? in.get(o2 + 2) : in.get(o2), r2 = bIdx ? in.get(o2) : in.get(o2 + 2), g2 = in.get(o2 + 1);
            const int y1 = sat8(((1 << 13) + r1 * 4211 + g1 * 8258 + b1 * 1606 + (1 << 14) * 16) >> 14);
            const int y2 = sat8(((1 << 13) + r2 * 4211 + g2 * 8258 + b2 * 1606 + (1 << 14) * 16) >> 14);
            const int sr = r1 + r2, sg = g1 + g2, sb = b1 + b2;
            const int u = sat8(((1 << 13) + sr * -1212 + sg * -2384 + sb * 3596 + (1 << 13) * 256) >> 14);
            const int v = sat8(((1 << 13) + sr * 3596 + sg * -3015 + sb * -582 + (1 << 13) * 256) >> 14);
            out.w[q] = ((unsigned)y1 << (8 * ycn)) | ((unsigned)y2 << (8 * (ycn + 2))) | ((unsigned)u << (8 * uidx)) | ((unsigned)v << (8 * vidx));
        }
    }
};

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

// CV_32F (RGB2XYZ_f<float> color_lab.cpp:183-247, XYZ2RGB_f<float> :576-642): three products and two sums per output, every one rounded (the reference's v_fma is a
// product and a sum without FMA3 in the baseline).  Its row loop associates  a*C0 + (b*C1 + c*C2)  in the vector body and  (a*C0 + b*C1) + c*C2  in the scalar tail:
// the last W % 4 pixels of a row (4 = pixels per vector of the SSE baseline color_lab.cpp is built for; it is not a dispatched file) take the tail form.
struct Mat3f { float c[9]; };
template <int SCN, int DCN>
__global__ __launch_bounds__(256) void k_xyz_f32(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int nv, Mat3f m)
{
    PIXEL_XY(W, H);
    const float* s = (const float*)(src + (size_t)y * sstep) + (size_t)x * SCN;
    float* d = (float*)(dst + (size_t)y * dstep) + (size_t)x * DCN;
    const float a = s[0], b = s[1], c = s[2];
    const bool vec = x < nv;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float pa = __fmul_rn(a, m.c[3 * k]), pb = __fmul_rn(b, m.c[3 * k + 1]), pc = __fmul_rn(c, m.c[3 * k + 2]);
        d[k] = vec ? __fadd_rn(pa, __fadd_rn(pb, pc)) : __fadd_rn(__fadd_rn(pa, pb), pc);
    }
    if (DCN == 4) d[3] = 1.f;
}

template <int SCN, int DCN>
struct OpXyz8 {
    Mat3 m;
    __device__ __forceinline__ void operator()(const pix4::Px<SCN>& in, pix4::Px<DCN>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int a = in.get(p * SCN), b = in.get(p * SCN + 1), c = in.get(p * SCN + 2);
#pragma unroll
            for (int k = 0; k < 3; k++) out.put(p * DCN + k, sat8((a * m.c[3 * k] + b * m.c[3 * k + 1] + c * m.c[3 * k + 2] + (1 << 11)) >> 12));
            if (DCN == 4) out.put(p * DCN + 3, 255);
        }
    }
};

// ---------------------------------------------------------------- 16-bit packed formats
template <int SCN>
struct OpTo5x5 {
    int bidx, gb;
    __device__ __forceinline__ void operator()(const pix4::Px<SCN>& in, pix4::Px<2>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int b = bidx ? in.get(p * SCN + 2) : in.get(p * SCN), r = bidx ? in.get(p * SCN) : in.get(p * SCN + 2), g = in.get(p * SCN + 1);
            const int a = SCN == 4 ? in.get(p * SCN + 3) : 0;
            const unsigned v = gb == 6 ? (unsigned)((b >> 3) | ((g & ~3) << 3) | ((r & ~7) << 8)) : (unsigned)((b >> 3) | ((g & ~7) << 2) | ((r & ~7) << 7) | (a ? 0x8000 : 0));
            out.w[p >> 1] |= v << (16 * (p & 1));
        }
    }
};

template <int DCN>
struct OpFrom5x5 {
    int bidx, gb;
    __device__ __forceinline__ void operator()(const pix4::Px<2>& in, pix4::Px<DCN>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const unsigned t = (in.w[p >> 1] >> (16 * (p & 1))) & 0xffffu;
            const int b = (t << 3) & 255, g = gb == 6 ? (t >> 3) & 0xfc : (t >> 2) & 0xf8, r = gb == 6 ? (t >> 8) & 0xf8 : (t >> 7) & 0xf8;
            out.put(p * DCN + 1, g);
            if (bidx) { out.put(p * DCN, r); out.put(p * DCN + 2, b); } else { out.put(p * DCN, b); out.put(p * DCN + 2, r); }
            if (DCN == 4) out.put(p * DCN + 3, gb == 6 ? 255 : (int)(t >> 15) * 255);
        }
    }
};

struct Op5x5ToGray {
    int gb;
    __device__ __forceinline__ void operator()(const pix4::Px<2>& in, pix4::Px<1>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int t = (int)((in.w[p >> 1] >> (16 * (p & 1))) & 0xffffu);
            const int b = (t << 3) & 0xf8, g = gb == 6 ? (t >> 3) & 0xfc : (t >> 2) & 0xf8, r = gb == 6 ? (t >> 8) & 0xf8 : (t >> 7) & 0xf8;
            out.put(p, (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
        }
    }
};

struct OpGrayTo5x5 {
    int gb;
    __device__ __forceinline__ void operator()(const pix4::Px<1>& in, pix4::Px<2>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int t = in.get(p), t3 = t >> 3;
            const unsigned v = gb == 6 ? (unsigned)(t3 | ((t & ~3) << 3) | (t3 << 11)) : (unsigned)(t3 | (t3 << 5) | (t3 << 10));
            out.w[p >> 1] |= v << (16 * (p & 1));
        }
    }
};

template <bool UNDO>
struct OpPremul {
    __device__ __forceinline__ void operator()(const pix4::Px<4>& in, pix4::Px<4>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int a = in.get(4 * p + 3);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int v = in.get(4 * p + k);
                out.put(4 * p + k, UNDO ? (a == 0 ? 0 : min((v * 255 + a / 2) / a, 255)) : (v * a + 128) / 255);
            }
            out.put(4 * p + 3, a);
        }
    }
};

// the checks and staging every hook below shares
#define MISC_PROLOGUE(sRowBytes, sRows, dRowBytes, dRows)                                                                        \
    Stager stg; /* first: a declined call must also put the host's device back (~Stager) */                                       \
    if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");                                                  \
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");                           \
    size_t dss, dds;                                                                                                  \
    const uchar* ds = stg.in(src_data, src_step, (size_t)(sRowBytes), (sRows), &dss);                                             \
    uchar* dd = stg.out(dst_data, dst_step, (size_t)(dRowBytes), (dRows), &dds);                                                  \
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");                                                                               \
    hipStream_t st = stream()

} // namespace

extern "C" {

MI355CV_API int mi355cv_cvtBGRtoTwoPlaneYUV(const uchar* src_data, size_t src_step, uchar* y_data, size_t y_step, uchar* uv_data, size_t uv_step,
                                            int width, int height, int scn, bool swapBlue, int uIdx)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (height & 1)) return mi355::declined(__func__, __LINE__, "disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (height & 1)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    size_t dss, ys, uvs;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn, height, &dss);
    uchar* dy = stg.out(y_data, y_step, (size_t)width, height, &ys);
    uchar* duv = stg.out(uv_data, uv_step, (size_t)width, height / 2, &uvs);
    if (!ds || !dy || !duv) return mi355::declined(__func__, __LINE__, "!ds || !dy || !duv");
    dim3 grid(divUp(divUp(width, 4), 64), divUp(height / 2, 4));
    const int al = ((((uintptr_t)ds | dss | (uintptr_t)dy | ys | (uintptr_t)duv | uvs) & 3) == 0) ? 1 : 0;
    if (scn == 3) hipLaunchKernelGGL((k_enc420<3, true>), grid, dim3(256), 0, stream(), ds, dss, dy, ys, duv, uvs, width, height, swapBlue ? 1 : 0, uIdx == 2 ? 1 : 0, al);
    else hipLaunchKernelGGL((k_enc420<4, true>), grid, dim3(256), 0, stream(), ds, dss, dy, ys, duv, uvs, width, height, swapBlue ? 1 : 0, uIdx == 2 ? 1 : 0, al);
    return stg.finish("cvtBGRtoTwoPlaneYUV");
}

MI355CV_API int mi355cv_cvtBGRtoThreePlaneYUV(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                              int scn, bool swapBlue, int uIdx)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (height & 1)) return mi355::declined(__func__, __LINE__, "disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (height & 1)");
    MISC_PROLOGUE(width * scn, height, width, height * 3 / 2);
    dim3 grid(divUp(divUp(width, 4), 64), divUp(height / 2, 4));
    uchar* uv = dd + dds * height;
    const int al = ((((uintptr_t)ds | dss | (uintptr_t)dd | dds) & 3) == 0) ? 1 : 0;
    if (scn == 3) hipLaunchKernelGGL((k_enc420<3, false>), grid, dim3(256), 0, st, ds, dss, dd, dds, uv, dds, width, height, swapBlue ? 1 : 0, uIdx == 2 ? 1 : 0, al);
    else hipLaunchKernelGGL((k_enc420<4, false>), grid, dim3(256), 0, st, ds, dss, dd, dds, uv, dds, width, height, swapBlue ? 1 : 0, uIdx == 2 ? 1 : 0, al);
    return stg.finish("cvtBGRtoThreePlaneYUV");
}

MI355CV_API int mi355cv_cvtOnePlaneYUVtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                            int dcn, bool swapBlue, int uIdx, int ycn)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0 || (width & 1) || (uIdx != 0 && uIdx != 1) || (ycn != 0 && ycn != 1) || (uIdx == 1 && ycn == 1))
        return mi355::declined(__func__, __LINE__, "disabled() || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0 || (width & 1) || (uIdx != 0 && uIdx != 1) || (ycn != 0 && ycn != 1) || (uIdx == 1 && ycn == 1)");
    MISC_PROLOGUE(width * 2, height, width * dcn, height);
    const int uidx = 1 - ycn + uIdx * 2, vidx = (2 + uidx) % 4;
    if (dcn == 3) pix4::launch<2, 3>(st, ds, dss, dd, dds, width, height, OpDec422<3>{swapBlue ? 2 : 0, uidx, vidx, ycn});
    else pix4::launch<2, 4>(st, ds, dss, dd, dds, width, height, OpDec422<4>{swapBlue ? 2 : 0, uidx, vidx, ycn});
    return stg.finish("cvtOnePlaneYUVtoBGR");
}

MI355CV_API int mi355cv_cvtOnePlaneBGRtoYUV(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                            int scn, bool swapBlue, int uIdx, int ycn)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (uIdx != 0 && uIdx != 1) || (ycn != 0 && ycn != 1) || (uIdx == 1 && ycn == 1))
        return mi355::declined(__func__, __LINE__, "disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0 || (width & 1) || (uIdx != 0 && uIdx != 1) || (ycn != 0 && ycn != 1) || (uIdx == 1 && ycn == 1)");
    MISC_PROLOGUE(width * scn, height, width * 2, height);
    const int uidx = 1 - ycn + uIdx * 2, vidx = (2 + uidx) % 4;
    if (scn == 3) pix4::launch<3, 2>(st, ds, dss, dd, dds, width, height, OpEnc422<3>{swapBlue ? 2 : 0, uidx, vidx, ycn});
    else pix4::launch<4, 2>(st, ds, dss, dd, dds, width, height, OpEnc422<4>{swapBlue ? 2 : 0, uidx, vidx, ycn});
    return stg.finish("cvtOnePlaneBGRtoYUV");
}

MI355CV_API int mi355cv_cvtBGRtoXYZ(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int scn, bool swapBlue)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_32F) || (scn != 3 && scn != 4) || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_32F) || (scn != 3 && scn != 4) || width <= 0 || height <= 0");
    const int e = depth == MI355CV_8U ? 1 : depth == MI355CV_16U ? 2 : 4;
    MISC_PROLOGUE(width * scn * e, height, width * 3 * e, height);
    if (depth == MI355CV_32F) {
        static const double kf[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};      // sRGB2XYZ_D65, color_lab.cpp:118
        Mat3f mf; for (int i = 0; i < 9; i++) mf.c[i] = (float)kf[i];
        if (!swapBlue) for (int r = 0; r < 3; r++) std::swap(mf.c[3 * r], mf.c[3 * r + 2]);
        dim3 gridf(divUp(width, 64), divUp(height, 4));
        if (scn == 3) hipLaunchKernelGGL((k_xyz_f32<3, 3>), gridf, dim3(256), 0, st, ds, dss, dd, dds, width, height, width & ~3, mf);
        else hipLaunchKernelGGL((k_xyz_f32<4, 3>), gridf, dim3(256), 0, st, ds, dss, dd, dds, width, height, width & ~3, mf);
        return stg.finish("cvtBGRtoXYZ");
    }
    static const int k[9] = {1689, 1465, 739, 871, 2929, 296, 79, 488, 3892};        // sRGB2XYZ_D65_i, color_lab.cpp:132
    Mat3 m; for (int i = 0; i < 9; i++) m.c[i] = k[i];
    if (!swapBlue) for (int r = 0; r < 3; r++) std::swap(m.c[3 * r], m.c[3 * r + 2]);
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (e == 1) { if (scn == 3) pix4::launch<3, 3>(st, ds, dss, dd, dds, width, height, OpXyz8<3, 3>{m});
                  else pix4::launch<4, 3>(st, ds, dss, dd, dds, width, height, OpXyz8<4, 3>{m}); }
    else { if (scn == 3) hipLaunchKernelGGL((k_xyz<unsigned short, 3, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m);
           else hipLaunchKernelGGL((k_xyz<unsigned short, 4, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m); }
    return stg.finish("cvtBGRtoXYZ");
}

MI355CV_API int mi355cv_cvtXYZtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int dcn, bool swapBlue)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_32F) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_32F) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0");
    const int e = depth == MI355CV_8U ? 1 : depth == MI355CV_16U ? 2 : 4;
    MISC_PROLOGUE(width * 3 * e, height, width * dcn * e, height);
    if (depth == MI355CV_32F) {
        static const double kf[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};   // XYZ2sRGB_D65, color_lab.cpp:125
        Mat3f mf; for (int i = 0; i < 9; i++) mf.c[i] = (float)kf[i];
        if (!swapBlue) for (int c = 0; c < 3; c++) std::swap(mf.c[c], mf.c[6 + c]);
        dim3 gridf(divUp(width, 64), divUp(height, 4));
        if (dcn == 3) hipLaunchKernelGGL((k_xyz_f32<3, 3>), gridf, dim3(256), 0, st, ds, dss, dd, dds, width, height, width & ~3, mf);
        else hipLaunchKernelGGL((k_xyz_f32<3, 4>), gridf, dim3(256), 0, st, ds, dss, dd, dds, width, height, width & ~3, mf);
        return stg.finish("cvtXYZtoBGR");
    }
    static const int k[9] = {13273, -6296, -2042, -3970, 7684, 170, 228, -836, 4331};  // XYZ2sRGB_D65_i, color_lab.cpp:139
    Mat3 m; for (int i = 0; i < 9; i++) m.c[i] = k[i];
    if (!swapBlue) for (int c = 0; c < 3; c++) std::swap(m.c[c], m.c[6 + c]);
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (e == 1) { if (dcn == 3) pix4::launch<3, 3>(st, ds, dss, dd, dds, width, height, OpXyz8<3, 3>{m});
                  else pix4::launch<3, 4>(st, ds, dss, dd, dds, width, height, OpXyz8<3, 4>{m}); }
    else { if (dcn == 3) hipLaunchKernelGGL((k_xyz<unsigned short, 3, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m);
           else hipLaunchKernelGGL((k_xyz<unsigned short, 3, 4>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, m); }
    return stg.finish("cvtXYZtoBGR");
}

MI355CV_API int mi355cv_cvtBGRtoBGR5x5(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                       int scn, bool swapBlue, int greenBits)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (scn != 3 && scn != 4) || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || (scn != 3 && scn != 4) || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0");
    MISC_PROLOGUE(width * scn, height, width * 2, height);
    if (scn == 3) pix4::launch<3, 2>(st, ds, dss, dd, dds, width, height, OpTo5x5<3>{swapBlue ? 2 : 0, greenBits});
    else pix4::launch<4, 2>(st, ds, dss, dd, dds, width, height, OpTo5x5<4>{swapBlue ? 2 : 0, greenBits});
    return stg.finish("cvtBGRtoBGR5x5");
}

MI355CV_API int mi355cv_cvtBGR5x5toBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                       int dcn, bool swapBlue, int greenBits)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (dcn != 3 && dcn != 4) || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || (dcn != 3 && dcn != 4) || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0");
    MISC_PROLOGUE(width * 2, height, width * dcn, height);
    if (dcn == 3) pix4::launch<2, 3>(st, ds, dss, dd, dds, width, height, OpFrom5x5<3>{swapBlue ? 2 : 0, greenBits});
    else pix4::launch<2, 4>(st, ds, dss, dd, dds, width, height, OpFrom5x5<4>{swapBlue ? 2 : 0, greenBits});
    return stg.finish("cvtBGR5x5toBGR");
}

MI355CV_API int mi355cv_cvtBGR5x5toGray(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int greenBits)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0");
    MISC_PROLOGUE(width * 2, height, width, height);
    pix4::launch<2, 1>(st, ds, dss, dd, dds, width, height, Op5x5ToGray{greenBits});
    return stg.finish("cvtBGR5x5toGray");
}

MI355CV_API int mi355cv_cvtGraytoBGR5x5(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int greenBits)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || (greenBits != 5 && greenBits != 6) || width <= 0 || height <= 0");
    MISC_PROLOGUE(width, height, width * 2, height);
    pix4::launch<1, 2>(st, ds, dss, dd, dds, width, height, OpGrayTo5x5{greenBits});
    return stg.finish("cvtGraytoBGR5x5");
}

MI355CV_API int mi355cv_cvtRGBAtoMultipliedRGBA(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0");
    MISC_PROLOGUE(width * 4, height, width * 4, height);
    pix4::launch<4, 4>(st, ds, dss, dd, dds, width, height, OpPremul<false>{});
    return stg.finish("cvtRGBAtoMultipliedRGBA");
}

MI355CV_API int mi355cv_cvtMultipliedRGBAtoRGBA(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0");
    MISC_PROLOGUE(width * 4, height, width * 4, height);
    pix4::launch<4, 4>(st, ds, dss, dd, dds, width, height, OpPremul<true>{});
    return stg.finish("cvtMultipliedRGBAtoRGBA");
}

} // extern "C"
