// color_yuv.hip -- SURVEY.md §8 f1 / f4 (video ingest): the CV_8U members of the YUV family behind
//   cv_hal_cvtBGRtoYUV (hal_replacement.hpp:500), cv_hal_cvtYUVtoBGR (:533), cv_hal_cvtTwoPlaneYUVtoBGR (:664),
//   cv_hal_cvtTwoPlaneYUVtoBGREx (:701) and cv_hal_cvtThreePlaneYUVtoBGR (:763); callers color_yuv.dispatch.cpp:33, :86, :166, :144, :202.
// Integer arithmetic of color_yuv.simd.hpp (its SIMD bodies and scalar tails agree, so one formula):
//   BGR->YUV/YCrCb  RGB2YCrCb_i<uchar> :398   Y = (c0*s0 + c1*s1 + c2*s2 + 2^13) >> 14,  Cr/V = ((R - Y)*c3 + 128*2^14 + 2^13) >> 14, ...
//   YUV/YCrCb->BGR  YCrCb2RGB_i<uchar> :739   b = Y + ((Cb-128)*c3 + 2^13) >> 14, ...
//   NV12 / NV21     YUV420sp2RGB8Invoker :1195, 20-bit ITU-R BT.601: r = (max(Y-16,0)*1220542 + 1673527*(V-128) + 2^19) >> 20, ...
// All HBM-bound: the packed conversions run on the pix4 launch shape (four pixels per lane, whole-dword traffic; pix4.h), the 4:2:0
// decoders take 4x2 pixels per lane on the same dword traffic.
#include "rt.h"
#include "pix4.h"
#include "hsv_math.h"
#include <cmath>

using namespace mi355;

namespace {

__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

struct YuvFwd { int c0, c1, c2, c3, c4, bidx, yuvOrder; };
struct YuvInv { int c0, c1, c2, c3, bidx, yuvOrder; };

template <int SCN>
struct OpBgr2Yuv {
    YuvFwd a;
    __device__ __forceinline__ void operator()(const pix4::Px<SCN>& in, pix4::Px<3>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int s0 = in.get(p * SCN), s1 = in.get(p * SCN + 1), s2 = in.get(p * SCN + 2);
            const int Y = (s0 * a.c0 + s1 * a.c1 + s2 * a.c2 + (1 << 13)) >> 14;
            const int r = a.bidx ? s0 : s2, b = a.bidx ? s2 : s0;
            const int Cr = sat8(((r - Y) * a.c3 + (128 << 14) + (1 << 13)) >> 14);
            const int Cb = sat8(((b - Y) * a.c4 + (128 << 14) + (1 << 13)) >> 14);
            out.put(p * 3, sat8(Y)); out.put(p * 3 + 1, a.yuvOrder ? Cb : Cr); out.put(p * 3 + 2, a.yuvOrder ? Cr : Cb);
        }
    }
};

template <int DCN>
struct OpYuv2Bgr {
    YuvInv a;
    __device__ __forceinline__ void operator()(const pix4::Px<3>& in, pix4::Px<DCN>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int Y = in.get(p * 3), c1 = in.get(p * 3 + 1), c2 = in.get(p * 3 + 2);
            const int Cr = a.yuvOrder ? c2 : c1, Cb = a.yuvOrder ? c1 : c2;
            const int b = sat8(Y + (((Cb - 128) * a.c3 + (1 << 13)) >> 14));
            const int g = sat8(Y + (((Cb - 128) * a.c2 + (Cr - 128) * a.c1 + (1 << 13)) >> 14));
            const int r = sat8(Y + (((Cr - 128) * a.c0 + (1 << 13)) >> 14));
            out.put(p * DCN, a.bidx ? r : b); out.put(p * DCN + 1, g); out.put(p * DCN + 2, a.bidx ? b : r);
            if (DCN == 4) out.put(p * DCN + 3, 255);
        }
    }
};

// 4:2:0 decoders: one lane per 4x2 pixels -- two luma dwords, two chroma pairs (one dword when interleaved; four byte loads when planar,
// where a chroma sample is addressed through its linear index inside the packed quarter planes that follow the luma rows,
// cvtThreePlaneYUVtoBGR color_yuv.simd.hpp:2060-2087), DCN output dwords per row.  Rows or pitches that are not multiples of four
// and the ragged end of a row use byte accesses.
template <int DCN, bool PLANAR>
__global__ __launch_bounds__(256) void k_dec420(const uchar* __restrict__ yp, size_t ystep, const uchar* __restrict__ cp, size_t cstep,
                                                uchar* __restrict__ dst, size_t dstep, int W, int H, int bIdx, int uIdx, int aligned)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y2 = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= W || 2 * y2 >= H) return;
    const int n = min(4, W - x4);                                           // 2 or 4 (W is even)
    const bool fast = n == 4 && aligned;
    int uu[2], vv[2];
    if (PLANAR) {
        const size_t plane = (size_t)(H / 2) * (size_t)(W / 2);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const size_t l0 = (size_t)y2 * (size_t)(W / 2) + (size_t)(x4 / 2 + (2 * q < n ? q : 0)), l1 = plane + l0;
            const int c0 = cp[(l0 / W) * cstep + l0 % W], c1 = cp[(l1 / W) * cstep + l1 % W];
            uu[q] = (uIdx ? c1 : c0) - 128; vv[q] = (uIdx ? c0 : c1) - 128;
        }
    } else {
        const uchar* uv = cp + (size_t)y2 * cstep + x4;
        unsigned w;
        if (fast) w = *(const unsigned*)uv;
        else { w = (unsigned)uv[0] | ((unsigned)uv[1] << 8); if (n == 4) w |= ((unsigned)uv[2] << 16) | ((unsigned)uv[3] << 24); }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c0 = (int)((w >> (16 * q)) & 255u), c1 = (int)((w >> (16 * q + 8)) & 255u);
            uu[q] = (uIdx ? c1 : c0) - 128; vv[q] = (uIdx ? c0 : c1) - 128;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uchar* yr = yp + (size_t)(2 * y2 + j) * ystep + x4;
        unsigned yw;
        if (fast) yw = *(const unsigned*)yr;
        else { yw = (unsigned)yr[0] | ((unsigned)yr[1] << 8); if (n == 4) yw |= ((unsigned)yr[2] << 16) | ((unsigned)yr[3] << 24); }
        pix4::Px<DCN> out; out.clear();
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int u = uu[p >> 1], v = vv[p >> 1];
            const int ruv = (1 << 19) + 1673527 * v, guv = (1 << 19) - 852492 * v - 409993 * u, buv = (1 << 19) + 2116026 * u;
            const int yv = max((int)((yw >> (8 * p)) & 255u) - 16, 0) * 1220542;
            const int r = sat8((yv + ruv) >> 20), g = sat8((yv + guv) >> 20), b = sat8((yv + buv) >> 20);
            out.put(p * DCN, bIdx ? r : b); out.put(p * DCN + 1, g); out.put(p * DCN + 2, bIdx ? b : r);
            if (DCN == 4) out.put(p * DCN + 3, 255);
        }
        uchar* d = dst + (size_t)(2 * y2 + j) * dstep + (size_t)x4 * DCN;
        if (fast) {
#pragma unroll
            for (int i = 0; i < DCN; i++) ((unsigned*)d)[i] = out.w[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4 * DCN; i++) if (i < n * DCN) d[i] = (uchar)out.get(i);
        }
    }
}

// BGR/RGB -> HSV, CV_8U: RGB2HSV_b color_hsv.simd.hpp:47-262 -- integer arithmetic with the two reciprocal tables (hsv_shift 12)
// that the host builds exactly as TablesSingleton does and passes in HBM.
template <int SCN>
struct OpBgr2Hsv {
    int bidx, hr; const int* sdiv; const int* hdiv;
    __device__ __forceinline__ void operator()(const pix4::Px<SCN>& in, pix4::Px<3>& out) const {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int c0 = in.get(p * SCN), g = in.get(p * SCN + 1), c2 = in.get(p * SCN + 2);
            const int b = bidx ? c2 : c0, r = bidx ? c0 : c2;
            const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
            const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
            const int sat = (diff * sdiv[v] + (1 << 11)) >> 12;
            int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
            hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
            hh += hh < 0 ? hr : 0;
            out.put(p * 3, sat8(hh)); out.put(p * 3 + 1, sat); out.put(p * 3 + 2, v);
        }
    }
};

// HSV -> BGR/RGB(A), CV_8U: HSV2RGB_b color_hsv.simd.hpp:518-667.  The reference truncates to 8 bits inside its vector loop (and its AVX2
// object evaluates 1 - s*x fused) but rounds in the scalar tail, so the result depends on where a row splits: `body` = the pixels the
// 8-lane (AVX2) loop covers, floor(W / 32) * 32 -- the width every x86 host with AVX2 runs (the build the parity tests compare with).
template <int DCN>
__global__ __launch_bounds__(256) void k_hsv2bgr_u8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                    int bidx, float hscale, int body)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const uchar* s = src + (size_t)y * sstep + (size_t)x * 3;
    uchar* d = dst + (size_t)y * dstep + (size_t)x * DCN;
    int b, g, r;
    mi355_hsv2bgr_px(s[0], s[1], s[2], x < body, hscale, b, g, r);
    d[bidx] = (uchar)b; d[1] = (uchar)g; d[bidx ^ 2] = (uchar)r;
    if (DCN == 4) d[3] = 255;
}

// BGR/RGB(A) <-> HLS for CV_8U / CV_32F and <-> HSV for CV_32F (hsv_math.h has the arithmetic and its provenance): one thread per pixel.
// MODE 0: CV_8U HLS (bytes in, bytes out, the reference's vector-body / scalar-tail split per 256-pixel block), 1: CV_32F HLS, 2: CV_32F HSV.
template <int SCN, int MODE>
__global__ __launch_bounds__(256) void k_bgr2hxx(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int bidx, float hscale)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    if (MODE == 0) {
        const uchar* s = src + (size_t)y * sstep + (size_t)x * SCN;
        uchar* d = dst + (size_t)y * dstep + (size_t)x * 3;
        float h, l, sa;
        mi355_rgb2hls_px(s[bidx ^ 2] * (1.f / 255.f), s[1] * (1.f / 255.f), s[bidx] * (1.f / 255.f), hscale, mi355_hls_in_vector_body(x, W), h, l, sa);
        d[0] = (uchar)mi355_round_sat8(h); d[1] = (uchar)mi355_round_sat8(l * 255.f); d[2] = (uchar)mi355_round_sat8(sa * 255.f);
    } else {
        const float* s = reinterpret_cast<const float*>(src + (size_t)y * sstep) + (size_t)x * SCN;
        float* d = reinterpret_cast<float*>(dst + (size_t)y * dstep) + (size_t)x * 3;
        float o0, o1, o2;
        if (MODE == 1) mi355_rgb2hls_px(s[bidx ^ 2], s[1], s[bidx], 1.f, false, o0, o1, o2);
        else mi355_rgb2hsv_f(s[bidx ^ 2], s[1], s[bidx], o0, o1, o2);
        d[0] = o0; d[1] = o1; d[2] = o2;
    }
}
template <int DCN, int MODE>
__global__ __launch_bounds__(256) void k_hxx2bgr(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int bidx, float hscale)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float b, g, r;
    if (MODE == 0) {
        const uchar* s = src + (size_t)y * sstep + (size_t)x * 3;
        uchar* d = dst + (size_t)y * dstep + (size_t)x * DCN;
        mi355_hls2rgb_px((float)s[0], s[1] * (1.f / 255.f), s[2] * (1.f / 255.f), hscale, mi355_hls_in_vector_body(x, W), b, g, r);
        d[bidx] = (uchar)mi355_round_sat8(b * 255.f); d[1] = (uchar)mi355_round_sat8(g * 255.f); d[bidx ^ 2] = (uchar)mi355_round_sat8(r * 255.f);
        if (DCN == 4) d[3] = 255;
    } else {
        const float* s = reinterpret_cast<const float*>(src + (size_t)y * sstep) + (size_t)x * 3;
        float* d = reinterpret_cast<float*>(dst + (size_t)y * dstep) + (size_t)x * DCN;
        if (MODE == 1) mi355_hls2rgb_px(s[0], s[1], s[2], 6.f / 360.f, false, b, g, r);
        else mi355_hsv2rgb_f(s[0], s[1], s[2], b, g, r);
        d[bidx] = b; d[1] = g; d[bidx ^ 2] = r;
        if (DCN == 4) d[3] = 1.f;
    }
}

// CV_16U / CV_32F members of the YUV / YCrCb family (RGB2YCrCb_i<ushort> color_yuv.simd.hpp:255-395, YCrCb2RGB_i<ushort> :890-1010: the 8-bit integer
// formulas with delta = 32768; RGB2YCrCb_f<float> :134-212, YCrCb2RGB_f<float> :616-689: fused multiply-adds in the order of the reference's vector
// loop).  One thread per pixel, 6-8 bytes (16U) / 12-16 bytes (32F) in and out per lane.
struct YuvFwdF { float c0, c1, c2, c3, c4; int bidx, yuvOrder; };
struct YuvInvF { float c0, c1, c2, c3; int bidx, yuvOrder; };

template <int SCN>
__global__ __launch_bounds__(256) void k_bgr2yuv16(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int w, int h, YuvFwd a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const unsigned short* s = reinterpret_cast<const unsigned short*>(src + (size_t)y * sstep) + (size_t)x * SCN;
    unsigned short* d = reinterpret_cast<unsigned short*>(dst + (size_t)y * dstep) + (size_t)x * 3;
    const int s0 = s[0], s1 = s[1], s2 = s[2];
    const int Y = (int)(((long long)s0 * a.c0 + (long long)s1 * a.c1 + (long long)s2 * a.c2 + (1 << 13)) >> 14);
    const int r = a.bidx ? s0 : s2, b = a.bidx ? s2 : s0;
    const int delta = 32768 * (1 << 14);
    const int Cr = ((r - Y) * a.c3 + delta + (1 << 13)) >> 14, Cb = ((b - Y) * a.c4 + delta + (1 << 13)) >> 14;
    const int cr = min(max(Cr, 0), 65535), cb = min(max(Cb, 0), 65535);
    d[0] = (unsigned short)min(max(Y, 0), 65535); d[1] = (unsigned short)(a.yuvOrder ? cb : cr); d[2] = (unsigned short)(a.yuvOrder ? cr : cb);
}

template <int DCN>
__global__ __launch_bounds__(256) void k_yuv2bgr16(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int w, int h, YuvInv a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const unsigned short* s = reinterpret_cast<const unsigned short*>(src + (size_t)y * sstep) + (size_t)x * 3;
    unsigned short* d = reinterpret_cast<unsigned short*>(dst + (size_t)y * dstep) + (size_t)x * DCN;
    const int Y = s[0], c1 = s[1], c2 = s[2];
    const int Cr = a.yuvOrder ? c2 : c1, Cb = a.yuvOrder ? c1 : c2;
    const int b = Y + (int)(((long long)(Cb - 32768) * a.c3 + (1 << 13)) >> 14);
    const int g = Y + (int)(((long long)(Cb - 32768) * a.c2 + (long long)(Cr - 32768) * a.c1 + (1 << 13)) >> 14);
    const int r = Y + (int)(((long long)(Cr - 32768) * a.c0 + (1 << 13)) >> 14);
    const unsigned short B = (unsigned short)min(max(b, 0), 65535), G = (unsigned short)min(max(g, 0), 65535), R = (unsigned short)min(max(r, 0), 65535);
    d[0] = a.bidx ? R : B; d[1] = G; d[2] = a.bidx ? B : R;
    if (DCN == 4) d[3] = 65535;
}

template <int SCN>
__global__ __launch_bounds__(256) void k_bgr2yuv32f(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int w, int h, YuvFwdF a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float* s = reinterpret_cast<const float*>(src + (size_t)y * sstep) + (size_t)x * SCN;
    float* d = reinterpret_cast<float*>(dst + (size_t)y * dstep) + (size_t)x * 3;
    const float s0 = s[0], s1 = s[1], s2 = s[2];
    const float Y = __fmaf_rn(s0, a.c0, __fmaf_rn(s1, a.c1, __fmul_rn(s2, a.c2)));
    const float r = a.bidx ? s0 : s2, b = a.bidx ? s2 : s0;
    const float Cr = __fmaf_rn(__fsub_rn(r, Y), a.c3, 0.5f), Cb = __fmaf_rn(__fsub_rn(b, Y), a.c4, 0.5f);
    d[0] = Y; d[1] = a.yuvOrder ? Cb : Cr; d[2] = a.yuvOrder ? Cr : Cb;
}

template <int DCN>
__global__ __launch_bounds__(256) void k_yuv2bgr32f(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int w, int h, YuvInvF a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float* s = reinterpret_cast<const float*>(src + (size_t)y * sstep) + (size_t)x * 3;
    float* d = reinterpret_cast<float*>(dst + (size_t)y * dstep) + (size_t)x * DCN;
    const float Y = s[0], c1 = s[1], c2 = s[2];
    const float cr = __fsub_rn(a.yuvOrder ? c2 : c1, 0.5f), cb = __fsub_rn(a.yuvOrder ? c1 : c2, 0.5f);
    const float b = __fmaf_rn(cb, a.c3, Y), g = __fmaf_rn(cr, a.c1, __fmaf_rn(cb, a.c2, Y)), r = __fmaf_rn(cr, a.c0, Y);
    d[0] = a.bidx ? r : b; d[1] = g; d[2] = a.bidx ? b : r;
    if (DCN == 4) d[3] = 1.f;
}

} // namespace


extern "C" {

MI355CV_API int mi355cv_cvtBGRtoYUV(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int scn, bool swapBlue, bool isCbCr)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_32F) || (scn != 3 && scn != 4) || width <= 0 || height <= 0)
        return mi355::declined(__func__, __LINE__, "disabled() || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_32F) || (scn != 3 && scn != 4) || width <= 0 || height <= 0");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    const size_t esz = depth == MI355CV_8U ? 1 : depth == MI355CV_16U ? 2 : 4;
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn * esz, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * 3 * esz, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    YuvFwd a; a.c0 = 4899; a.c1 = 9617; a.c2 = 1868; a.c3 = isCbCr ? 11682 : 14369; a.c4 = isCbCr ? 9241 : 8061;
    a.bidx = swapBlue ? 2 : 0; a.yuvOrder = isCbCr ? 0 : 1;
    if (a.bidx == 0) { const int t = a.c0; a.c0 = a.c2; a.c2 = t; }
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (depth == MI355CV_16U) {
        if (scn == 3) hipLaunchKernelGGL(k_bgr2yuv16<3>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, a);
        else hipLaunchKernelGGL(k_bgr2yuv16<4>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, a);
        return stg.finish("cvtBGRtoYUV");
    }
    if (depth == MI355CV_32F) {
        YuvFwdF f; f.c0 = 0.299f; f.c1 = 0.587f; f.c2 = 0.114f; f.c3 = isCbCr ? 0.713f : 0.877f; f.c4 = isCbCr ? 0.564f : 0.492f;   // R2YF.., YCRF / R2VF, YCBF / B2UF
        f.bidx = a.bidx; f.yuvOrder = a.yuvOrder;
        if (f.bidx == 0) { const float t = f.c0; f.c0 = f.c2; f.c2 = t; }
        if (scn == 3) hipLaunchKernelGGL(k_bgr2yuv32f<3>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, f);
        else hipLaunchKernelGGL(k_bgr2yuv32f<4>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, f);
        return stg.finish("cvtBGRtoYUV");
    }
    if (scn == 3) pix4::launch<3, 3>(stream(), ds, dss, dd, dds, width, height, OpBgr2Yuv<3>{a});
    else pix4::launch<4, 3>(stream(), ds, dss, dd, dds, width, height, OpBgr2Yuv<4>{a});
    return stg.finish("cvtBGRtoYUV");
}

MI355CV_API int mi355cv_cvtYUVtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int dcn, bool swapBlue, bool isCbCr)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_32F) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0)
        return mi355::declined(__func__, __LINE__, "disabled() || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_32F) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    const size_t esz = depth == MI355CV_8U ? 1 : depth == MI355CV_16U ? 2 : 4;
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * 3 * esz, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * dcn * esz, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    YuvInv a; a.c0 = isCbCr ? 22987 : 18678; a.c1 = isCbCr ? -11698 : -9519; a.c2 = isCbCr ? -5636 : -6472; a.c3 = isCbCr ? 29049 : 33292;
    a.bidx = swapBlue ? 2 : 0; a.yuvOrder = isCbCr ? 0 : 1;
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (depth == MI355CV_16U) {
        if (dcn == 3) hipLaunchKernelGGL(k_yuv2bgr16<3>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, a);
        else hipLaunchKernelGGL(k_yuv2bgr16<4>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, a);
        return stg.finish("cvtYUVtoBGR");
    }
    if (depth == MI355CV_32F) {
        YuvInvF f;                                             // CR2RF, CR2GF, CB2GF, CB2BF / V2RF, V2GF, U2GF, U2BF (color.simd_helpers / color_yuv.simd.hpp)
        f.c0 = isCbCr ? 1.403f : 1.140f; f.c1 = isCbCr ? -0.714f : -0.581f; f.c2 = isCbCr ? -0.344f : -0.395f; f.c3 = isCbCr ? 1.773f : 2.032f;
        f.bidx = a.bidx; f.yuvOrder = a.yuvOrder;
        if (dcn == 3) hipLaunchKernelGGL(k_yuv2bgr32f<3>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, f);
        else hipLaunchKernelGGL(k_yuv2bgr32f<4>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, f);
        return stg.finish("cvtYUVtoBGR");
    }
    if (dcn == 3) pix4::launch<3, 3>(stream(), ds, dss, dd, dds, width, height, OpYuv2Bgr<3>{a});
    else pix4::launch<3, 4>(stream(), ds, dss, dd, dds, width, height, OpYuv2Bgr<4>{a});
    return stg.finish("cvtYUVtoBGR");
}

MI355CV_API int mi355cv_cvtTwoPlaneYUVtoBGREx(const uchar* y_data, size_t y_step, const uchar* uv_data, size_t uv_step, uchar* dst_data, size_t dst_step,
                                              int dst_width, int dst_height, int dcn, bool swapBlue, int uIdx)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (dcn != 3 && dcn != 4) || dst_width <= 0 || dst_height <= 0 || (dst_width & 1) || (dst_height & 1) || (uIdx != 0 && uIdx != 1))
        return mi355::declined(__func__, __LINE__, "disabled() || (dcn != 3 && dcn != 4) || dst_width <= 0 || dst_height <= 0 || (dst_width & 1) || (dst_height & 1) || (uIdx != 0 && uIdx != 1)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(y_data, (size_t)dst_width * dst_height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(y_data, (size_t)dst_width * dst_height, minPixels())");
    size_t ys, uvs, dds;
    const uchar* dy = stg.in(y_data, y_step, (size_t)dst_width, dst_height, &ys);
    const uchar* duv = stg.in(uv_data, uv_step, (size_t)dst_width, dst_height / 2, &uvs);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)dst_width * dcn, dst_height, &dds);
    if (!dy || !duv || !dd) return mi355::declined(__func__, __LINE__, "!dy || !duv || !dd");
    dim3 grid(divUp(divUp(dst_width, 4), 64), divUp(dst_height / 2, 4));
    const int al = ((((uintptr_t)dy | ys | (uintptr_t)duv | uvs | (uintptr_t)dd | dds) & 3) == 0) ? 1 : 0;
    if (dcn == 3) hipLaunchKernelGGL((k_dec420<3, false>), grid, dim3(256), 0, stream(), dy, ys, duv, uvs, dd, dds, dst_width, dst_height, swapBlue ? 2 : 0, uIdx, al);
    else hipLaunchKernelGGL((k_dec420<4, false>), grid, dim3(256), 0, stream(), dy, ys, duv, uvs, dd, dds, dst_width, dst_height, swapBlue ? 2 : 0, uIdx, al);
    return stg.finish("cvtTwoPlaneYUVtoBGR");
}

// replaces hal_ni_cvtBGRtoHSV (hal_replacement.hpp:596; caller color_hsv.dispatch.cpp:65): HSV and HLS, CV_8U and CV_32F (cvtBGRtoHSV color_hsv.simd.hpp:1270-1293)
MI355CV_API int mi355cv_cvtBGRtoHSV(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int scn, bool swapBlue, bool isFullRange, bool isHSV)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (depth != MI355CV_8U && depth != MI355CV_32F) || (scn != 3 && scn != 4) || width <= 0 || height <= 0)
        return mi355::declined(__func__, __LINE__, "disabled() || (depth != MI355CV_8U && depth != MI355CV_32F) || (scn != 3 && scn != 4) || width <= 0 || height <= 0");
    if (depth != MI355CV_8U || !isHSV) {
        Stager stg;
        if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");
        if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "host image below the policy threshold");
        const size_t e = depth == MI355CV_8U ? 1 : 4;
        size_t dss, dds;
        const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn * e, height, &dss);
        uchar* dd = stg.out(dst_data, dst_step, (size_t)width * 3 * e, height, &dds);
        if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
        dim3 grid(divUp(width, 64), divUp(height, 4));
        const int bidx = swapBlue ? 2 : 0;
        const float hscale = (isFullRange ? 256.f : 180.f) / 360.f;
#define HXX_F(SCN_, MODE_) hipLaunchKernelGGL((k_bgr2hxx<SCN_, MODE_>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, bidx, hscale)
        if (depth == MI355CV_8U) { if (scn == 3) HXX_F(3, 0); else HXX_F(4, 0); }
        else if (!isHSV)         { if (scn == 3) HXX_F(3, 1); else HXX_F(4, 1); }
        else                     { if (scn == 3) HXX_F(3, 2); else HXX_F(4, 2); }
#undef HXX_F
        noteKernel("k_bgr2hxx<%d,%s> grid=%ux%u x256", scn, depth == MI355CV_8U ? "HLS 8U" : isHSV ? "HSV 32F" : "HLS 32F", grid.x, grid.y);
        return stg.finish("cvtBGRtoHSV");
    }
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * 3, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    const int hr = isFullRange ? 256 : 180;
    int tabs[512];                                                       // sdiv_table, hdiv_table (color_hsv.simd.hpp:70-78)
    tabs[0] = tabs[256] = 0;
    for (int i = 1; i < 256; i++) { tabs[i] = (int)nearbyint((255 << 12) / (1. * i)); tabs[256 + i] = (int)nearbyint((hr << 12) / (6. * i)); }
    const int* dt = (const int*)stg.param(tabs, sizeof tabs);
    if (!dt) return mi355::declined(__func__, __LINE__, "!dt");
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (scn == 3) pix4::launch<3, 3>(stream(), ds, dss, dd, dds, width, height, OpBgr2Hsv<3>{swapBlue ? 2 : 0, hr, dt, dt + 256});
    else pix4::launch<4, 3>(stream(), ds, dss, dd, dds, width, height, OpBgr2Hsv<4>{swapBlue ? 2 : 0, hr, dt, dt + 256});
    return stg.finish("cvtBGRtoHSV");
}

MI355CV_API int mi355cv_cvtThreePlaneYUVtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int dst_width, int dst_height,
                                              int dcn, bool swapBlue, int uIdx)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (dcn != 3 && dcn != 4) || dst_width <= 0 || dst_height <= 0 || (dst_width & 1) || (dst_height & 1) || (uIdx != 0 && uIdx != 1))
        return mi355::declined(__func__, __LINE__, "disabled() || (dcn != 3 && dcn != 4) || dst_width <= 0 || dst_height <= 0 || (dst_width & 1) || (dst_height & 1) || (uIdx != 0 && uIdx != 1)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)dst_width * dst_height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)dst_width * dst_height, minPixels())");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)dst_width, dst_height * 3 / 2, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)dst_width * dcn, dst_height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    dim3 grid(divUp(divUp(dst_width, 4), 64), divUp(dst_height / 2, 4));
    const int al = ((((uintptr_t)ds | dss | (uintptr_t)dd | dds) & 3) == 0) ? 1 : 0;
    const uchar* chroma = ds + dss * (size_t)dst_height;                    // the packed quarter planes start below the luma rows
    if (dcn == 3) hipLaunchKernelGGL((k_dec420<3, true>), grid, dim3(256), 0, stream(), ds, dss, chroma, dss, dd, dds, dst_width, dst_height, swapBlue ? 2 : 0, uIdx, al);
    else hipLaunchKernelGGL((k_dec420<4, true>), grid, dim3(256), 0, stream(), ds, dss, chroma, dss, dd, dds, dst_width, dst_height, swapBlue ? 2 : 0, uIdx, al);
    return stg.finish("cvtThreePlaneYUVtoBGR");
}

MI355CV_API int mi355cv_cvtTwoPlaneYUVtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int dst_width, int dst_height,
                                            int dcn, bool swapBlue, int uIdx)
{
    mi355::EntryGuard entry_(__func__);
    return mi355cv_cvtTwoPlaneYUVtoBGREx(src_data, src_step, src_data + src_step * (size_t)dst_height, src_step, dst_data, dst_step, dst_width, dst_height,
                                         dcn, swapBlue, uIdx);
}

// cv_hal_cvtHSVtoBGR (hal_replacement.hpp:613; caller color_hsv.dispatch.cpp:95): HSV and HLS, CV_8U and CV_32F (cvtHSVtoBGR color_hsv.simd.hpp:1296-1320)
MI355CV_API int mi355cv_cvtHSVtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int dcn, bool swapBlue, bool isFullRange, bool isHSV)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (depth != MI355CV_8U && depth != MI355CV_32F) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0)
        return mi355::declined(__func__, __LINE__, "disabled() || (depth != MI355CV_8U && depth != MI355CV_32F) || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0");
    if (depth != MI355CV_8U || !isHSV) {
        Stager stg;
        if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");
        if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "host image below the policy threshold");
        const size_t e = depth == MI355CV_8U ? 1 : 4;
        size_t dss, dds;
        const uchar* ds = stg.in(src_data, src_step, (size_t)width * 3 * e, height, &dss);
        uchar* dd = stg.out(dst_data, dst_step, (size_t)width * dcn * e, height, &dds);
        if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
        dim3 grid(divUp(width, 64), divUp(height, 4));
        const int bidx = swapBlue ? 2 : 0;
        const float hscale = 6.f / (isFullRange ? 255.f : 180.f);
#define HXX_I(DCN_, MODE_) hipLaunchKernelGGL((k_hxx2bgr<DCN_, MODE_>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, bidx, hscale)
        if (depth == MI355CV_8U) { if (dcn == 3) HXX_I(3, 0); else HXX_I(4, 0); }
        else if (!isHSV)         { if (dcn == 3) HXX_I(3, 1); else HXX_I(4, 1); }
        else                     { if (dcn == 3) HXX_I(3, 2); else HXX_I(4, 2); }
#undef HXX_I
        noteKernel("k_hxx2bgr<%d,%s> grid=%ux%u x256", dcn, depth == MI355CV_8U ? "HLS 8U" : isHSV ? "HSV 32F" : "HLS 32F", grid.x, grid.y);
        return stg.finish("cvtHSVtoBGR");
    }
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * 3, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * dcn, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    const float hscale = 6.0f / (isFullRange ? 255 : 180);
    const int body = (width / 32) * 32;
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (dcn == 3) hipLaunchKernelGGL(k_hsv2bgr_u8<3>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, hscale, body);
    else hipLaunchKernelGGL(k_hsv2bgr_u8<4>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, swapBlue ? 2 : 0, hscale, body);
    return stg.finish("cvtHSVtoBGR");
}

} // extern "C"
