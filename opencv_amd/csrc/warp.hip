// warp.hip -- rows a7/a8/a9 of SURVEY.md §8: cv::resize, cv::warpAffine, cv::warpPerspective, cv::remap (32F maps).
//
// Reference semantics restated (all coordinate arithmetic is integer or IEEE double/float without contraction, so the
// device reproduces it bit for bit):
//   resize   hal::resize resize.cpp:3826-4194.  NEAREST: sx = min(floor(dx/fx), w-1) (:1026-1100).  LINEAR:
//            fx = float((dx+.5)*scale-.5), sx = floor(fx) (:4097-4190); 8U runs in fixed point -- 11-bit taps,
//            dst = ((b0*(t0>>4))>>16) + ((b1*(t1>>4))>>16) + 2) >> 2 (:1963-1989); other depths float mul/mul/add
//            (HResizeLinear :1877, VResizeLinear :1931; resize.cpp is built without FMA).  LINEAR at exactly 1/2 and
//            INTER_AREA with integer ratios take resizeAreaFast_ (:2919-3060): (a+b+c+d+2)>>2 for 2x2 integers.
//   warpAffine  hal::warpAffine imgwarp.cpp:2673: X = (sat_int((M1*y+M2)*1024) + 16 + sat_int(M0*x*1024)) >> 5,
//            pixel = X>>5, fraction = X&31 (:2254-2272, :2772-2780); then remapBilinear :675 with the 32x32 weight
//            table of initInterTab2D :213-288 (Q15 for 8U, float otherwise) and the border rules :819-900.
//   warpPerspective  imgwarp.cpp:3160-3226: X0,Y0,W0 evaluated at the 64-pixel block origin, then
//            (X0 + M0*x1) * (32 / (W0 + M6*x1)) in double, cvRound (:3349-3361).
//   remap    RemapInvoker imgwarp.cpp:1130-: 32F maps -> cvRound(map*32) -> same sampler.
// All kernels are gather-bound: one thread per output pixel (all channels), reads through L1/L2.
#include "rt.h"
#include <type_traits>
#include <map>
#include <memory>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>
#include <algorithm>
#include "warp8.h"
#include "resize_tab8.h"

using namespace mi355;

namespace {

enum { D8U = MI355CV_8U, D16U = MI355CV_16U, D16S = MI355CV_16S, D32F = MI355CV_32F };
__host__ __device__ inline int eszOf(int d) { return d == D8U ? 1 : d == D32F ? 4 : 2; }

__device__ __forceinline__ int cvFloorD(double v) { int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int satIntD(double v) { return v >= 2147483647.0 ? 2147483647 : v <= -2147483648.0 ? (int)-2147483648LL : (int)__double2int_rn(v); }
__device__ __forceinline__ int satShort(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
__device__ __forceinline__ int clipI(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

__device__ __forceinline__ float ldV(const uchar* p, int depth, int idx)
{
    switch (depth) { case D8U: return (float)p[idx]; case D16U: return (float)reinterpret_cast<const unsigned short*>(p)[idx];
                     case D16S: return (float)reinterpret_cast<const short*>(p)[idx]; default: return reinterpret_cast<const float*>(p)[idx]; }
}
__device__ __forceinline__ void stRound(uchar* p, int depth, int idx, float v)
{
    const float r = rintf(v);
    switch (depth) {
    case D8U:  p[idx] = (uchar)(int)fminf(fmaxf(r, 0.f), 255.f); break;
    case D16U: reinterpret_cast<unsigned short*>(p)[idx] = (unsigned short)(int)fminf(fmaxf(r, 0.f), 65535.f); break;
    case D16S: reinterpret_cast<short*>(p)[idx] = (short)(int)fminf(fmaxf(r, -32768.f), 32767.f); break;
    default:   reinterpret_cast<float*>(p)[idx] = v;
    }
}
__device__ __forceinline__ void copyPix(uchar* d, const uchar* s, int bytes) { for (int i = 0; i < bytes; i++) d[i] = s[i]; }

// ---------------------------------------------------------------------------------- resize
struct ResizeArgs { int sw, sh, dw, dh, depth, cn; double scale_x, scale_y, inv_x, inv_y; int mode /*0 nn,1 linear,2 area-as-linear,3 areafast*/; int isx, isy;
                    int nnExact, ifx, ifx0, ify, ify0;   /* INTER_NEAREST_EXACT: 16.16 steps and half-pixel offsets of resizeNN_bitexact (resize.cpp:1267-1289) */
                    size_t sframe, dframe; /* bytes between the frames of a batch (grid z = frame) */ };

__device__ __forceinline__ void linCoef(int d, double scale, double inv, int areaMode, int& s, float& f)
{
    if (!areaMode) { f = (float)((d + 0.5) * scale - 0.5); s = cvFloorF(f); f -= s; }
    else { s = cvFloorD(d * scale); f = (float)((d + 1) - (s + 1) * inv); f = f <= 0 ? 0.f : f - cvFloorF(f); }
}

__global__ __launch_bounds__(256) void k_resize(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, ResizeArgs a)
{
    src += (size_t)blockIdx.z * a.sframe; dst += (size_t)blockIdx.z * a.dframe;
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= a.dw || dy >= a.dh) return;
    const int e = eszOf(a.depth), cn = a.cn;
    uchar* D = dst + (size_t)dy * dstep;
    if (a.mode == 0) {
        int sy = a.nnExact ? (a.ify * dy + a.ify0) >> 16 : cvFloorD(dy * a.scale_y); sy = sy > a.sh - 1 ? a.sh - 1 : sy;
        int sx = a.nnExact ? (a.ifx * dx + a.ifx0) >> 16 : cvFloorD(dx * a.scale_x); sx = sx > a.sw - 1 ? a.sw - 1 : sx;
        copyPix(D + (size_t)dx * cn * e, src + (size_t)sy * sstep + (size_t)sx * cn * e, cn * e);
        return;
    }
    if (a.mode == 3) {
        const int area = a.isx * a.isy;
        const float scale = 1.f / area;
        const bool fast2 = a.isx == 2 && a.isy == 2 && (cn == 1 || cn == 3 || cn == 4) && a.depth != D32F;
        const int sy0 = dy * a.isy;
        const int wfull = sy0 + a.isy <= a.sh ? a.sw / a.isx : 0;
        for (int c = 0; c < cn; c++) {
            const int idx = dx * cn + c;
            if (sy0 >= a.sh) { stRound(D, a.depth, idx, 0.f); continue; }
            if (dx < wfull) {
                if (fast2) {
                    int s = 0;
                    for (int sy = 0; sy < 2; sy++) for (int sx = 0; sx < 2; sx++)
                        s += (int)ldV(src + (size_t)(sy0 + sy) * sstep, a.depth, (dx * 2 + sx) * cn + c);
                    s = (s + 2) >> 2;
                    if (a.depth == D8U) D[idx] = (uchar)s; else if (a.depth == D16U) reinterpret_cast<unsigned short*>(D)[idx] = (unsigned short)s;
                    else reinterpret_cast<short*>(D)[idx] = (short)s;
                } else if (a.depth == D8U) {
                    int s = 0;
                    for (int sy = 0; sy < a.isy; sy++) for (int sx = 0; sx < a.isx; sx++)
                        s += src[(size_t)(sy0 + sy) * sstep + (dx * a.isx + sx) * cn + c];
                    stRound(D, a.depth, idx, s * scale);
                } else if (a.depth == D32F && a.isx == 2 && a.isy == 2) {
                    const float* r0 = reinterpret_cast<const float*>(src + (size_t)sy0 * sstep);
                    const float* r1 = reinterpret_cast<const float*>(src + (size_t)(sy0 + 1) * sstep);
                    // ResizeAreaFastVec_SIMD_32f (resize.cpp:2856-2905) sums pairwise, (s00 + s01) + (s10 + s11), for 1 channel below the last multiple of its 4 lanes and
                    // for every 4-channel pixel; the scalar loop behind it (:3013-3025: the 1-channel tail, 2 and 3 channels) sums in order, ((s00 + s01) + s10) + s11
                    const float s00 = r0[(dx * 2) * cn + c], s01 = r0[(dx * 2 + 1) * cn + c], s10 = r1[(dx * 2) * cn + c], s11 = r1[(dx * 2 + 1) * cn + c];
                    const bool pairwise = cn == 4 || (cn == 1 && dx < (wfull / 4) * 4);
                    const float s = pairwise ? __fadd_rn(__fadd_rn(s00, s01), __fadd_rn(s10, s11)) : __fadd_rn(__fadd_rn(__fadd_rn(s00, s01), s10), s11);
                    reinterpret_cast<float*>(D)[idx] = __fmul_rn(s, 0.25f);
                } else {
                    float s = 0;
                    for (int sy = 0; sy < a.isy; sy++) for (int sx = 0; sx < a.isx; sx++)
                        s += ldV(src + (size_t)(sy0 + sy) * sstep, a.depth, (dx * a.isx + sx) * cn + c);
                    stRound(D, a.depth, idx, s * scale);
                }
            } else {
                float s = 0; int is = 0, count = 0;
                const int sx0 = dx * a.isx * cn + c;
                for (int sy = 0; sy < a.isy; sy++) {
                    if (sy0 + sy >= a.sh) break;
                    for (int sx = 0; sx < a.isx * cn; sx += cn) {
                        if (sx0 - c + sx >= a.sw * cn) break;
                        if (a.depth == D8U) is += src[(size_t)(sy0 + sy) * sstep + sx0 + sx];
                        else s += ldV(src + (size_t)(sy0 + sy) * sstep, a.depth, sx0 + sx);
                        count++;
                    }
                }
                if (count == 0) { stRound(D, a.depth, idx, 0.f); continue; }
                stRound(D, a.depth, idx, (a.depth == D8U ? (float)is : s) / count);
            }
        }
        return;
    }
    const int areaMode = a.mode == 2;
    int sy, sx; float fy, fx;
    linCoef(dy, a.scale_y, a.inv_y, areaMode, sy, fy);
    linCoef(dx, a.scale_x, a.inv_x, areaMode, sx, fx);
    const int y0 = clipI(sy, 0, a.sh), y1 = clipI(sy + 1, 0, a.sh);
    if (sx < 0) { fx = 0; sx = 0; }
    bool edge = false;
    if (sx + 1 >= a.sw) { edge = true; if (sx >= a.sw - 1) { fx = 0; sx = a.sw - 1; } }
    const float b0f = 1.f - fy, b1f = fy, a0f = 1.f - fx, a1f = fx;
    const uchar* r0 = src + (size_t)y0 * sstep;
    const uchar* r1 = src + (size_t)y1 * sstep;
    if (a.depth == D8U) {
        const int b0 = satShort(__float2int_rn(b0f * 2048)), b1 = satShort(__float2int_rn(b1f * 2048));
        const int a0 = satShort(__float2int_rn(a0f * 2048)), a1 = satShort(__float2int_rn(a1f * 2048));
        for (int c = 0; c < cn; c++) {
            const int i0 = sx * cn + c, i1 = i0 + cn;
            const int t0 = edge ? r0[i0] * 2048 : r0[i0] * a0 + r0[i1] * a1;
            const int t1 = edge ? r1[i0] * 2048 : r1[i0] * a0 + r1[i1] * a1;
            D[dx * cn + c] = (uchar)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2);
        }
        return;
    }
    for (int c = 0; c < cn; c++) {
        const int i0 = sx * cn + c, i1 = i0 + cn;
        const float p00 = ldV(r0, a.depth, i0), p10 = ldV(r1, a.depth, i0);
        float t0, t1;
        if (edge) { t0 = p00; t1 = p10; }
        else {
            const float p01 = ldV(r0, a.depth, i1), p11 = ldV(r1, a.depth, i1);
            t0 = __fadd_rn(__fmul_rn(p00, a0f), __fmul_rn(p01, a1f));
            t1 = __fadd_rn(__fmul_rn(p10, a0f), __fmul_rn(p11, a1f));
        }
        stRound(D, a.depth, dx * cn + c, __fadd_rn(__fmul_rn(t0, b0f), __fmul_rn(t1, b1f)));
    }
}

// INTER_AREA by exactly 2 x 2 on CV_8U (resizeAreaFast_, ResizeAreaFastVec_SIMD_8u, resize.cpp:2806-2900: (s00 + s01 + s10 + s11 + 2) >> 2), even
// source sizes: a lane produces FOUR destination pixels from 8 CN source bytes of each of the two source rows -- 8-byte loads, 4 CN-byte stores --
// instead of a thread per destination pixel with byte loads; HBM-bound (5 bytes per destination byte)
template <int CN>
__global__ __launch_bounds__(256) void k_area2x2_u8(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                    int dw, int dh)
{
    constexpr int NS = 2 * CN;                                     // source dwords per row and lane (8 CN bytes)
    const int g = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int ng = (dw + 3) >> 2;
    if (g >= ng || dy >= dh) return;
    const uchar* r0 = src + (size_t)blockIdx.z * sframe + (size_t)(2 * dy) * sstep + (size_t)g * 8 * CN;
    const uchar* r1 = r0 + sstep;
    uchar* D = dst + (size_t)blockIdx.z * dframe + (size_t)dy * dstep + (size_t)g * 4 * CN;
    if (4 * g + 4 <= dw) {
        uint32_t a[NS], b[NS];
#pragma unroll
        for (int i = 0; i < CN; i++) {
            const uint2 va = reinterpret_cast<const uint2*>(r0)[i], vb = reinterpret_cast<const uint2*>(r1)[i];
            a[2 * i] = va.x; a[2 * i + 1] = va.y; b[2 * i] = vb.x; b[2 * i + 1] = vb.y;
        }
        uint32_t o[CN] = {};
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int c = 0; c < CN; c++) {
                const int i0 = (2 * p) * CN + c, i1 = i0 + CN, ob = p * CN + c;
                const uint32_t s = ((a[i0 >> 2] >> (8 * (i0 & 3))) & 255u) + ((a[i1 >> 2] >> (8 * (i1 & 3))) & 255u) +
                                   ((b[i0 >> 2] >> (8 * (i0 & 3))) & 255u) + ((b[i1 >> 2] >> (8 * (i1 & 3))) & 255u);
                o[ob >> 2] |= ((s + 2) >> 2) << (8 * (ob & 3));
            }
#pragma unroll
        for (int i = 0; i < CN; i++) reinterpret_cast<uint32_t*>(D)[i] = o[i];
    } else {
        for (int p = 0; 4 * g + p < dw; p++)
            for (int c = 0; c < CN; c++) {
                const int i0 = (2 * p) * CN + c, i1 = i0 + CN;
                D[p * CN + c] = (uchar)((r0[i0] + r0[i1] + r1[i0] + r1[i1] + 2) >> 2);
            }
    }
}

// INTER_AREA by exactly 2 x 2 on CV_32FC1 (cfg3's 8K -> 4K; ResizeAreaFastVec_SIMD_32f resize.cpp:2928-2960: (s00 + s01) + (s10 + s11), times 0.25f), even
// source sizes: a lane produces FOUR destination floats from two dwordx4 loads of each of the two source rows and stores one dwordx4 -- the thread-per-pixel
// form ran at 54 % of HBM (its 8-byte loads and 4-byte stores leave the memory system half-used); 20 bytes per destination float
__global__ __launch_bounds__(256) void k_area2x2_f32(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                     int dw, int dh)
{
    const int g = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (4 * g >= dw || dy >= dh) return;
    const float* r0 = reinterpret_cast<const float*>(src + (size_t)blockIdx.z * sframe + (size_t)(2 * dy) * sstep) + 8 * (size_t)g;
    const float* r1 = reinterpret_cast<const float*>(reinterpret_cast<const uchar*>(r0) + sstep);
    float* D = reinterpret_cast<float*>(dst + (size_t)blockIdx.z * dframe + (size_t)dy * dstep) + 4 * (size_t)g;
    if (4 * g + 4 <= dw) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 a0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r0)), a1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r0) + 1);
        const f32x4 b0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r1)), b1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r1) + 1);
        f32x4 o;
        o.x = __fmul_rn(__fadd_rn(__fadd_rn(a0.x, a0.y), __fadd_rn(b0.x, b0.y)), 0.25f);
        o.y = __fmul_rn(__fadd_rn(__fadd_rn(a0.z, a0.w), __fadd_rn(b0.z, b0.w)), 0.25f);
        o.z = __fmul_rn(__fadd_rn(__fadd_rn(a1.x, a1.y), __fadd_rn(b1.x, b1.y)), 0.25f);
        o.w = __fmul_rn(__fadd_rn(__fadd_rn(a1.z, a1.w), __fadd_rn(b1.z, b1.w)), 0.25f);
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(D));
    } else {
        // the last dw % 4 columns are the reference's scalar tail (resize.cpp:3013-3025): ((s00 + s01) + s10) + s11, not the vector body's pairwise sums
        for (int p = 0; 4 * g + p < dw; p++) D[p] = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(r0[2 * p], r0[2 * p + 1]), r1[2 * p]), r1[2 * p + 1]), 0.25f);
    }
}

// resize, bilinear (and INTER_AREA upscaling, which is bilinear with other coefficients), single channel CV_32F / CV_8U: the
// arithmetic of k_resize specialised.  A thread owns a destination column for RROWS rows: its horizontal coefficient is
// computed once, both horizontal taps come from one unaligned 8-byte / 2-byte load per source row, and a wave walks down
// so that the source rows shared by consecutive destination rows are L1 hits.
constexpr int RROWS = 8;
template <typename T>
__global__ __launch_bounds__(256) void k_resize_lin1(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, ResizeArgs a)
{
    src += (size_t)blockIdx.z * a.sframe; dst += (size_t)blockIdx.z * a.dframe;
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yb = (blockIdx.y * 4 + (threadIdx.x >> 6)) * RROWS;
    if (dx >= a.dw || yb >= a.dh) return;
    const int areaMode = a.mode == 2;
    int sx; float fx;
    linCoef(dx, a.scale_x, a.inv_x, areaMode, sx, fx);
    if (sx < 0) { fx = 0; sx = 0; }
    bool edge = false;
    if (sx + 1 >= a.sw) { edge = true; if (sx >= a.sw - 1) { fx = 0; sx = a.sw - 1; } }
    const float a0f = 1.f - fx, a1f = fx;
    const int a0 = satShort(__float2int_rn(a0f * 2048)), a1 = satShort(__float2int_rn(a1f * 2048));
    const int xoff = (edge ? sx - 1 : sx);                  // the pair (xoff, xoff+1) is always inside the row when sw >= 2
    const int ye = min(yb + RROWS, a.dh);
    for (int dy = yb; dy < ye; dy++) {
        int sy; float fy;
        linCoef(dy, a.scale_y, a.inv_y, areaMode, sy, fy);
        const int y0 = clipI(sy, 0, a.sh), y1 = clipI(sy + 1, 0, a.sh);
        const float b0f = 1.f - fy, b1f = fy;
        const uchar* r0 = src + (size_t)y0 * sstep + (size_t)xoff * sizeof(T);
        const uchar* r1 = src + (size_t)y1 * sstep + (size_t)xoff * sizeof(T);
        if (sizeof(T) == 4) {
            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
            const f2u p0 = *reinterpret_cast<const f2u*>(r0), p1 = *reinterpret_cast<const f2u*>(r1);
            float t0, t1;
            if (edge) { t0 = p0.y; t1 = p1.y; }
            else { t0 = __fadd_rn(__fmul_rn(p0.x, a0f), __fmul_rn(p0.y, a1f)); t1 = __fadd_rn(__fmul_rn(p1.x, a0f), __fmul_rn(p1.y, a1f)); }
            reinterpret_cast<float*>(dst + (size_t)dy * dstep)[dx] = __fadd_rn(__fmul_rn(t0, b0f), __fmul_rn(t1, b1f));
        } else {
            typedef unsigned short u16u __attribute__((aligned(1)));
            const int q0 = *reinterpret_cast<const u16u*>(r0), q1 = *reinterpret_cast<const u16u*>(r1);
            const int b0 = satShort(__float2int_rn(b0f * 2048)), b1 = satShort(__float2int_rn(b1f * 2048));
            const int t0 = edge ? (q0 >> 8) * 2048 : (q0 & 255) * a0 + (q0 >> 8) * a1;
            const int t1 = edge ? (q1 >> 8) * 2048 : (q1 & 255) * a0 + (q1 >> 8) * a1;
            (dst + (size_t)dy * dstep)[dx] = (uchar)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

// the same for interleaved channels: a thread owns one destination PIXEL column; the 2*CN source elements of its two
// horizontal taps are adjacent in memory and come from one unaligned load per source row (8 bytes for 3 or 4 8-bit channels,
// CN float pairs otherwise); the last source columns, where that load would leave the row, read element by element
template <typename T, int CN>
__global__ __launch_bounds__(256) void k_resize_linC(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, ResizeArgs a)
{
    src += (size_t)blockIdx.z * a.sframe; dst += (size_t)blockIdx.z * a.dframe;
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yb = (blockIdx.y * 4 + (threadIdx.x >> 6)) * RROWS;
    if (dx >= a.dw || yb >= a.dh) return;
    const int areaMode = a.mode == 2;
    int sx; float fx;
    linCoef(dx, a.scale_x, a.inv_x, areaMode, sx, fx);
    if (sx < 0) { fx = 0; sx = 0; }
    bool edge = false;
    if (sx + 1 >= a.sw) { edge = true; if (sx >= a.sw - 1) { fx = 0; sx = a.sw - 1; } }
    const float a0f = 1.f - fx, a1f = fx;
    const int a0 = satShort(__float2int_rn(a0f * 2048)), a1 = satShort(__float2int_rn(a1f * 2048));
    const bool wide = sizeof(T) == 4 ? !edge : (sx * CN + 8 <= a.sw * CN);       // the 8-byte load stays inside the row
    const int ye = min(yb + RROWS, a.dh);
    for (int dy = yb; dy < ye; dy++) {
        int sy; float fy;
        linCoef(dy, a.scale_y, a.inv_y, areaMode, sy, fy);
        const int y0 = clipI(sy, 0, a.sh), y1 = clipI(sy + 1, 0, a.sh);
        const float b0f = 1.f - fy, b1f = fy;
        const uchar* r0 = src + (size_t)y0 * sstep + (size_t)sx * CN * sizeof(T);
        const uchar* r1 = src + (size_t)y1 * sstep + (size_t)sx * CN * sizeof(T);
        if (sizeof(T) == 4) {
            float* D = reinterpret_cast<float*>(dst + (size_t)dy * dstep) + (size_t)dx * CN;
#pragma unroll
            for (int c = 0; c < CN; c++) {
                const float p00 = reinterpret_cast<const float*>(r0)[c], p10 = reinterpret_cast<const float*>(r1)[c];
                float t0 = p00, t1 = p10;
                if (!edge) {
                    const float p01 = reinterpret_cast<const float*>(r0)[CN + c], p11 = reinterpret_cast<const float*>(r1)[CN + c];
                    t0 = __fadd_rn(__fmul_rn(p00, a0f), __fmul_rn(p01, a1f)); t1 = __fadd_rn(__fmul_rn(p10, a0f), __fmul_rn(p11, a1f));
                }
                D[c] = __fadd_rn(__fmul_rn(t0, b0f), __fmul_rn(t1, b1f));
            }
        } else {
            const int b0 = satShort(__float2int_rn(b0f * 2048)), b1 = satShort(__float2int_rn(b1f * 2048));
            unsigned long long q0 = 0, q1 = 0;
            if (wide) {
                typedef unsigned long long u64u __attribute__((aligned(1)));
                q0 = *reinterpret_cast<const u64u*>(r0); q1 = *reinterpret_cast<const u64u*>(r1);
            } else {
                const int nb = edge ? CN : 2 * CN;
                for (int k = 0; k < nb; k++) { q0 |= (unsigned long long)r0[k] << (8 * k); q1 |= (unsigned long long)r1[k] << (8 * k); }
            }
            uchar* D = dst + (size_t)dy * dstep + (size_t)dx * CN;
            unsigned pk = 0;
#pragma unroll
            for (int c = 0; c < CN; c++) {
                const int p00 = (int)((q0 >> (8 * c)) & 255), p01 = (int)((q0 >> (8 * (CN + c))) & 255);
                const int p10 = (int)((q1 >> (8 * c)) & 255), p11 = (int)((q1 >> (8 * (CN + c))) & 255);
                const int t0 = edge ? p00 * 2048 : p00 * a0 + p01 * a1;
                const int t1 = edge ? p10 * 2048 : p10 * a0 + p11 * a1;
                const unsigned r = (unsigned)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2) & 255u;
                pk |= r << (8 * c);
            }
            if (CN == 4) *reinterpret_cast<unsigned*>(D) = pk;
            else { D[0] = (uchar)pk; D[1] = (uchar)(pk >> 8); D[2] = (uchar)(pk >> 16); }
        }
    }
}

// true INTER_AREA (shrinking by non-integer ratios): computeResizeAreaTab resize.cpp:3334 + ResizeArea_Invoker :3181.  The two
// tap tables are built on the host exactly as the reference builds them (double arithmetic, float weights); per output element
// sum = b0*buf0, sum += bj*bufj with bufj = ((0 + S[k0]*a0) + S[k1]*a1) + ..., float multiply and add kept separate, then
// saturate_cast<T>.  Thread per output element; the taps of a 4K -> 720p shrink are 3-4 per axis.
struct AreaTap { int si; float alpha; };

template <typename T>
__global__ __launch_bounds__(256) void k_resize_area(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int dw, int dh, int cn, int depth,
                                                     const AreaTap* __restrict__ xt, const int* __restrict__ xo, const AreaTap* __restrict__ yt, const int* __restrict__ yo)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= dw * cn || dy >= dh) return;
    const int dx = e / cn, c = e - dx * cn;
    const int k0 = xo[dx], k1 = xo[dx + 1], j0 = yo[dy], j1 = yo[dy + 1];
    float sum = 0.f;
    for (int j = j0; j < j1; j++) {
        const T* S = reinterpret_cast<const T*>(src + (size_t)yt[j].si * sstep);
        float buf = 0.f;
        for (int k = k0; k < k1; k++) buf = __fadd_rn(buf, __fmul_rn((float)S[xt[k].si * cn + c], xt[k].alpha));
        const float t = __fmul_rn(yt[j].alpha, buf);
        sum = j == j0 ? t : __fadd_rn(sum, t);
    }
    stRound(dst + (size_t)dy * dstep, depth, e, sum);
}

int buildAreaTab(int ssize, int dsize, double scale, std::vector<AreaTap>& tab, std::vector<int>& ofs)
{
    tab.clear(); ofs.assign((size_t)dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
        ofs[(size_t)dx] = (int)tab.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cellWidth = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, (float)((sx1 - fsx1) / cellWidth)});
        for (int sx = sx1; sx < sx2; sx++) tab.push_back({sx, (float)(1.0 / cellWidth)});
        if (fsx2 - sx2 > 1e-3) tab.push_back({sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth)});
    }
    ofs[(size_t)dsize] = (int)tab.size();
    return (int)tab.size();
}

// INTER_CUBIC for CV_8U / CV_32F (resize.cpp: coefficient set-up :4097-4190, interpolateCubic :964, HResizeCubic :1993,
// VResizeCubic :2045).  The per-column / per-row taps are built on the host exactly as the reference builds them (float, A = -0.75;
// 8U: * 2048 rounded to short).  The reference's vertical pass has a SIMD body and a scalar tail that round differently; both are
// reproduced per element index: body = float S0*b0 + (S1*b1 + (S2*b2 + S3*b3)) (8U: taps * 2^-22, round half-even, saturate) for
// e < (dw*cn / 8) * 8 (8U) or (dw*cn / 4) * 4 (32F); tail = exact integers (sum + 2^21) >> 22 (8U) / left-to-right float (32F).
// CV_16U / CV_16S take the float path of CV_32F (HResizeCubic / HResizeLanczos4 <ushort | short, float, float>, resize.cpp:3890-3925): samples converted to
// float, vertical result through Cast<float, T> = cvRound + saturation.  Their vector bodies are 8 elements wide (VResizeCubicVec_32f16u / 32f16s :1444-1488,
// VResizeLanczos4Vec_32f16s :1562-1594: nested from the last row, like CV_32F's); CV_16U Lanczos runs the SSE4.1 routine on x86, which sums
// left to right like the scalar tail (resize.sse4_1.cpp:189-226), so its body and tail coincide.
template <typename T> __device__ __forceinline__ float ldE(const uchar* row, int idx) { return (float)reinterpret_cast<const T*>(row)[idx]; }
template <typename T> __device__ __forceinline__ void stE(uchar* row, int idx, float v)
{
    if constexpr (sizeof(T) == 4) reinterpret_cast<float*>(row)[idx] = v;
    else if constexpr (std::is_same<T, unsigned short>::value) { const int r = __float2int_rn(v); reinterpret_cast<unsigned short*>(row)[idx] = (unsigned short)(r < 0 ? 0 : r > 65535 ? 65535 : r); }
    else { const int r = __float2int_rn(v); reinterpret_cast<short*>(row)[idx] = (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r); }
}
template <typename T, int NT> __device__ __forceinline__ int vecBody(int width)      // elements the reference's nested (last row first) vector form covers
{
    if (NT == 8 && std::is_same<T, unsigned short>::value) return 0;
    return sizeof(T) == 4 ? (width / 4) * 4 : (width / 8) * 8;
}

typedef rt8::Tap<4> CubicTap;                                            // resize_tab8.h: {int s; float f[4]; short i[4];}

template <typename T>
__global__ __launch_bounds__(256) void k_resize_cubic(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int sw, int sh, int dw, int dh, int cn,
                                                      const CubicTap* __restrict__ xt, const CubicTap* __restrict__ yt)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int width = dw * cn;
    if (e >= width || dy >= dh) return;
    const int dx = e / cn, c = e - dx * cn;
    const CubicTap tx = xt[dx], ty = yt[dy];
    int xs[4];
#pragma unroll
    for (int j = 0; j < 4; j++) xs[j] = clipI(tx.s - 1 + j, 0, sw) * cn + c;
    if (sizeof(T) == 1) {
        int S[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uchar* R = src + (size_t)clipI(ty.s - 1 + k, 0, sh) * sstep;
            S[k] = R[xs[0]] * tx.i[0] + R[xs[1]] * tx.i[1] + R[xs[2]] * tx.i[2] + R[xs[3]] * tx.i[3];
        }
        int r;
        if (e < (width / 8) * 8) {
            const float sc = 1.f / (2048.f * 2048.f);
            float t = __fmul_rn((float)S[3], __fmul_rn((float)ty.i[3], sc));
            t = __fadd_rn(__fmul_rn((float)S[2], __fmul_rn((float)ty.i[2], sc)), t);
            t = __fadd_rn(__fmul_rn((float)S[1], __fmul_rn((float)ty.i[1], sc)), t);
            t = __fadd_rn(__fmul_rn((float)S[0], __fmul_rn((float)ty.i[0], sc)), t);
            r = __float2int_rn(t);
        } else
            r = (S[0] * ty.i[0] + S[1] * ty.i[1] + S[2] * ty.i[2] + S[3] * ty.i[3] + (1 << 21)) >> 22;
        (dst + (size_t)dy * dstep)[e] = (uchar)(r < 0 ? 0 : r > 255 ? 255 : r);
    } else {
        float S[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uchar* R = src + (size_t)clipI(ty.s - 1 + k, 0, sh) * sstep;
            float v = __fmul_rn(ldE<T>(R, xs[0]), tx.f[0]);
            v = __fadd_rn(v, __fmul_rn(ldE<T>(R, xs[1]), tx.f[1]));
            v = __fadd_rn(v, __fmul_rn(ldE<T>(R, xs[2]), tx.f[2]));
            v = __fadd_rn(v, __fmul_rn(ldE<T>(R, xs[3]), tx.f[3]));
            S[k] = v;
        }
        float r;
        if (e < vecBody<T, 4>(width)) {
            float t = __fmul_rn(S[3], ty.f[3]);
            t = __fadd_rn(__fmul_rn(S[2], ty.f[2]), t);
            t = __fadd_rn(__fmul_rn(S[1], ty.f[1]), t);
            r = __fadd_rn(__fmul_rn(S[0], ty.f[0]), t);
        } else {
            float t = __fmul_rn(S[0], ty.f[0]);
            t = __fadd_rn(t, __fmul_rn(S[1], ty.f[1]));
            t = __fadd_rn(t, __fmul_rn(S[2], ty.f[2]));
            r = __fadd_rn(t, __fmul_rn(S[3], ty.f[3]));
        }
        stE<T>(dst + (size_t)dy * dstep, e, r);
    }
}

using rt8::buildCubicTab;                                                // coefficients: resize_tab8.h

// INTER_LANCZOS4 (resize.cpp:974-1003 coefficients, :2066-2158 passes): 8 x 8 taps at s-3 .. s+4, clamped.  CV_8U is integer throughout
// (taps * 2048 as shorts, (sum + 2^21) >> 22); CV_32F sums the row left to right and the column as the reference's vector body does
// (S0*b0 + (S1*b1 + ( ... + S7*b7))) below the last multiple of four elements, left to right in its scalar tail.
typedef rt8::Tap<8> LanczosTap;

template <typename T>
__global__ __launch_bounds__(256) void k_resize_lanczos(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int sw, int sh, int dw, int dh, int cn,
                                                        const LanczosTap* __restrict__ xt, const LanczosTap* __restrict__ yt)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int width = dw * cn;
    if (e >= width || dy >= dh) return;
    const int dx = e / cn, c = e - dx * cn;
    const LanczosTap tx = xt[dx], ty = yt[dy];
    int xs[8];
#pragma unroll
    for (int j = 0; j < 8; j++) xs[j] = clipI(tx.s - 3 + j, 0, sw) * cn + c;
    if (sizeof(T) == 1) {
        int r = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uchar* R = src + (size_t)clipI(ty.s - 3 + k, 0, sh) * sstep;
            int v = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) v += R[xs[j]] * tx.i[j];
            r += v * ty.i[k];
        }
        r = (r + (1 << 21)) >> 22;
        (dst + (size_t)dy * dstep)[e] = (uchar)(r < 0 ? 0 : r > 255 ? 255 : r);
    } else {
        float S[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uchar* R = src + (size_t)clipI(ty.s - 3 + k, 0, sh) * sstep;
            float v = __fmul_rn(ldE<T>(R, xs[0]), tx.f[0]);
#pragma unroll
            for (int j = 1; j < 8; j++) v = __fadd_rn(v, __fmul_rn(ldE<T>(R, xs[j]), tx.f[j]));
            S[k] = v;
        }
        float r;
        if (e < vecBody<T, 8>(width)) {
            r = __fmul_rn(S[7], ty.f[7]);
#pragma unroll
            for (int k = 6; k >= 0; k--) r = __fadd_rn(__fmul_rn(S[k], ty.f[k]), r);
        } else {
            r = __fmul_rn(S[0], ty.f[0]);
#pragma unroll
            for (int k = 1; k < 8; k++) r = __fadd_rn(r, __fmul_rn(S[k], ty.f[k]));
        }
        stE<T>(dst + (size_t)dy * dstep, e, r);
    }
}

using rt8::buildLanczosTab;

// The separable form of the two kernels above for tiles of 64 x 16 output elements: the horizontal sums of every source row a tile
// needs are computed ONCE into LDS (they are exactly the S_k of the per-output kernels, same tap order), then each output combines NT of
// them vertically with the reference's body / tail formulas.  A 2x cubic upscale goes from 16 gathers + 20 MACs per output to about 3 + 7.
// The host checks that no tile needs more than RMAX source rows (strong minification does; it stays on the per-output kernels).
template <int NT> using TapT = rt8::Tap<NT>;                             // CubicTap (NT = 4) and LanczosTap (NT = 8)

template <typename T, int NT>
__global__ __launch_bounds__(256) void k_resize_tiled(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int sw, int sh, int dw, int dh, int cn,
                                                      const TapT<NT>* __restrict__ xt, const TapT<NT>* __restrict__ yt)
{
    constexpr int TW = 64, TH = 16, OFF = NT / 2 - 1;
    typedef typename std::conditional<sizeof(T) == 1, int, float>::type HT;
    extern __shared__ __attribute__((aligned(16))) uchar ldsRaw[];
    HT* H = reinterpret_cast<HT*>(ldsRaw);
    const int width = dw * cn;
    const int lx = threadIdx.x & 63, e = blockIdx.x * TW + lx;
    const int dy0 = blockIdx.y * TH, dyLast = min(dy0 + TH, dh) - 1;
    const int rmin = yt[dy0].s - OFF, R = yt[dyLast].s - OFF + NT - 1 - rmin + 1;
    if (e < width) {
        const int dx = e / cn, c = e - dx * cn;
        const TapT<NT> tx = xt[dx];
        int xs[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) xs[j] = clipI(tx.s - OFF + j, 0, sw) * cn + c;
        for (int r = threadIdx.x >> 6; r < R; r += 4) {
            const uchar* rowp = src + (size_t)clipI(rmin + r, 0, sh) * sstep;
            if (sizeof(T) == 1) {
                int v = 0;
#pragma unroll
                for (int j = 0; j < NT; j++) v += rt8::mul24((int)rowp[xs[j]], (int)tx.i[j]);      // a byte times a tap * 2048: 24-bit multiply (full rate)
                H[r * TW + lx] = (HT)v;
            } else {
                float v = __fmul_rn(ldE<T>(rowp, xs[0]), tx.f[0]);
#pragma unroll
                for (int j = 1; j < NT; j++) v = __fadd_rn(v, __fmul_rn(ldE<T>(rowp, xs[j]), tx.f[j]));
                H[r * TW + lx] = (HT)v;
            }
        }
    }
    __syncthreads();
    if (e >= width) return;
#pragma unroll
    for (int k4 = 0; k4 < TH / 4; k4++) {
        const int dy = dy0 + (threadIdx.x >> 6) + 4 * k4;
        if (dy >= dh) continue;
        const TapT<NT> ty = yt[dy];
        const HT* S = H + (ty.s - OFF - rmin) * TW + lx;                // S[k * TW] = horizontal sum of window row k
        if (sizeof(T) == 1) {
            int r;
            if (NT == 4 && e < (width / 8) * 8) {                       // VResizeCubicVec_32s8u: float, taps * 2^-22, nested from the last row
                const float sc = 1.f / (2048.f * 2048.f);
                float t = __fmul_rn((float)S[3 * TW], __fmul_rn((float)ty.i[3], sc));
                t = __fadd_rn(__fmul_rn((float)S[2 * TW], __fmul_rn((float)ty.i[2], sc)), t);
                t = __fadd_rn(__fmul_rn((float)S[1 * TW], __fmul_rn((float)ty.i[1], sc)), t);
                t = __fadd_rn(__fmul_rn((float)S[0], __fmul_rn((float)ty.i[0], sc)), t);
                r = __float2int_rn(t);
            } else {
                int acc = 0;
#pragma unroll
                for (int k = 0; k < NT; k++) acc += rt8::mul24((int)S[k * TW], (int)ty.i[k]);
                r = (acc + (1 << 21)) >> 22;
            }
            (dst + (size_t)dy * dstep)[e] = (uchar)(r < 0 ? 0 : r > 255 ? 255 : r);
        } else {
            float r;
            if (e < vecBody<T, NT>(width)) {
                r = __fmul_rn((float)S[(NT - 1) * TW], ty.f[NT - 1]);
#pragma unroll
                for (int k = NT - 2; k >= 0; k--) r = __fadd_rn(__fmul_rn((float)S[k * TW], ty.f[k]), r);
            } else {
                r = __fmul_rn((float)S[0], ty.f[0]);
#pragma unroll
                for (int k = 1; k < NT; k++) r = __fadd_rn(r, __fmul_rn((float)S[k * TW], ty.f[k]));
            }
            stE<T>(dst + (size_t)dy * dstep, e, r);
        }
    }
}

// CV_8U cubic / Lanczos on tiles of 256 x 16 elements, four per lane, source bytes staged in LDS (resize_tab8.h; `rows`: the most source rows any tile needs)
template <int NT>
__global__ __launch_bounds__(256) void k_resize_tab8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, rt8::Geom g,
                                                     const rt8::Tap<NT>* __restrict__ xt, const rt8::Tap<NT>* __restrict__ yt, int rows)
{
    extern __shared__ __attribute__((aligned(16))) uchar ldsRaw[];
    int* H = reinterpret_cast<int*>(ldsRaw);
    uchar* ldsSrc = ldsRaw + (size_t)rows * rt8::TW * 4;
    const rt8::Tile<NT> t = rt8::tileOf<NT>(g, blockIdx.x, blockIdx.y, xt, yt);
    rt8::HTaps<NT> ht; rt8::VTaps<NT> vt;
    rt8::loadTaps<NT>(threadIdx.x, g, t, xt, yt, ht, vt);
    rt8::stage<NT>(threadIdx.x, g, t, src, sstep, ldsSrc);
    __syncthreads();
    rt8::hpass<NT>(threadIdx.x, g, t, ht, ldsSrc, H);
    __syncthreads();
    rt8::vpass<NT>(threadIdx.x, g, t, vt, H, dst, dstep);
}

// INTER_LINEAR_EXACT (resize_bitExact<ET, interpolationLinear<ET>>, resize.cpp:789-950): per-axis tables of (offset, weight of the second tap in
// Q8 for 8U / Q16 for 16U and 16S) built on the host in the reference's arithmetic; horizontal pass exact in Q(shift), vertical pass a two-term dot
// product rounded once.  A weight of 0 means the second tap is not read (left of the image / from the last pixel on: the clamped cases).
struct ExactTap { int ofs; int c1; };
template <typename T, int SHIFT>
__global__ __launch_bounds__(256) void k_resize_exact(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int dw, int dh, int cn,
                                                      const ExactTap* __restrict__ tx, const ExactTap* __restrict__ ty)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= dw * cn || y >= dh) return;
    const int x = e / cn, c = e - x * cn;
    const ExactTap ax = tx[x], ay = ty[y];
    if constexpr (sizeof(T) == 1) {
        // CV_8U, Q8 weights: every term fits 24 bits (255 * 256 * 256 < 2^24), so the two passes are 24-bit multiply-adds at full rate
        // instead of 64-bit products (cv::ORB builds its pyramid with this kernel)
        int H[2] = {0, 0};
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (r == 1 && ay.c1 == 0) break;
            const uchar* row = src + (size_t)(ay.ofs + r) * sstep;
            const int b0 = rt8::mul24(ax.ofs, cn) + c;
            const int p0 = row[b0], p1 = ax.c1 ? (int)row[b0 + cn] : 0;
            H[r] = rt8::mul24(256 - ax.c1, p0) + rt8::mul24(ax.c1, p1);
        }
        const int v = rt8::mul24(256 - ay.c1, H[0]) + rt8::mul24(ay.c1, H[1]);
        const int r8 = (v + (1 << 15)) >> 16;
        dst[(size_t)y * dstep + e] = (uchar)(r8 > 255 ? 255 : r8);
        return;
    }
    const long long one = 1LL << SHIFT;
    long long H[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r == 1 && ay.c1 == 0) break;
        const T* row = reinterpret_cast<const T*>(src + (size_t)(ay.ofs + r) * sstep);
        const long long p0 = (long long)row[ax.ofs * cn + c];
        const long long p1 = ax.c1 ? (long long)row[(ax.ofs + 1) * cn + c] : 0;
        H[r] = (one - ax.c1) * p0 + (long long)ax.c1 * p1;
    }
    const long long v = (one - ay.c1) * H[0] + (long long)ay.c1 * H[1];
    const long long r = (v + (1LL << (2 * SHIFT - 1))) >> (2 * SHIFT);
    T* D = reinterpret_cast<T*>(dst + (size_t)y * dstep);
    if (sizeof(T) == 1) D[e] = (T)(r < 0 ? 0 : r > 255 ? 255 : r);
    else if (T(-1) > T(0)) D[e] = (T)(r < 0 ? 0 : r > 65535 ? 65535 : r);
    else D[e] = (T)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

// interpolationLinear<ET>::getCoeffs / getMinMax (resize.cpp:794-826); softdouble there, IEEE double here (same roundings: built with -ffp-contract=off)
void buildExactTaps(double inv_scale, int ssize, int dsize, int shift, std::vector<ExactTap>& t)
{
    t.assign((size_t)dsize, ExactTap{0, 0});
    const double scale = 1.0 / inv_scale;
    int minofst = 0, maxofst = dsize;
    for (int d = 0; d < dsize; d++) {
        const double fval = scale * ((double)d + 0.5) - 0.5;
        const int ival = (int)std::floor(fval);
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) { t[(size_t)d].ofs = ival; t[(size_t)d].c1 = (int)std::nearbyint((fval - (double)ival) * (double)(1 << shift)); }
            else { t[(size_t)d].ofs = ssize - 1; maxofst = std::min(maxofst, d); }
        } else minofst = std::max(minofst, d + 1);
    }
    for (int d = 0; d < dsize; d++) {
        if (d < minofst) t[(size_t)d] = ExactTap{0, 0};
        else if (d >= maxofst) t[(size_t)d] = ExactTap{ssize - 1, 0};
    }
}

// rows of LDS the tiled kernel needs for the tallest tile, or 0 when that exceeds what it may use
template <class Tab> int tiledRows(const std::vector<Tab>& yt, int nt)
{
    int mx = 0;
    const int dh = (int)yt.size();
    for (int dy0 = 0; dy0 < dh; dy0 += 16) mx = std::max(mx, yt[(size_t)std::min(dy0 + 16, dh) - 1].s - yt[(size_t)dy0].s + nt);
    static const bool off = getenv("MI355CV_RESIZE_TILED") && atoi(getenv("MI355CV_RESIZE_TILED")) == 0;
    return (mx <= 64 && !off) ? mx : 0;
}

// Tap tables depend on (interpolation, destination length, scale) only, and building one costs more host time than the kernel that uses
// it takes (Lanczos: two libm calls and eight divisions per entry, 0.2 ms for a 4K axis): keep the most recent ones resident in HBM.
// An entry stays alive while the cache or a hook that fetched it holds a reference: eviction only drops the cache's, and the last owner frees
// (hipFree waits for the kernels that still read the table), so a table cannot disappear between a fetch and the launch that uses it.
struct DevBlock { void* p; ~DevBlock() { if (p) (void)hipFree(p); } };
typedef std::shared_ptr<DevBlock> DevRef;
struct TabKey { int dev, kind, n; double scale;
                bool operator<(const TabKey& o) const { return dev != o.dev ? dev < o.dev : kind != o.kind ? kind < o.kind : n != o.n ? n < o.n : scale < o.scale; } };
struct TabEntry { DevRef mem; int rows; unsigned long long stamp; };
template <class Tab, class Build>
bool cachedTab(int kind, int n, double scale, int nt, Build build, const Tab** dev, int* tileRows, DevRef* keep)
{
    static std::mutex mu;
    static std::map<TabKey, TabEntry> cache;
    static unsigned long long clock = 0;
    std::lock_guard<std::mutex> lk(mu);
    const TabKey key{activeDevice(), kind, n, scale};
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() >= 32) {                                           // drop the least recently used table
            auto old = cache.begin();
            for (auto j = cache.begin(); j != cache.end(); ++j) if (j->second.stamp < old->second.stamp) old = j;
            cache.erase(old);
        }
        std::vector<Tab> host;
        build(n, scale, host);
        void* d = nullptr;
        if (hipMalloc(&d, host.size() * sizeof(Tab)) != hipSuccess) { (void)hipGetLastError(); return false; }
        DevRef mem(new DevBlock{d});
        if (hipMemcpy(d, host.data(), host.size() * sizeof(Tab), hipMemcpyHostToDevice) != hipSuccess) return false;
        it = cache.emplace(key, TabEntry{mem, tiledRows(host, nt), 0}).first;
    }
    it->second.stamp = ++clock;
    *dev = (const Tab*)it->second.mem->p; *tileRows = it->second.rows; *keep = it->second.mem;
    return true;
}

// the same residency for the INTER_AREA tables (taps + per-output offsets in one allocation), keyed by (source length, destination length, scale)
struct AreaDev { const AreaTap* tab; const int* ofs; DevRef keep; };
bool cachedAreaTab(int ssize, int dsize, double scale, AreaDev* out)
{
    struct Key { int dev, s, d; double sc;
                 bool operator<(const Key& o) const { return dev != o.dev ? dev < o.dev : s != o.s ? s < o.s : d != o.d ? d < o.d : sc < o.sc; } };
    struct Entry { DevRef mem; size_t ofsAt; unsigned long long stamp; };
    static std::mutex mu;
    static std::map<Key, Entry> cache;
    static unsigned long long clock = 0;
    std::lock_guard<std::mutex> lk(mu);
    const Key key{activeDevice(), ssize, dsize, scale};
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() >= 32) {
            auto old = cache.begin();
            for (auto j = cache.begin(); j != cache.end(); ++j) if (j->second.stamp < old->second.stamp) old = j;
            cache.erase(old);
        }
        std::vector<AreaTap> tab; std::vector<int> ofs;
        buildAreaTab(ssize, dsize, scale, tab, ofs);
        const size_t tabBytes = (tab.size() * sizeof(AreaTap) + 15) & ~(size_t)15, ofsBytes = ofs.size() * sizeof(int);
        void* d = nullptr;
        if (hipMalloc(&d, tabBytes + ofsBytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        DevRef mem(new DevBlock{d});
        if ((!tab.empty() && hipMemcpy(d, tab.data(), tab.size() * sizeof(AreaTap), hipMemcpyHostToDevice) != hipSuccess) ||
            hipMemcpy((uchar*)d + tabBytes, ofs.data(), ofsBytes, hipMemcpyHostToDevice) != hipSuccess) return false;
        it = cache.emplace(key, Entry{mem, tabBytes, 0}).first;
    }
    it->second.stamp = ++clock;
    out->tab = (const AreaTap*)it->second.mem->p; out->ofs = (const int*)((const uchar*)it->second.mem->p + it->second.ofsAt); out->keep = it->second.mem;
    return true;
}

// and for the INTER_LINEAR_EXACT tap tables, keyed by (source length, destination length, scale, fixed-point shift): a caller that resizes the same
// geometry frame after frame (cv::ORB's pyramid: fourteen tables per frame) finds them resident instead of building and uploading them per call
bool cachedExactTab(double inv_scale, int ssize, int dsize, int shift, const ExactTap** dev, DevRef* keep)
{
    struct Key { int dev, s, d, shift; double sc;
                 bool operator<(const Key& o) const { return dev != o.dev ? dev < o.dev : s != o.s ? s < o.s : d != o.d ? d < o.d : shift != o.shift ? shift < o.shift : sc < o.sc; } };
    struct Entry { DevRef mem; unsigned long long stamp; };
    static std::mutex mu;
    static std::map<Key, Entry> cache;
    static unsigned long long clock = 0;
    std::lock_guard<std::mutex> lk(mu);
    const Key key{activeDevice(), ssize, dsize, shift, inv_scale};
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() >= 96) {
            auto old = cache.begin();
            for (auto j = cache.begin(); j != cache.end(); ++j) if (j->second.stamp < old->second.stamp) old = j;
            cache.erase(old);
        }
        std::vector<ExactTap> tab;
        buildExactTaps(inv_scale, ssize, dsize, shift, tab);
        void* d = nullptr;
        if (hipMalloc(&d, tab.size() * sizeof(ExactTap)) != hipSuccess) { (void)hipGetLastError(); return false; }
        DevRef mem(new DevBlock{d});
        if (hipMemcpy(d, tab.data(), tab.size() * sizeof(ExactTap), hipMemcpyHostToDevice) != hipSuccess) return false;
        it = cache.emplace(key, Entry{mem, 0}).first;
    }
    it->second.stamp = ++clock;
    *dev = (const ExactTap*)it->second.mem->p; *keep = it->second.mem;
    return true;
}

// ---------------------------------------------------------------------------------- sampler
// Q15 bilinear table, generated exactly as initInterTab2D does -- including its fix-up loop, which for ksize == 2
// walks k1,k2 over {1,2} and therefore compares against (and may write into) the NEXT, not yet computed entry.
short g_tabHost[1024 * 4 + 16];
std::once_flag g_tabHostOnce;
constexpr int TAB_MAX_DEV = 16;
short* g_tabDevs[TAB_MAX_DEV];
std::mutex g_tabMu;

void buildTabHost()
{
    float t1[64];
    const float scale = 1.f / 32;
    for (int i = 0; i < 32; i++) { float x = i * scale; t1[2 * i] = 1.f - x; t1[2 * i + 1] = x; }
    memset(g_tabHost, 0, sizeof g_tabHost);
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            short* it = g_tabHost + (i * 32 + j) * 4;
            int isum = 0;
            for (int k1 = 0; k1 < 2; k1++) {
                const float vy = t1[i * 2 + k1];
                for (int k2 = 0; k2 < 2; k2++) {
                    const float v = vy * t1[j * 2 + k2];
                    int q = (int)lrintf(v * 32768);
                    it[k1 * 2 + k2] = (short)(q < -32768 ? -32768 : q > 32767 ? 32767 : q);
                    isum += it[k1 * 2 + k2];
                }
            }
            if (isum != 32768) {
                const int diff = isum - 32768;
                int Mk1 = 1, Mk2 = 1, mk1 = 1, mk2 = 1;
                for (int k1 = 1; k1 < 3; k1++)
                    for (int k2 = 1; k2 < 3; k2++) {
                        if (it[k1 * 2 + k2] < it[mk1 * 2 + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (it[k1 * 2 + k2] > it[Mk1 * 2 + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) it[Mk1 * 2 + Mk2] = (short)(it[Mk1 * 2 + Mk2] - diff);
                else it[mk1 * 2 + mk2] = (short)(it[mk1 * 2 + mk2] - diff);
            }
        }
}

// the table in the calling thread's device memory (one copy per device, made on first use)
const short* deviceTab()
{
    std::call_once(g_tabHostOnce, buildTabHost);
    const int dev = activeDevice();
    if (dev < 0 || dev >= TAB_MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_tabMu);
    if (!g_tabDevs[dev]) {
        short* d = nullptr;
        if (hipMalloc((void**)&d, 1024 * 4 * sizeof(short)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemcpy(d, g_tabHost, 1024 * 4 * sizeof(short), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
        g_tabDevs[dev] = d;
    }
    return g_tabDevs[dev];
}

// The weight tables of INTER_CUBIC / INTER_LANCZOS4 for the warps (initInterTab1D / initInterTab2D imgwarp.cpp:152-262): per-axis float taps for the 32
// fractions, and for CV_8U their 2-D products * 2^15 rounded to short, the sum of every entry forced to 2^15 by moving the difference onto the largest /
// smallest of the four central weights.  Built on the host once (the Lanczos taps take the host's double sin / cos, as the reference does), one copy per device.
struct TapTabs { short cubicI[1024 * 16]; short lanczosI[1024 * 64]; float cubic1[32 * 4]; float lanczos1[32 * 8]; };
TapTabs g_tapHost;
std::once_flag g_tapHostOnce;
TapTabs* g_tapDevs[TAB_MAX_DEV];

void tapCubic(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}
void tapLanczos(float x, float* c)
{
    const double s45 = 0.70710678118654752440084436210485;
    const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-7f) { for (int i = 0; i < 8; i++) c[i] = 0; c[3] = 1; return; }
    float sum = 0;
    const double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
    for (int i = 0; i < 8; i++) {
        const double y = -(x + 3 - i) * 3.1415926535897932384626433832795 * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) c[i] *= sum;
}
void buildTapHost()
{
    for (int m = 0; m < 2; m++) {
        const int ks = m ? 8 : 4;
        float* t1 = m ? g_tapHost.lanczos1 : g_tapHost.cubic1;
        short* ti = m ? g_tapHost.lanczosI : g_tapHost.cubicI;
        for (int i = 0; i < 32; i++) { if (m) tapLanczos(i * (1.f / 32), t1 + i * ks); else tapCubic(i * (1.f / 32), t1 + i * ks); }
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                short* it = ti + (i * 32 + j) * ks * ks;
                int isum = 0;
                for (int k1 = 0; k1 < ks; k1++)
                    for (int k2 = 0; k2 < ks; k2++) {
                        const float v = t1[i * ks + k1] * t1[j * ks + k2];
                        const int q = (int)lrintf(v * 32768);
                        it[k1 * ks + k2] = (short)(q < -32768 ? -32768 : q > 32767 ? 32767 : q);
                        isum += it[k1 * ks + k2];
                    }
                if (isum != 32768) {
                    const int diff = isum - 32768, k0 = ks / 2;
                    int Mk1 = k0, Mk2 = k0, mk1 = k0, mk2 = k0;
                    for (int k1 = k0; k1 < k0 + 2; k1++)
                        for (int k2 = k0; k2 < k0 + 2; k2++) {
                            if (it[k1 * ks + k2] < it[mk1 * ks + mk2]) { mk1 = k1; mk2 = k2; }
                            else if (it[k1 * ks + k2] > it[Mk1 * ks + Mk2]) { Mk1 = k1; Mk2 = k2; }
                        }
                    if (diff < 0) it[Mk1 * ks + Mk2] = (short)(it[Mk1 * ks + Mk2] - diff);
                    else it[mk1 * ks + mk2] = (short)(it[mk1 * ks + mk2] - diff);
                }
            }
    }
}
const TapTabs* deviceTapTabs()
{
    std::call_once(g_tapHostOnce, buildTapHost);
    const int dev = activeDevice();
    if (dev < 0 || dev >= TAB_MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_tabMu);
    if (!g_tapDevs[dev]) {
        TapTabs* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(TapTabs)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemcpy(d, &g_tapHost, sizeof(TapTabs), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
        g_tapDevs[dev] = d;
    }
    return g_tapDevs[dev];
}

struct SampleArgs { int sw, sh, depth, cn, linear, border; float cval[4]; double cvalD[4]; /* CV_64F images: saturate_cast<double>(borderValue) */ };

__device__ void samplePixel(const uchar* __restrict__ src, size_t sstep, uchar* D, const SampleArgs& a, int sx, int sy, int ax, int ay,
                            const short* __restrict__ tab)
{
    const int e = eszOf(a.depth), cn = a.cn;
    if (!a.linear) {
        if ((unsigned)sx < (unsigned)a.sw && (unsigned)sy < (unsigned)a.sh) { copyPix(D, src + (size_t)sy * sstep + (size_t)sx * cn * e, cn * e); return; }
        if (a.border == B_REPLICATE) { sx = clipI(sx, 0, a.sw); sy = clipI(sy, 0, a.sh); copyPix(D, src + (size_t)sy * sstep + (size_t)sx * cn * e, cn * e); return; }
        if (a.border == B_CONSTANT) { for (int k = 0; k < cn; k++) stRound(D, a.depth, k, a.cval[k & 3]); return; }
        if (a.border == B_TRANSPARENT) return;
        sx = mi355_borderInterpolate(sx, a.sw, a.border); sy = mi355_borderInterpolate(sy, a.sh, a.border);
        copyPix(D, src + (size_t)sy * sstep + (size_t)sx * cn * e, cn * e);
        return;
    }
    if (a.border == B_CONSTANT && (sx >= a.sw || sx + 1 < 0 || sy >= a.sh || sy + 1 < 0)) { for (int k = 0; k < cn; k++) stRound(D, a.depth, k, a.cval[k & 3]); return; }
    int x0, x1, y0, y1;
    if ((unsigned)sx < (unsigned)(a.sw - 1) && (unsigned)sy < (unsigned)(a.sh - 1)) { x0 = sx; x1 = sx + 1; y0 = sy; y1 = sy + 1; }
    else if (a.border == B_TRANSPARENT) {
        // remapBilinear imgwarp.cpp:786-815: a point on the source's last column / row is blended from the neighbours it has and rescaled by
        // (sum of all four weights) / (sum of the weights used); anything further out keeps the destination's previous contents
        if (!(sx >= 0 && sx <= a.sw - 1 && sy >= 0 && sy <= a.sh - 1)) return;
        const bool has1 = sx < a.sw - 1, has2 = sy < a.sh - 1, has3 = has1 && has2;
        const uchar* S = src + (size_t)sy * sstep;
        if (a.depth == D8U) {
            const short* w = tab + (ay * 32 + ax) * 4;
            int wTot = w[0];
            if (has1) wTot += w[1];
            if (has2) wTot += w[2];
            if (has3) wTot += w[3];
            if (wTot == 0) return;
            const int wIni = (int)w[0] + w[1] + w[2] + w[3];
            for (int k = 0; k < cn; k++) {
                int t0 = S[sx * cn + k] * w[0];
                if (has1) t0 += S[(sx + 1) * cn + k] * w[1];
                if (has2) t0 += S[sstep + sx * cn + k] * w[2];
                if (has3) t0 += S[sstep + (sx + 1) * cn + k] * w[3];
                t0 = (int)__fdiv_rn(__fmul_rn((float)t0, (float)wIni), (float)wTot);
                const int r = (t0 + (1 << 14)) >> 15;
                D[k] = (uchar)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
            return;
        }
        const float s32 = 1.f / 32, fx = ax * s32, fy = ay * s32;
        const float w0 = __fmul_rn(1.f - fy, 1.f - fx), w1 = __fmul_rn(1.f - fy, fx), w2 = __fmul_rn(fy, 1.f - fx), w3 = __fmul_rn(fy, fx);
        float wTot = w0;
        if (has1) wTot = __fadd_rn(wTot, w1);
        if (has2) wTot = __fadd_rn(wTot, w2);
        if (has3) wTot = __fadd_rn(wTot, w3);
        if (wTot == 0.f) return;
        const float wIni = __fadd_rn(__fadd_rn(__fadd_rn(w0, w1), w2), w3);
        for (int k = 0; k < cn; k++) {
            float t0 = __fmul_rn(ldV(S, a.depth, sx * cn + k), w0);
            if (has1) t0 = __fadd_rn(t0, __fmul_rn(ldV(S, a.depth, (sx + 1) * cn + k), w1));
            if (has2) t0 = __fadd_rn(t0, __fmul_rn(ldV(S + sstep, a.depth, sx * cn + k), w2));
            if (has3) t0 = __fadd_rn(t0, __fmul_rn(ldV(S + sstep, a.depth, (sx + 1) * cn + k), w3));
            stRound(D, a.depth, k, __fdiv_rn(__fmul_rn(t0, wIni), wTot));
        }
        return;
    }
    else if (a.border == B_REPLICATE) { x0 = clipI(sx, 0, a.sw); x1 = clipI(sx + 1, 0, a.sw); y0 = clipI(sy, 0, a.sh); y1 = clipI(sy + 1, 0, a.sh); }
    else { x0 = mi355_borderInterpolate(sx, a.sw, a.border); x1 = mi355_borderInterpolate(sx + 1, a.sw, a.border);
           y0 = mi355_borderInterpolate(sy, a.sh, a.border); y1 = mi355_borderInterpolate(sy + 1, a.sh, a.border); }
    const uchar* r0 = src + (size_t)(y0 < 0 ? 0 : y0) * sstep;
    const uchar* r1 = src + (size_t)(y1 < 0 ? 0 : y1) * sstep;
    if (a.depth == D8U) {
        const short* w = tab + (ay * 32 + ax) * 4;
        const int w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        for (int k = 0; k < cn; k++) {
            const int cv = (int)fminf(fmaxf(rintf(a.cval[k & 3]), 0.f), 255.f);
            const int v0 = (x0 >= 0 && y0 >= 0) ? r0[x0 * cn + k] : cv;
            const int v1 = (x1 >= 0 && y0 >= 0) ? r0[x1 * cn + k] : cv;
            const int v2 = (x0 >= 0 && y1 >= 0) ? r1[x0 * cn + k] : cv;
            const int v3 = (x1 >= 0 && y1 >= 0) ? r1[x1 * cn + k] : cv;
            const int r = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
            D[k] = (uchar)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
        return;
    }
    const float s32 = 1.f / 32;
    const float fx = ax * s32, fy = ay * s32;
    const float wy0 = 1.f - fy, wy1 = fy, wx0 = 1.f - fx, wx1 = fx;
    const float w0 = __fmul_rn(wy0, wx0), w1 = __fmul_rn(wy0, wx1), w2 = __fmul_rn(wy1, wx0), w3 = __fmul_rn(wy1, wx1);
    for (int k = 0; k < cn; k++) {
        float cv = a.cval[k & 3];
        if (a.depth == D16U) cv = fminf(fmaxf(rintf(cv), 0.f), 65535.f);
        else if (a.depth == D16S) cv = fminf(fmaxf(rintf(cv), -32768.f), 32767.f);
        const float v0 = (x0 >= 0 && y0 >= 0) ? ldV(r0, a.depth, x0 * cn + k) : cv;
        const float v1 = (x1 >= 0 && y0 >= 0) ? ldV(r0, a.depth, x1 * cn + k) : cv;
        const float v2 = (x0 >= 0 && y1 >= 0) ? ldV(r1, a.depth, x0 * cn + k) : cv;
        const float v3 = (x1 >= 0 && y1 >= 0) ? ldV(r1, a.depth, x1 * cn + k) : cv;
        float t = __fadd_rn(__fmul_rn(v0, w0), __fmul_rn(v1, w1));
        t = __fadd_rn(t, __fmul_rn(v2, w2));
        t = __fadd_rn(t, __fmul_rn(v3, w3));
        stRound(D, a.depth, k, t);
    }
}

struct WarpArgs { double M[9]; int dw, dh, kind /*0 affine, 1 perspective, 2 remap32f*/; int bw0; int band /* XCD-banded tile order */; int gx, gy;
                  int pexact; /* MI355CV_PERSP_EXACT=1: the IEEE division in every pixel (A/B of persp::xy) */
                  int rel; /* WARP_RELATIVE_MAP: the map holds offsets, the destination pixel's own (x, y) is added to the INTEGER source coordinates after their
                              saturation to short (imgwarp.cpp:354-359, :708-712: XY[dx*2] + _offset.x + dx) */
                  size_t sframe, dframe; /* bytes between the frames of a batch (grid z = frame) */ };

// Tile order.  Blocks are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8) and every XCD has its own L2, so with the plain
// x-fastest order the 8 horizontally adjacent tiles -- whose source footprints share cache lines -- land in 8 different L2s and each fetches
// the shared lines again.  Banded order hands XCD k the k-th horizontal band of the tile grid (x-fastest inside the band): neighbours in
// both directions then meet in one L2.
__device__ __forceinline__ void tileOf(const WarpArgs& w, int& tx, int& ty)
{
    if (!w.band) { tx = blockIdx.x; ty = blockIdx.y; return; }
    // bijection for any tile count: XCD k runs the blocks b = k, k + 8, ... -- q + (k < r) of them (q = total / 8, r = total % 8) -- and is
    // given that many consecutive tiles, starting after the tiles of XCDs 0..k-1
    const int b = blockIdx.y * gridDim.x + blockIdx.x, total = w.gx * w.gy;
    const int k = b & 7, q = total >> 3, r = total & 7;
    const int t = k * q + min(k, r) + (b >> 3);
    ty = t / w.gx; tx = t - ty * w.gx;
}


// ---- cv::warpPolar with WARP_INVERSE_MAP (imgwarp.cpp:3795-3845): the map of a destination pixel is (rho / Kmag, phi / Kangle + 1) with (rho, phi) from
// cv::cartToPolar of (x - cx, y - cy) and, for the semi-log form, rho <- cv::log(rho + 1).  Both are the reference's float approximations in the form its
// AVX2 build runs them (core mathfuncs_core.simd.hpp: cartToPolar32f_ :123-168 with v_atan_f32 :78-119, log32f :759-827 over the 256-entry table), fused
// multiply-adds where that build fuses; rows are cut into blocks of 1024 (mathfuncs.cpp:298) and a block of fewer than 16 (cartToPolar) / a row of fewer
// than 8 (log) elements takes the scalar form, whose association differs.
__device__ __forceinline__ float polarAtan(float y, float x, bool vec)
{
    const float k = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = __fmul_rn(0.9997878412794807f, k), p3 = __fmul_rn(-0.3258083974640975f, k), p5 = __fmul_rn(0.1555786518463281f, k), p7 = __fmul_rn(-0.04432655554792128f, k);
    const float eps = (float)2.2204460492503131e-16, scale = (float)(3.1415926535897932384626433832795 / 180);
    const float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (vec) {
        const float mn = ax < ay ? ax : ay, mx = ax > ay ? ax : ay;
        const float c = __fdiv_rn(mn, __fadd_rn(mx, eps)), cc = __fmul_rn(c, c);
        a = __fmul_rn(__fmaf_rn(__fmaf_rn(__fmaf_rn(cc, p7, p5), cc, p3), cc, p1), c);
        if (!(ax >= ay)) a = __fsub_rn(90.f, a);
    } else if (ax >= ay) {
        const float c = __fdiv_rn(ay, __fadd_rn(ax, eps)), c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1), c);
    } else {
        const float c = __fdiv_rn(ax, __fadd_rn(ay, eps)), c2 = __fmul_rn(c, c);
        a = __fmaf_rn(-(__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1)), c, 90.f);
    }
    if (x < 0.f) a = __fsub_rn(180.f, a);
    if (y < 0.f) a = __fsub_rn(360.f, a);
    return __fmul_rn(a, scale);
}
__device__ __forceinline__ float polarLog(float v, bool vec, const float* __restrict__ logTab /* 256 x (log(1 + i/256), 1 / (1 + i/256)) */)
{
    const uint32_t i0 = __float_as_uint(v);
    const float bf = __uint_as_float((i0 & ((1u << 15) - 1)) | (127u << 23));
    const int idx = (int)((i0 >> (23 - 8 - 1)) & (255 * 2));
    const float e = (float)((int)((i0 >> 23) & 0xff) - 127), ln2 = (float)0.69314718055994530941723212145818;
    const float A0 = 0.3333333333333333333333333f, A1 = -0.5f, A2 = 1.f, delta = idx == 510 ? -1.f / 512 : 0.f;
    const float y0 = __fmaf_rn(e, ln2, logTab[idx]);
    const float x0 = __fmaf_rn(__fsub_rn(bf, 1.f), logTab[idx + 1], delta);
    if (vec) { float z = __fmaf_rn(x0, A0, A1); z = __fmaf_rn(z, x0, A2); return __fmaf_rn(z, x0, y0); }
    return __fmaf_rn(__fmaf_rn(__fmaf_rn(A0, x0, A1), x0, A2), x0, y0);
}

// Integer source coordinates (sx, sy) and 5-bit fractions (ax, ay) of destination pixel (x, y) for every map form the warps and cv::remap take (the fixed-point
// coordinate generation of WarpAffineInvoker / WarpPerspectiveInvoker / RemapInvoker); s.linear: 0 nearest (no fractions), otherwise the 1/32 grid every
// interpolating sampler -- bilinear, bicubic, Lanczos -- shares
__device__ __forceinline__ void warpCoord(const SampleArgs& s, const WarpArgs& w, int x, int y, const uchar* __restrict__ mapx, size_t mxstep,
                                          const uchar* __restrict__ mapy, size_t mystep, int& sx, int& sy, int& ax, int& ay)
{
    int X, Y;
    if (w.kind == 0) {
        const int rd = s.linear ? 16 : 512;
        const int X0 = satIntD(__dmul_rn(__dadd_rn(__dmul_rn(w.M[1], (double)y), w.M[2]), 1024.0)) + rd;
        const int Y0 = satIntD(__dmul_rn(__dadd_rn(__dmul_rn(w.M[4], (double)y), w.M[5]), 1024.0)) + rd;
        const int ad = satIntD(__dmul_rn(__dmul_rn(w.M[0], (double)x), 1024.0));
        const int bd = satIntD(__dmul_rn(__dmul_rn(w.M[3], (double)x), 1024.0));
        if (s.linear) { X = (X0 + ad) >> 5; Y = (Y0 + bd) >> 5; }
        else { X = (X0 + ad) >> 10; Y = (Y0 + bd) >> 10; }
    } else if (w.kind == 1) {
        const int xb = (x / w.bw0) * w.bw0, x1 = x - xb;
        const double X0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[0], (double)xb), __dmul_rn(w.M[1], (double)y)), w.M[2]);
        const double Y0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[3], (double)xb), __dmul_rn(w.M[4], (double)y)), w.M[5]);
        const double W0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[6], (double)xb), __dmul_rn(w.M[7], (double)y)), w.M[8]);
        double W = __dadd_rn(W0, __dmul_rn(w.M[6], (double)x1));
        W = W != 0 ? __ddiv_rn(s.linear ? 32.0 : 1.0, W) : 0;
        double fX = __dmul_rn(__dadd_rn(X0, __dmul_rn(w.M[0], (double)x1)), W);
        double fY = __dmul_rn(__dadd_rn(Y0, __dmul_rn(w.M[3], (double)x1)), W);
        fX = fmax(-2147483648.0, fmin(2147483647.0, fX));
        fY = fmax(-2147483648.0, fmin(2147483647.0, fY));
        X = satIntD(fX); Y = satIntD(fY);
    } else if (w.kind == 7) {
        // inverse warpPolar: M = {cx, cy, Kmag, Kangle, semiLog}; mapx = the log table; the source is the polar image with one wrapped row above and below
        const float bx = __fsub_rn((float)x, (float)w.M[0]), by = __fsub_rn((float)y, (float)w.M[1]);
        const int blockLen = min(1024, w.dw - (x & ~1023));
        float rho = sqrtf(__fmaf_rn(bx, bx, __fmul_rn(by, by)));                   // (sqrtf is correctly rounded under hipcc's default; __fsqrt_rn maps to the native approximation)
        const float phi = polarAtan(by, bx, blockLen >= 16);
        if (w.M[4] != 0.0) rho = polarLog(__fadd_rn(rho, 1.f), w.dw >= 8, reinterpret_cast<const float*>(mapx));
        const float mx = (float)__ddiv_rn((double)rho, w.M[2]);
        const float my = __fadd_rn((float)__ddiv_rn((double)phi, w.M[3]), 1.f);
        if (s.linear) { X = satIntD((double)__fmul_rn(mx, 32.f)); Y = satIntD((double)__fmul_rn(my, 32.f)); }
        else { X = satIntD((double)mx); Y = satIntD((double)my); }
    } else if (w.kind == 2 || w.kind == 3 || w.kind == 6) {
        // float maps: kind 2 = two CV_32FC1 planes, kind 3 = one CV_32FC2 (RemapInvoker imgwarp.cpp:1233-1300), kind 6 = warpPolar's forward map
        // evaluated in place (imgwarp.cpp:3776-3792: (float)(rho[x] * cos(phi_y) + cx) in double, the per-row cos / sin and the per-column
        // radii come from the host, mapx = the radii, mapy = interleaved (cos, sin) doubles)
        float mx, my;
        if (w.kind == 2) {
            mx = reinterpret_cast<const float*>(mapx + (size_t)y * mxstep)[x];
            my = reinterpret_cast<const float*>(mapy + (size_t)y * mystep)[x];
        } else if (w.kind == 3) {
            const float2 m = reinterpret_cast<const float2*>(mapx + (size_t)y * mxstep)[x];
            mx = m.x; my = m.y;
        } else {
            const double rho = (double)reinterpret_cast<const float*>(mapx)[x];
            const double cp = reinterpret_cast<const double*>(mapy)[2 * y], sp = reinterpret_cast<const double*>(mapy)[2 * y + 1];
            mx = (float)__dadd_rn(__dmul_rn(rho, cp), w.M[0]);
            my = (float)__dadd_rn(__dmul_rn(rho, sp), w.M[1]);
        }
        if (s.linear) { X = satIntD((double)__fmul_rn(mx, 32.f)); Y = satIntD((double)__fmul_rn(my, 32.f)); }
        else { X = satIntD((double)mx); Y = satIntD((double)my); }
    } else {
        // fixed-point maps (kind 4: CV_16SC2 + CV_16UC1 / CV_16SC1 fractions, kind 5: CV_16SC2 alone, nearest): integer source coordinates and,
        // for bilinear, the index of the weight-table entry (ay * 32 + ax); nearest rounds with the fraction's halves (NNDeltaTab_i :237-238)
        const short2 xy = reinterpret_cast<const short2*>(mapx + (size_t)y * mxstep)[x];
        const int a = w.kind == 4 ? (int)(reinterpret_cast<const unsigned short*>(mapy + (size_t)y * mystep)[x] & 1023) : 0;
        const int rx = w.rel ? x : 0, ry = w.rel ? y : 0;
        if (s.linear) { sx = xy.x + rx; sy = xy.y + ry; ax = a & 31; ay = a >> 5; return; }
        const int dx = w.kind == 4 ? ((a & 31) < 16 ? 1 : 0) : 0, dy = w.kind == 4 ? ((a >> 5) < 16 ? 1 : 0) : 0;
        sx = (short)(xy.x + dx) + rx; sy = (short)(xy.y + dy) + ry; ax = ay = 0;
        return;
    }
    const int rx = w.rel ? x : 0, ry = w.rel ? y : 0;
    if (s.linear) { sx = satShort(X >> 5) + rx; sy = satShort(Y >> 5) + ry; ax = X & 31; ay = Y & 31; }
    else { sx = satShort(X) + rx; sy = satShort(Y) + ry; ax = ay = 0; }
}

// the same for the interpolating samplers with the affine terms of every destination column and row taken from a table built once per call (k_warp32_terms:
// colX[dw], colY[dw], rowX[dh], rowY[dh], the rows with the rounding term 16 added): two adds and two shifts per pixel instead of ~30 double-precision instructions
__device__ __forceinline__ void warpCoordT(const SampleArgs& s, const WarpArgs& w, const int* __restrict__ terms, int x, int y, const uchar* __restrict__ mapx, size_t mxstep,
                                           const uchar* __restrict__ mapy, size_t mystep, int& sx, int& sy, int& ax, int& ay)
{
    if (!terms) { warpCoord(s, w, x, y, mapx, mxstep, mapy, mystep, sx, sy, ax, ay); return; }
    const int X = (terms[2 * w.dw + y] + terms[x]) >> 5, Y = (terms[2 * w.dw + w.dh + y] + terms[w.dw + x]) >> 5;
    sx = satShort(X >> 5); sy = satShort(Y >> 5); ax = X & 31; ay = Y & 31;
}

__global__ __launch_bounds__(256) void k_warp(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
                                              SampleArgs s, WarpArgs w, const short* __restrict__ tab,
                                              const uchar* __restrict__ mapx, size_t mxstep, const uchar* __restrict__ mapy, size_t mystep)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w.dw || y >= w.dh) return;
    src += (size_t)blockIdx.z * w.sframe; dst += (size_t)blockIdx.z * w.dframe;
    uchar* D = dst + (size_t)y * dstep + (size_t)x * s.cn * eszOf(s.depth);
    int sx, sy, ax, ay;
    warpCoord(s, w, x, y, mapx, mxstep, mapy, mystep, sx, sy, ax, ay);
    samplePixel(src, sstep, D, s, sx, sy, ax, ay, tab);
}

// ---- INTER_CUBIC / INTER_LANCZOS4 in warpAffine / warpPerspective / remap (remapBicubic imgwarp.cpp:905-1010, remapLanczos4 :1013-1120).  KS x KS taps from
// (sx - KS/2 + 1, sy - KS/2 + 1); weights: CV_8U the Q15 table of initInterTab2D (tabI: [ay * 32 + ax][KS * KS] shorts, built on the host with the reference's
// sum fix-up), the other depths the float products ty[k1] * tx[k2] of the per-axis table (tab1: [fraction][KS]).  Inside the image each row's products are
// summed left to right and the rows added one after the other (the reference's float order; CV_8U is exact integer arithmetic and the order is free); next to
// the border sum = cval + SUM (S - cval) * w over the taps that exist.  BORDER_TRANSPARENT leaves pixels whose centre tap is outside untouched and reflects
// (REFLECT_101) for the rest.  One thread per destination pixel, all channels.
// CV_8U, all KS x KS taps inside the image: a row's taps are KS * CN contiguous bytes -- KS * CN / 4 unaligned dword loads --, a tap pair of one channel is one
// v_perm_b32 into two 16-bit halves and one v_dot2_i32_i16 with the table's weight pair (the Q15 entry read as dwordx4s): KS * KS VALU instructions per channel
// where the byte-wise form spends that many loads.  Exact integers, so the order of the sums is free.
template <int KS, int CN>
__device__ __forceinline__ void tapsInside8(const uchar* __restrict__ p /* first tap of the first row */, size_t sstep, uchar* D, const short* __restrict__ w)
{
    typedef uint32_t u32u __attribute__((aligned(1)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    constexpr int NB = KS * CN / 4, NW = KS * KS / 2;
    uint32_t px[KS][NB], wt[NW];
#pragma unroll
    for (int r = 0; r < KS; r++) {
        const u32u* q = reinterpret_cast<const u32u*>(p + (size_t)r * sstep);
#pragma unroll
        for (int i = 0; i < NB; i++) px[r][i] = q[i];
    }
    const uint4* wq = reinterpret_cast<const uint4*>(w);
#pragma unroll
    for (int i = 0; i < NW / 4; i++) { const uint4 v = wq[i]; wt[4 * i] = v.x; wt[4 * i + 1] = v.y; wt[4 * i + 2] = v.z; wt[4 * i + 3] = v.w; }
#pragma unroll
    for (int k = 0; k < CN; k++) {
        int sum = 1 << 14;
#pragma unroll
        for (int r = 0; r < KS; r++)
#pragma unroll
            for (int j = 0; j < KS / 2; j++) {
                constexpr int dummy = 0; (void)dummy;
                const int b0 = 2 * j * CN + k, b1 = (2 * j + 1) * CN + k;                 // byte positions of taps 2j, 2j+1 of channel k in the row
                const int d0 = b0 >> 2, d1 = b1 >> 2;
                // v_perm_b32(hi, lo, sel): selector bytes 0-3 pick from lo, 4-7 from hi, 0x0c = zero
                const uint32_t sel = (uint32_t)(b0 & 3) | (0x0cu << 8) | ((uint32_t)((d1 == d0 ? 0 : 4) + (b1 & 3)) << 16) | (0x0cu << 24);
                const uint32_t pair = __builtin_amdgcn_perm(px[r][d1], px[r][d0], sel);
                sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pair), __builtin_bit_cast(s16x2, wt[r * (KS / 2) + j]), sum, false);
            }
        const int v = sum >> 15;
        D[k] = (uchar)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

template <int KS>
__device__ __forceinline__ void samplePixelN(const uchar* __restrict__ src, size_t sstep, uchar* D, const SampleArgs& a, int sx, int sy, int ax, int ay,
                                             const short* __restrict__ tabI, const float* __restrict__ tab1)
{
    constexpr int OFF = KS / 2 - 1;
    sx -= OFF; sy -= OFF;
    const int cn = a.cn, depth = a.depth;
    const unsigned width1 = (unsigned)max(a.sw - (KS - 1), 0), height1 = (unsigned)max(a.sh - (KS - 1), 0);
    const bool inside = (unsigned)sx < width1 && (unsigned)sy < height1;
    int xi[KS], yi[KS];
    if (inside) {
#pragma unroll
        for (int i = 0; i < KS; i++) { xi[i] = sx + i; yi[i] = sy + i; }
    } else {
        if (a.border == B_TRANSPARENT && ((unsigned)(sx + OFF) >= (unsigned)a.sw || (unsigned)(sy + OFF) >= (unsigned)a.sh)) return;
        const int b1 = a.border != B_TRANSPARENT ? a.border : B_REFLECT_101;
        if (b1 == B_CONSTANT && (sx >= a.sw || sx + KS <= 0 || sy >= a.sh || sy + KS <= 0)) { for (int k = 0; k < cn; k++) stRound(D, depth, k, a.cval[k & 3]); return; }
#pragma unroll
        for (int i = 0; i < KS; i++) { xi[i] = mi355_borderInterpolate(sx + i, a.sw, b1); yi[i] = mi355_borderInterpolate(sy + i, a.sh, b1); }
    }
    if (depth == D8U) {
        const short* __restrict__ w = tabI + (ay * 32 + ax) * (KS * KS);
        if (inside && cn <= 4) {
            const uchar* p = src + (size_t)sy * sstep + (size_t)sx * cn;
            if (cn == 1) tapsInside8<KS, 1>(p, sstep, D, w);
            else if (cn == 2) tapsInside8<KS, 2>(p, sstep, D, w);
            else if (cn == 3) tapsInside8<KS, 3>(p, sstep, D, w);
            else tapsInside8<KS, 4>(p, sstep, D, w);
            return;
        }
        for (int k = 0; k < cn; k++) {
            int sum;
            if (inside) {
                sum = 0;
#pragma unroll
                for (int r = 0; r < KS; r++) {
                    const uchar* S = src + (size_t)yi[r] * sstep + k;
#pragma unroll
                    for (int c = 0; c < KS; c++) sum += (int)S[xi[c] * cn] * (int)w[r * KS + c];
                }
            } else {
                const int cv = (int)fminf(fmaxf(rintf(a.cval[k & 3]), 0.f), 255.f);
                sum = cv << 15;
#pragma unroll
                for (int r = 0; r < KS; r++) {
                    if (yi[r] < 0) continue;
                    const uchar* S = src + (size_t)yi[r] * sstep + k;
#pragma unroll
                    for (int c = 0; c < KS; c++) if (xi[c] >= 0) sum += ((int)S[xi[c] * cn] - cv) * (int)w[r * KS + c];
                }
            }
            const int v = (sum + (1 << 14)) >> 15;
            D[k] = (uchar)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
        return;
    }
    float wy[KS], wx[KS];
#pragma unroll
    for (int i = 0; i < KS; i++) { wy[i] = tab1[ay * KS + i]; wx[i] = tab1[ax * KS + i]; }
    for (int k = 0; k < cn; k++) {
        float sum;
        if (inside) {
            sum = 0.f;
#pragma unroll
            for (int r = 0; r < KS; r++) {
                const uchar* S = src + (size_t)yi[r] * sstep;
                float row = __fmul_rn(ldV(S, depth, xi[0] * cn + k), __fmul_rn(wy[r], wx[0]));
#pragma unroll
                for (int c = 1; c < KS; c++) row = __fadd_rn(row, __fmul_rn(ldV(S, depth, xi[c] * cn + k), __fmul_rn(wy[r], wx[c])));
                // remapBicubic starts from the first row's sum, remapLanczos4 from WT sum = 0 (0 + row: differs for a row sum of -0 only, and then in CV_32F's sign bit)
                sum = (r == 0 && KS == 4) ? row : __fadd_rn(sum, row);
            }
        } else {
            float cv = a.cval[k & 3];
            if (depth == D16U) cv = fminf(fmaxf(rintf(cv), 0.f), 65535.f);
            else if (depth == D16S) cv = fminf(fmaxf(rintf(cv), -32768.f), 32767.f);
            sum = cv;
#pragma unroll
            for (int r = 0; r < KS; r++) {
                if (yi[r] < 0) continue;
                const uchar* S = src + (size_t)yi[r] * sstep;
#pragma unroll
                for (int c = 0; c < KS; c++)
                    if (xi[c] >= 0) sum = __fadd_rn(sum, __fmul_rn(__fsub_rn(ldV(S, depth, xi[c] * cn + k), cv), __fmul_rn(wy[r], wx[c])));
            }
        }
        stRound(D, depth, k, sum);
    }
}

template <int KS>
__global__ __launch_bounds__(256) void k_warp_taps(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
                                                   SampleArgs s, WarpArgs w, const short* __restrict__ tabI, const float* __restrict__ tab1,
                                                   const uchar* __restrict__ mapx, size_t mxstep, const uchar* __restrict__ mapy, size_t mystep, int onlyOutside,
                                                   const int* __restrict__ terms)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w.dw || y >= w.dh) return;
    src += (size_t)blockIdx.z * w.sframe; dst += (size_t)blockIdx.z * w.dframe;
    uchar* D = dst + (size_t)y * dstep + (size_t)x * s.cn * eszOf(s.depth);
    int sx, sy, ax, ay;
    warpCoordT(s, w, terms, x, y, mapx, mxstep, mapy, mystep, sx, sy, ax, ay);
    // onlyOutside: the second launch behind k_warp_taps_lds, which leaves the pixels whose taps are not all inside the image
    if (onlyOutside && (unsigned)(sx - (KS / 2 - 1)) < (unsigned)max(s.sw - (KS - 1), 0) && (unsigned)(sy - (KS / 2 - 1)) < (unsigned)max(s.sh - (KS - 1), 0)) return;
    samplePixelN<KS>(src, sstep, D, s, sx, sy, ax, ay, tabI, tab1);
}

// ---- the same samplers with the per-pixel memory instructions cut down (round 4, profiles/r04_warp_taps.txt: k_warp_taps spends its time in the address
// unit -- every per-lane load instruction of a 4K frame costs ~4 us whatever its width --, 6 of them per bicubic CV_8UC1 pixel, 24 per CV_32FC1 pixel):
//   * the weights come from LDS: the per-axis float taps (32 x KS floats) and, for CV_8U, the whole Q15 table (32 KB bicubic, 128 KB Lanczos), copied in once by
//     workgroups that then walk many tiles;
//   * a tap row of a pixel inside the image is KS * CN contiguous elements: one to four 16-byte loads at element alignment, whatever the depth;
//   * pixels whose taps leave the image are left to a second launch of k_warp_taps (which then returns at once for all the others): their border arithmetic inlined
//     here cost every pixel its registers (166 instead of 67-100 VGPRs for the Lanczos forms).
// Results are those of k_warp_taps bit for bit: same products, same order of sums.
template <int N> __device__ __forceinline__ void loadRowDwords(uint32_t (&d)[N], const uchar* __restrict__ p)
{
    typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(1)));
    typedef uint32_t u2u __attribute__((ext_vector_type(2), aligned(1)));
    typedef uint32_t u1u __attribute__((aligned(1)));
#pragma unroll
    for (int i = 0; i + 4 <= N; i += 4) { const u4u v = *reinterpret_cast<const u4u*>(p + 4 * i); d[i] = v.x; d[i + 1] = v.y; d[i + 2] = v.z; d[i + 3] = v.w; }
    if constexpr ((N & 3) >= 2) { const u2u v = *reinterpret_cast<const u2u*>(p + 4 * (N & ~3)); d[N & ~3] = v.x; d[(N & ~3) + 1] = v.y; }
    if constexpr (N & 1) d[N - 1] = *reinterpret_cast<const u1u*>(p + 4 * (N - 1));
}
template <int DEPTH> __device__ __forceinline__ float tapElem(const uint32_t* d, int e)          // element e of a row held as dwords
{
    if constexpr (DEPTH == D32F) return __uint_as_float(d[e]);
    else if constexpr (DEPTH == D16U) return (float)((d[e >> 1] >> (16 * (e & 1))) & 0xffffu);
    else return (float)((int)(d[e >> 1] << (16 * (1 - (e & 1)))) >> 16);
}

template <int KS, int DEPTH, int CN, int BLOCK, int PP /* pixels per thread and tile; > 1 only with the affine term table */>
__global__ __launch_bounds__(BLOCK) void k_warp_taps_lds(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
                                                         SampleArgs s, WarpArgs w, const short* __restrict__ tabI, const float* __restrict__ tab1,
                                                         const uchar* __restrict__ mapx, size_t mxstep, const uchar* __restrict__ mapy, size_t mystep, int tilesX, int tilesY, int nframes,
                                                         const int* __restrict__ terms, uchar* __restrict__ stripFlag)
{
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    constexpr int ESZ = DEPTH == D8U ? 1 : DEPTH == D32F ? 4 : 2;
    constexpr int NB = KS * CN * ESZ / 4;                             // dwords of a tap row
    constexpr int ROWS = BLOCK / 64;                                  // tile: 64 x (ROWS * PP) destination pixels, PP per thread (rows wv, wv + ROWS, ...)
    extern __shared__ uint4 tapLds[];
    float* l1 = reinterpret_cast<float*>(tapLds);                     // [32][KS] per-axis taps
    uint4* l2 = tapLds + 32 * KS / 4;                                 // CV_8U: [1024][KS * KS / 8] Q15 weight pairs
    for (int i = threadIdx.x; i < 32 * KS / 4; i += BLOCK) tapLds[i] = reinterpret_cast<const uint4*>(tab1)[i];
    // a table entry is KS * KS / 8 uint4s; entries are ESTR uint4s apart (48 / 144 bytes): with the natural 32 / 128 bytes the 64 lanes' ds_read_b128s fell on 8 / 2
    // bank groups (SQ_LDS_BANK_CONFLICT 67 % / 88 % of the LDS cycles, profiles/r04_why_slow_taps.txt)
    constexpr int EU = KS * KS / 8, ESTR = EU + 1;
    if constexpr (DEPTH == D8U)
        for (int i = threadIdx.x; i < 1024 * EU; i += BLOCK) l2[(i / EU) * ESTR + (i % EU)] = reinterpret_cast<const uint4*>(tabI)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int perFrame = tilesX * tilesY, total = perFrame * nframes;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int f = t / perFrame, tt = t - f * perFrame, ty = tt / tilesX, tx = tt - ty * tilesX;
        const int x = tx * 64 + lane;
        const uchar* S = src + (size_t)f * w.sframe;
        int axA[PP], ayA[PP]; bool ok[PP];
        uint32_t px[PP][KS][NB];
        // the coordinates of the thread's pixels, then ALL their row loads (unconditional: a pixel that is not served here reads the frame's first bytes), then the
        // arithmetic: the loads of PP pixels are in flight together
#pragma unroll
        for (int q = 0; q < PP; q++) {
            const int y = (ty * PP + q) * ROWS + wv;
            const bool live = x < w.dw && y < w.dh;
            int sx, sy;
            if (PP > 1 || terms) {
                const int xc = min(x, w.dw - 1), yc = min(y, w.dh - 1);
                const int X = (terms[2 * w.dw + yc] + terms[xc]) >> 5, Y = (terms[2 * w.dw + w.dh + yc] + terms[w.dw + xc]) >> 5;
                sx = satShort(X >> 5); sy = satShort(Y >> 5); axA[q] = X & 31; ayA[q] = Y & 31;
            } else {
                sx = sy = 0; axA[q] = ayA[q] = 0;
                if (live) warpCoord(s, w, x, y, mapx, mxstep, mapy, mystep, sx, sy, axA[q], ayA[q]);
            }
            const int fx = sx - (KS / 2 - 1), fy = sy - (KS / 2 - 1);
            const bool inside = (unsigned)fx < (unsigned)max(s.sw - (KS - 1), 0) && (unsigned)fy < (unsigned)max(s.sh - (KS - 1), 0);
            ok[q] = live && inside;
            // a 64-pixel row strip with pixels whose taps leave the image is flagged for k_warp_taps_strips (which skips the strip's other pixels)
            if (__builtin_amdgcn_ballot_w64(live && !inside) != 0 && lane == 0) stripFlag[((size_t)f * w.dh + y) * tilesX + tx] = 1;
            const uchar* p = ok[q] ? S + (size_t)fy * sstep + (size_t)fx * (CN * ESZ) : S;
            const size_t rs = ok[q] ? sstep : 0;
#pragma unroll
            for (int r = 0; r < KS; r++) loadRowDwords<NB>(px[q][r], p + (size_t)r * rs);
        }
#pragma unroll
        for (int q = 0; q < PP; q++) {
            if (!ok[q]) continue;
            const int y = (ty * PP + q) * ROWS + wv, ax = axA[q], ay = ayA[q];
            uchar* D = dst + (size_t)f * w.dframe + (size_t)y * dstep + (size_t)x * (CN * ESZ);
            if constexpr (DEPTH == D8U) {
                uint32_t wt[KS * KS / 2];
                const uint4* wq = l2 + (ay * 32 + ax) * ESTR;
#pragma unroll
                for (int i = 0; i < KS * KS / 8; i++) { const uint4 v = wq[i]; wt[4 * i] = v.x; wt[4 * i + 1] = v.y; wt[4 * i + 2] = v.z; wt[4 * i + 3] = v.w; }
                uint32_t out = 0;
#pragma unroll
                for (int k = 0; k < CN; k++) {
                    int sum = 1 << 14;
#pragma unroll
                    for (int r = 0; r < KS; r++)
#pragma unroll
                        for (int j = 0; j < KS / 2; j++) {
                            const int b0 = 2 * j * CN + k, b1 = (2 * j + 1) * CN + k, d0 = b0 >> 2, d1 = b1 >> 2;
                            const uint32_t sel = (uint32_t)(b0 & 3) | (0x0cu << 8) | ((uint32_t)((d1 == d0 ? 0 : 4) + (b1 & 3)) << 16) | (0x0cu << 24);
                            sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, __builtin_amdgcn_perm(px[q][r][d1], px[q][r][d0], sel)),
                                                         __builtin_bit_cast(s16x2, wt[r * (KS / 2) + j]), sum, false);
                        }
                    int v = sum >> 15;
                    // (kept opaque: left to itself the compiler folds shift + clamp + pack of two channels into v_ashr_pk_u8_i32, whose upper 16 result bits it then ORs
                    // the other channels into as if they were zero -- on the MI355X they are not: CV_8UC4 results came back with stray bits in byte 2, GPU call r04f)
                    asm volatile("" : "+v"(v));
                    out |= (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v) << (8 * k);
                }
                if constexpr (CN == 1) D[0] = (uchar)out;
                else if constexpr (CN == 4) *reinterpret_cast<uint32_t*>(D) = out;
                else { D[0] = (uchar)out; D[1] = (uchar)(out >> 8); D[2] = (uchar)(out >> 16); }
            } else {
                float wy[KS], wx[KS];
#pragma unroll
                for (int i = 0; i < KS; i += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(l1 + ay * KS + i), b = *reinterpret_cast<const float4*>(l1 + ax * KS + i);
                    wy[i] = a.x; wy[i + 1] = a.y; wy[i + 2] = a.z; wy[i + 3] = a.w; wx[i] = b.x; wx[i + 1] = b.y; wx[i + 2] = b.z; wx[i + 3] = b.w;
                }
#pragma unroll
                for (int k = 0; k < CN; k++) {
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < KS; r++) {
                        float row = __fmul_rn(tapElem<DEPTH>(px[q][r], k), __fmul_rn(wy[r], wx[0]));
#pragma unroll
                        for (int c = 1; c < KS; c++) row = __fadd_rn(row, __fmul_rn(tapElem<DEPTH>(px[q][r], c * CN + k), __fmul_rn(wy[r], wx[c])));
                        sum = (r == 0 && KS == 4) ? row : __fadd_rn(sum, row);
                    }
                    stRound(D, DEPTH, k, sum);
                }
            }
        }
    }
}

// the row strips k_warp_taps_lds flagged: a workgroup looks at 16 consecutive strips (four per wave), a flagged strip's pixels with all taps inside are skipped.
// (A launch of one thread per pixel that returns at once for the served ones cost 13 us per 4K frame in workgroup dispatch alone, a quarter of the bicubic warp; a
// list of strips worked off by a fixed grid serialised the slow sampler in too few waves -- 200 us: profiles/r04_warp_taps.txt.)
template <int KS>
__global__ __launch_bounds__(256) void k_warp_taps_strips(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
                                                          SampleArgs s, WarpArgs w, const short* __restrict__ tabI, const float* __restrict__ tab1,
                                                          const uchar* __restrict__ mapx, size_t mxstep, const uchar* __restrict__ mapy, size_t mystep, int tilesX, int nframes,
                                                          const int* __restrict__ terms, const uchar* __restrict__ stripFlag)
{
    const int lane = threadIdx.x & 63;
    const size_t strips = (size_t)tilesX * w.dh * nframes;
    const size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
        const size_t sid = base + j;
        if (sid >= strips) return;
        if (!__builtin_amdgcn_readfirstlane((int)stripFlag[sid])) continue;
        const int tx = (int)(sid % (size_t)tilesX); const size_t r = sid / (size_t)tilesX; const int y = (int)(r % (size_t)w.dh), f = (int)(r / (size_t)w.dh);
        const int x = tx * 64 + lane;
        if (x >= w.dw) continue;
        int sx, sy, ax, ay;
        warpCoordT(s, w, terms, x, y, mapx, mxstep, mapy, mystep, sx, sy, ax, ay);
        if ((unsigned)(sx - (KS / 2 - 1)) < (unsigned)max(s.sw - (KS - 1), 0) && (unsigned)(sy - (KS / 2 - 1)) < (unsigned)max(s.sh - (KS - 1), 0)) continue;
        samplePixelN<KS>(src + (size_t)f * w.sframe, sstep, dst + (size_t)f * w.dframe + (size_t)y * dstep + (size_t)x * s.cn * eszOf(s.depth), s, sx, sy, ax, ay, tabI, tab1);
    }
}

// ---- CV_64F images in warpAffine / warpPerspective / remap (remapNearest<double>, remapBilinear<Cast<double, double>, RemapNoVec, float>, remapBicubic / remapLanczos4
// <Cast<double, double>, float, 1>: imgwarp.cpp:1736-1790): the weights are the float tables, the products and sums double, in the reference's order; one thread per pixel.
__device__ void samplePixel64(const uchar* __restrict__ src, size_t sstep, double* D, const SampleArgs& a, int sx, int sy, int ax, int ay,
                              const float* __restrict__ tabC, const float* __restrict__ tabL)
{
    const int cn = a.cn, mode = a.linear;
    auto S = [&](int y, int x, int k) { return reinterpret_cast<const double*>(src + (size_t)y * sstep)[x * cn + k]; };
    if (mode == 0) {
        if (!((unsigned)sx < (unsigned)a.sw && (unsigned)sy < (unsigned)a.sh)) {
            if (a.border == B_REPLICATE) { sx = clipI(sx, 0, a.sw); sy = clipI(sy, 0, a.sh); }
            else if (a.border == B_CONSTANT) { for (int k = 0; k < cn; k++) D[k] = a.cvalD[k & 3]; return; }
            else if (a.border == B_TRANSPARENT) return;
            else { sx = mi355_borderInterpolate(sx, a.sw, a.border); sy = mi355_borderInterpolate(sy, a.sh, a.border); }
        }
        for (int k = 0; k < cn; k++) D[k] = S(sy, sx, k);
        return;
    }
    if (mode == 1) {
        const float s32 = 1.f / 32, fx = ax * s32, fy = ay * s32;
        const float w[4] = {__fmul_rn(1.f - fy, 1.f - fx), __fmul_rn(1.f - fy, fx), __fmul_rn(fy, 1.f - fx), __fmul_rn(fy, fx)};
        if (a.border == B_CONSTANT && (sx >= a.sw || sx + 1 < 0 || sy >= a.sh || sy + 1 < 0)) { for (int k = 0; k < cn; k++) D[k] = a.cvalD[k & 3]; return; }
        int x0, x1, y0, y1;
        if ((unsigned)sx < (unsigned)(a.sw - 1) && (unsigned)sy < (unsigned)(a.sh - 1)) { x0 = sx; x1 = sx + 1; y0 = sy; y1 = sy + 1; }
        else if (a.border == B_TRANSPARENT) {
            if (!(sx >= 0 && sx <= a.sw - 1 && sy >= 0 && sy <= a.sh - 1)) return;
            const bool has1 = sx < a.sw - 1, has2 = sy < a.sh - 1, has3 = has1 && has2;
            double wTot = 0; wTot = __dadd_rn(wTot, (double)w[0]);
            if (has1) wTot = __dadd_rn(wTot, (double)w[1]);
            if (has2) wTot = __dadd_rn(wTot, (double)w[2]);
            if (has3) wTot = __dadd_rn(wTot, (double)w[3]);
            if (wTot == 0.0) return;
            const double wIni = __dadd_rn(__dadd_rn(__dadd_rn((double)w[0], (double)w[1]), (double)w[2]), (double)w[3]);
            for (int k = 0; k < cn; k++) {
                double t0 = 0; t0 = __dadd_rn(t0, __dmul_rn(S(sy, sx, k), (double)w[0]));
                if (has1) t0 = __dadd_rn(t0, __dmul_rn(S(sy, sx + 1, k), (double)w[1]));
                if (has2) t0 = __dadd_rn(t0, __dmul_rn(S(sy + 1, sx, k), (double)w[2]));
                if (has3) t0 = __dadd_rn(t0, __dmul_rn(S(sy + 1, sx + 1, k), (double)w[3]));
                D[k] = __ddiv_rn(__dmul_rn(t0, (double)(float)wIni), wTot);
            }
            return;
        }
        else if (a.border == B_REPLICATE) { x0 = clipI(sx, 0, a.sw); x1 = clipI(sx + 1, 0, a.sw); y0 = clipI(sy, 0, a.sh); y1 = clipI(sy + 1, 0, a.sh); }
        else { x0 = mi355_borderInterpolate(sx, a.sw, a.border); x1 = mi355_borderInterpolate(sx + 1, a.sw, a.border);
               y0 = mi355_borderInterpolate(sy, a.sh, a.border); y1 = mi355_borderInterpolate(sy + 1, a.sh, a.border); }
        for (int k = 0; k < cn; k++) {
            const double cv = a.cvalD[k & 3];
            const double v0 = (x0 >= 0 && y0 >= 0) ? S(y0, x0, k) : cv, v1 = (x1 >= 0 && y0 >= 0) ? S(y0, x1, k) : cv;
            const double v2 = (x0 >= 0 && y1 >= 0) ? S(y1, x0, k) : cv, v3 = (x1 >= 0 && y1 >= 0) ? S(y1, x1, k) : cv;
            double t = __dadd_rn(__dmul_rn(v0, (double)w[0]), __dmul_rn(v1, (double)w[1]));
            t = __dadd_rn(t, __dmul_rn(v2, (double)w[2]));
            D[k] = __dadd_rn(t, __dmul_rn(v3, (double)w[3]));
        }
        return;
    }
    const int KS = mode == 4 ? 8 : 4, OFF = KS / 2 - 1;
    const float* __restrict__ t1 = mode == 4 ? tabL : tabC;
    sx -= OFF; sy -= OFF;
    const bool inside = (unsigned)sx < (unsigned)max(a.sw - (KS - 1), 0) && (unsigned)sy < (unsigned)max(a.sh - (KS - 1), 0);
    int xi[8], yi[8];
    if (!inside) {
        if (a.border == B_TRANSPARENT && ((unsigned)(sx + OFF) >= (unsigned)a.sw || (unsigned)(sy + OFF) >= (unsigned)a.sh)) return;
        const int b1 = a.border != B_TRANSPARENT ? a.border : B_REFLECT_101;
        if (b1 == B_CONSTANT && (sx >= a.sw || sx + KS <= 0 || sy >= a.sh || sy + KS <= 0)) { for (int k = 0; k < cn; k++) D[k] = a.cvalD[k & 3]; return; }
        for (int i = 0; i < KS; i++) { xi[i] = mi355_borderInterpolate(sx + i, a.sw, b1); yi[i] = mi355_borderInterpolate(sy + i, a.sh, b1); }
    }
    for (int k = 0; k < cn; k++) {
        double sum;
        if (inside) {
            sum = 0;
            for (int r = 0; r < KS; r++) {
                double row = __dmul_rn(S(sy + r, sx, k), (double)__fmul_rn(t1[ay * KS + r], t1[ax * KS]));
                for (int c = 1; c < KS; c++) row = __dadd_rn(row, __dmul_rn(S(sy + r, sx + c, k), (double)__fmul_rn(t1[ay * KS + r], t1[ax * KS + c])));
                sum = (r == 0 && KS == 4) ? row : __dadd_rn(sum, row);
            }
        } else {
            const double cv = a.cvalD[k & 3];
            sum = cv;
            for (int r = 0; r < KS; r++) {
                if (yi[r] < 0) continue;
                for (int c = 0; c < KS; c++)
                    if (xi[c] >= 0) sum = __dadd_rn(sum, __dmul_rn(__dsub_rn(S(yi[r], xi[c], k), cv), (double)__fmul_rn(t1[ay * KS + r], t1[ax * KS + c])));
            }
        }
        D[k] = sum;
    }
}

__global__ __launch_bounds__(256) void k_warp64(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, SampleArgs s, WarpArgs w,
                                                const float* __restrict__ tabC, const float* __restrict__ tabL,
                                                const uchar* __restrict__ mapx, size_t mxstep, const uchar* __restrict__ mapy, size_t mystep)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w.dw || y >= w.dh) return;
    src += (size_t)blockIdx.z * w.sframe; dst += (size_t)blockIdx.z * w.dframe;
    int sx, sy, ax, ay;
    warpCoord(s, w, x, y, mapx, mxstep, mapy, mystep, sx, sy, ax, ay);
    samplePixel64(src, sstep, reinterpret_cast<double*>(dst + (size_t)y * dstep) + (size_t)x * s.cn, s, sx, sy, ax, ay, tabC, tabL);
}

// cv::convertMaps, float -> fixed point (imgwarp.cpp:2017-2120): ix = cvRound(x * 32), dst1 = (ix >> 5, iy >> 5) saturated to short,
// dst2 = (iy & 31) * 32 + (ix & 31); with nninterpolate dst1 = the rounded coordinates and there is no dst2
__global__ __launch_bounds__(256) void k_convert_maps_to_fixed(const uchar* __restrict__ m1, size_t m1step, const uchar* __restrict__ m2, size_t m2step, int interleaved,
                                                               uchar* __restrict__ d1, size_t d1step, uchar* __restrict__ d2, size_t d2step, int w, int h, int nn)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    float fx, fy;
    if (interleaved) { const float2 m = reinterpret_cast<const float2*>(m1 + (size_t)y * m1step)[x]; fx = m.x; fy = m.y; }
    else { fx = reinterpret_cast<const float*>(m1 + (size_t)y * m1step)[x]; fy = reinterpret_cast<const float*>(m2 + (size_t)y * m2step)[x]; }
    short2 o;
    if (nn) { o.x = (short)satShort(satIntD((double)fx)); o.y = (short)satShort(satIntD((double)fy)); }
    else {
        const int ix = satIntD((double)__fmul_rn(fx, 32.f)), iy = satIntD((double)__fmul_rn(fy, 32.f));
        o.x = (short)satShort(ix >> 5); o.y = (short)satShort(iy >> 5);
        reinterpret_cast<unsigned short*>(d2 + (size_t)y * d2step)[x] = (unsigned short)((iy & 31) * 32 + (ix & 31));
    }
    reinterpret_cast<short2*>(d1 + (size_t)y * d1step)[x] = o;
}

// ... and back (imgwarp.cpp:2122-2200): x = X + (fxy & 31) / 32, y = Y + (fxy >> 5) / 32 in float, one multiply-add each (the reference's
// v_muladd contracts on FMA builds and its scalar tail does not: the products are exact in float, so both give the same value)
__global__ __launch_bounds__(256) void k_convert_maps_to_float(const uchar* __restrict__ m1, size_t m1step, const uchar* __restrict__ m2, size_t m2step,
                                                               uchar* __restrict__ d1, size_t d1step, uchar* __restrict__ d2, size_t d2step, int interleaved, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const short2 xy = reinterpret_cast<const short2*>(m1 + (size_t)y * m1step)[x];
    const int fxy = m2 ? (int)(reinterpret_cast<const unsigned short*>(m2 + (size_t)y * m2step)[x] & 1023) : 0;
    const float scale = 1.f / 32;
    const float fx = __fadd_rn((float)xy.x, __fmul_rn((float)(fxy & 31), scale)), fy = __fadd_rn((float)xy.y, __fmul_rn((float)(fxy >> 5), scale));
    if (interleaved) reinterpret_cast<float2*>(d1 + (size_t)y * d1step)[x] = make_float2(fx, fy);
    else { reinterpret_cast<float*>(d1 + (size_t)y * d1step)[x] = fx; reinterpret_cast<float*>(d2 + (size_t)y * d2step)[x] = fy; }
}

// warpAffine, bilinear, single channel CV_32F / CV_8U: same arithmetic as k_warp + samplePixel, specialised so that the
// inner loop is straight-line code.  A thread owns one destination column for WROWS consecutive rows (its column terms
// sat_int(M0*x*1024), sat_int(M3*x*1024) are computed once); the two horizontally adjacent taps come from ONE unaligned
// 8-byte (32F) / 2-byte (8U) load per source row; pixels whose 2x2 footprint is not strictly inside the source take the
// generic sampler.  A wave walks down WROWS rows, so the source lines it touched for one row are in L1 for the next.
// rows per thread: 8 for CV_32F, 4 for CV_8U (measured, tools/warp_probe.py: 8-bit sources 39 / 25 / 31 us for C3 / C1 / C4 at 4 rows against 45 / 28 / 40
// at 8 and 61 / 38 / 56 at 16; CV_32F 63.5 us at 8, 65.9 at 4, 72.8 at 16)
// ---- warpPerspective coordinates without the IEEE division in (almost) every pixel (round 5).
// The reference (imgwarp.cpp:3199-3240) forms, in double,  q = RN(32 / W),  p = RN(num * q),  X = cvRound(p)  with W = W0 + M6 x1, num = X0 + M0 x1.  The exact form costs, per
// pixel and in every lane, three row terms (12 double operations), a correctly rounded division (~14), four clamps and two saturating conversions: half the kernel's time
// (profiles/r04_warp32_ab.txt: "the per-pixel double division dominates").  Here:
//   * the row terms X0, Y0, W0 depend on (block start xb, row y) only; the reference's blocks are 64 columns wide and tiles are 64-aligned, so they are WAVE-UNIFORM:
//     lane l computes them for row l of the wave's rows, every row then reads them back with v_readlane (persp::RowTerms);
//   * the column terms M0 x1, M3 x1, M6 x1 are per-lane constants for all rows;
//   * q' = 32 r with r from v_rcp_f64 and two Newton steps: |q' - 32 / W| <= 2^-50 |q| (the seed is good to >= 2^-13, two steps square that twice; one rounding each), so
//     |q' - q| <= 2^-49 |q| and p' = RN(num q') lies within 2^-48 |p| of the reference's p.  cvRound(p') != cvRound(p) needs a half-integer between the two, i.e. within
//     2^-48 |p| of p'.  A pixel whose p' (either coordinate) lies within 2^-19 of a half-integer, or beyond 2^21 in magnitude (2^-19 >= 2^-40 |p| there: a 2^8 margin on
//     the bound), or whose W is zero / not finite, is RE-EVALUATED with the exact division -- about one pixel in 2^17; everything else is proven equal.
//   * the clamps to [INT_MIN, INT_MAX] are what the saturating conversion does anyway.
namespace persp {
struct RowTerms { double X0, Y0, W0; };
__device__ __forceinline__ double bcast(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ RowTerms rowTerms(const WarpArgs& w, int xb, int y)
{
    RowTerms t;
    t.X0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[0], (double)xb), __dmul_rn(w.M[1], (double)y)), w.M[2]);
    t.Y0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[3], (double)xb), __dmul_rn(w.M[4], (double)y)), w.M[5]);
    t.W0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[6], (double)xb), __dmul_rn(w.M[7], (double)y)), w.M[8]);
    return t;
}
// (X, Y) in 1/32 pixel from the row terms and the lane's column terms c0 = M0 x1, c3 = M3 x1, c6 = M6 x1
__device__ __forceinline__ void xy(const RowTerms& t, double c0, double c3, double c6, int& X, int& Y)
{
    const double W = __dadd_rn(t.W0, c6), nx = __dadd_rn(t.X0, c0), ny = __dadd_rn(t.Y0, c3);
    double r = __builtin_amdgcn_rcp(W);
    r = __fma_rn(r, __fma_rn(-W, r, 1.0), r);
    r = __fma_rn(r, __fma_rn(-W, r, 1.0), r);
    const double q = __dmul_rn(r, 32.0);
    const double fX = __dmul_rn(nx, q), fY = __dmul_rn(ny, q);
    const double dX = __builtin_amdgcn_fract(fX) - 0.5, dY = __builtin_amdgcn_fract(fY) - 0.5;
    const double big = fmax(fabs(fX), fabs(fY));
    // (comparisons written so that a NaN anywhere -- W = 0, infinities -- lands on the exact side)
    const bool safe = fmin(fabs(dX), fabs(dY)) > 0x1p-19 && big < 0x1p21;
    if (safe) { X = (int)__double2int_rn(fX); Y = (int)__double2int_rn(fY); return; }
    const double qe = W != 0 ? __ddiv_rn(32.0, W) : 0;
    double eX = __dmul_rn(nx, qe), eY = __dmul_rn(ny, qe);
    eX = fmax(-2147483648.0, fmin(2147483647.0, eX));
    eY = fmax(-2147483648.0, fmin(2147483647.0, eY));
    X = satIntD(eX); Y = satIntD(eY);
}
}

template <typename T> constexpr int warpRows() { return sizeof(T) == 4 ? 8 : 4; }
template <int KIND>
__device__ __forceinline__ void warpXY(const WarpArgs& w, int y, int ad, int bd, int xb, int x1, int& X, int& Y)
{
    if (KIND == 0) {
        const int X0 = satIntD(__dmul_rn(__dadd_rn(__dmul_rn(w.M[1], (double)y), w.M[2]), 1024.0)) + 16;
        const int Y0 = satIntD(__dmul_rn(__dadd_rn(__dmul_rn(w.M[4], (double)y), w.M[5]), 1024.0)) + 16;
        X = (X0 + ad) >> 5; Y = (Y0 + bd) >> 5;
    } else {
        const double X0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[0], (double)xb), __dmul_rn(w.M[1], (double)y)), w.M[2]);
        const double Y0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[3], (double)xb), __dmul_rn(w.M[4], (double)y)), w.M[5]);
        const double W0 = __dadd_rn(__dadd_rn(__dmul_rn(w.M[6], (double)xb), __dmul_rn(w.M[7], (double)y)), w.M[8]);
        double W = __dadd_rn(W0, __dmul_rn(w.M[6], (double)x1));
        W = W != 0 ? __ddiv_rn(32.0, W) : 0;
        double fX = __dmul_rn(__dadd_rn(X0, __dmul_rn(w.M[0], (double)x1)), W);
        double fY = __dmul_rn(__dadd_rn(Y0, __dmul_rn(w.M[3], (double)x1)), W);
        fX = fmax(-2147483648.0, fmin(2147483647.0, fX));
        fY = fmax(-2147483648.0, fmin(2147483647.0, fY));
        X = satIntD(fX); Y = satIntD(fY);
    }
}

// The kernel is bound by its VALU work, not by HBM or the gather (a pure shift, perfectly coalesced, ran no faster than a rotation), so the
// per-pixel instruction count is what is kept low:
//  * affine row terms X0(y), Y0(y) -- double arithmetic with saturation -- are wave-uniform: lane i (i < 8) evaluates row i of the wave's WROWS
//    rows once, the loop takes them with v_readlane; the column terms are evaluated once per thread;
//  * byte offsets are 32-bit (the host checks that both images are below 4 GB), one v_mad per tap address, saddr-form loads and stores;
//  * the taps of a group of G rows are loaded back to back before the first is used (pixels whose footprint is not strictly inside the source
//    load from offset 0 and are redone by the generic sampler afterwards).
template <typename T, int CN, int KIND /*0 affine, 1 perspective*/>
__global__ __launch_bounds__(256) void k_warp_lin(const uchar* __restrict__ src, uint32_t sstep, uchar* __restrict__ dst, uint32_t dstep,
                                                  SampleArgs s, WarpArgs w, const short* __restrict__ tab)
{
    constexpr int WROWS = warpRows<T>();
    constexpr int G0 = sizeof(T) == 4 ? (CN == 1 ? 8 : CN == 3 ? 4 : 2) : 8, G = G0 < WROWS ? G0 : WROWS;   // rows whose taps are in flight together
    constexpr uint32_t ESZ = CN * sizeof(T);
    typedef float fNu __attribute__((ext_vector_type(2), aligned(4)));
    typedef unsigned short u16u __attribute__((aligned(1)));
    typedef unsigned long long u64u __attribute__((aligned(1)));
    // 8-bit sources: the 1024 x 4 Q15 weights of initInterTab2D (8 KB) are copied into LDS once per workgroup -- read per pixel from global
    // memory they were a third gather, and the most scattered one (64 lanes anywhere in 8 KB), on a path that is bound by the vector L1
    __shared__ uint2 ltab[sizeof(T) == 1 ? 1024 : 1];
    if (sizeof(T) == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) ltab[threadIdx.x + 256 * i] = reinterpret_cast<const uint2*>(tab)[threadIdx.x + 256 * i];
        __syncthreads();
    }
    int tx, ty;
    tileOf(w, tx, ty);
    src += (size_t)blockIdx.z * w.sframe; dst += (size_t)blockIdx.z * w.dframe;
    const int lane = threadIdx.x & 63;
    const int x = tx * 64 + lane;
    const int yb = __builtin_amdgcn_readfirstlane((ty * 4 + (int)(threadIdx.x >> 6)) * WROWS);
    if (yb >= w.dh) return;                                                              // wave-uniform
    int X0v = 0, Y0v = 0;
    if (KIND == 0) {                                                                     // all 64 lanes are still active here
        const int yl = yb + (lane & (WROWS - 1));
        X0v = satIntD(__dmul_rn(__dadd_rn(__dmul_rn(w.M[1], (double)yl), w.M[2]), 1024.0)) + 16;
        Y0v = satIntD(__dmul_rn(__dadd_rn(__dmul_rn(w.M[4], (double)yl), w.M[5]), 1024.0)) + 16;
    }
    // perspective: the reference's 64-column blocks coincide with the tiles -> the row terms are wave-uniform, lane l holds those of row l of the wave (persp:: above)
    const bool puni = KIND == 1 && w.bw0 == 64 && !w.pexact;
    persp::RowTerms prt = {0, 0, 0};
    if (KIND == 1 && puni) prt = persp::rowTerms(w, tx * 64, yb + (lane & (WROWS - 1)));
    if (x >= w.dw) return;
    const int ad = KIND == 0 ? satIntD(__dmul_rn(__dmul_rn(w.M[0], (double)x), 1024.0)) : 0;
    const int bd = KIND == 0 ? satIntD(__dmul_rn(__dmul_rn(w.M[3], (double)x), 1024.0)) : 0;
    const int xb = KIND == 1 ? (x / w.bw0) * w.bw0 : 0, x1 = x - xb;
    const double pc0 = KIND == 1 ? __dmul_rn(w.M[0], (double)x1) : 0, pc3 = KIND == 1 ? __dmul_rn(w.M[3], (double)x1) : 0, pc6 = KIND == 1 ? __dmul_rn(w.M[6], (double)x1) : 0;
    // the pair (sx, sx+1) is read as ONE load of 2*CN elements where that cannot leave the row (3 channels: an 8-byte load for 6)
    const int xlim = CN == 3 ? s.sw - 2 : s.sw - 1;
    const uint32_t xoff = (uint32_t)x * ESZ;
    unsigned slow = 0;                                                                   // rows left to the generic sampler
#pragma unroll
    for (int g0 = 0; g0 < WROWS; g0 += G) {
        int Xs[G], Ys[G];
        bool in[G];
        fNu f0[sizeof(T) == 4 ? G * CN : 1], f1[sizeof(T) == 4 ? G * CN : 1];
        unsigned long long q0[sizeof(T) == 4 ? 1 : G], q1[sizeof(T) == 4 ? 1 : G];
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int y = yb + g0 + i;
            if (KIND == 0) {
                Xs[i] = (__builtin_amdgcn_readlane(X0v, g0 + i) + ad) >> 5;
                Ys[i] = (__builtin_amdgcn_readlane(Y0v, g0 + i) + bd) >> 5;
            } else if (KIND == 1 && puni) {
                persp::RowTerms t;
                t.X0 = persp::bcast(prt.X0, g0 + i); t.Y0 = persp::bcast(prt.Y0, g0 + i); t.W0 = persp::bcast(prt.W0, g0 + i);
                persp::xy(t, pc0, pc3, pc6, Xs[i], Ys[i]);
            } else warpXY<KIND>(w, y, ad, bd, xb, x1, Xs[i], Ys[i]);
            // the saturation to short of the reference cannot turn an outside position into an inside one (sw, sh <= 32767): the unsaturated
            // values decide; the generic sampler saturates for the others
            const int sx = Xs[i] >> 5, sy = Ys[i] >> 5;
            const bool inside = (unsigned)sx < (unsigned)xlim && (unsigned)sy < (unsigned)(s.sh - 1);
            in[i] = y < w.dh && inside;
            if (y < w.dh && !inside) slow |= 1u << (g0 + i);
            const uint32_t off = in[i] ? __umul24((uint32_t)sy, sstep) + (uint32_t)sx * ESZ : 0u;       // sy < 2^15, sstep < 2^24 (host check)
            const uchar* r0 = src + off;
            const uchar* r1 = src + (off + sstep);
            if (sizeof(T) == 4) {
#pragma unroll
                for (int q = 0; q < CN; q++) { f0[i * CN + q] = reinterpret_cast<const fNu*>(r0)[q]; f1[i * CN + q] = reinterpret_cast<const fNu*>(r1)[q]; }
            } else if (CN == 1) { q0[i] = *reinterpret_cast<const u16u*>(r0); q1[i] = *reinterpret_cast<const u16u*>(r1); }
            else { q0[i] = *reinterpret_cast<const u64u*>(r0); q1[i] = *reinterpret_cast<const u64u*>(r1); }
        }
        // (taking a pixel's upper taps from the registers of the pixel above it, where they are the same two source pixels, and masking those
        // lanes out of the load was tried: the per-lane conditions cost more than the lighter gather saves -- 8K 32F, 7 degrees: 69.7 vs 63.5 us)
        if (sizeof(T) == 4 && CN == 1) {
            // CV_32FC1: two rows per instruction in packed float arithmetic (v_pk_mul_f32 / v_pk_add_f32; the same products and the same order of sums per row), computed
            // for every row pair -- rows outside the source loaded element 0 and are simply not stored.  Round 4: the kernel issues ~41 instructions per pixel and is bound
            // by that (a pure shift is no faster than a rotation); the weights and the blend were 19 of them.
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < G; i += 2) {
                const int j = i + 1 < G ? i + 1 : i;
                const f2 s32 = {1.f / 32, 1.f / 32}, one = {1.f, 1.f};
                const f2 fx = f2{(float)(Xs[i] & 31), (float)(Xs[j] & 31)} * s32, fy = f2{(float)(Ys[i] & 31), (float)(Ys[j] & 31)} * s32;
                const f2 wy0 = one - fy, wx0 = one - fx;
                const f2 w0 = wy0 * wx0, w1 = wy0 * fx, w2 = fy * wx0, w3 = fy * fx;
                const f2 p00 = {f0[i].x, f0[j].x}, p01 = {f0[i].y, f0[j].y}, p10 = {f1[i].x, f1[j].x}, p11 = {f1[i].y, f1[j].y};
                f2 t = p00 * w0 + p01 * w1;
                t = t + p10 * w2;
                t = t + p11 * w3;
                if (in[i]) *reinterpret_cast<float*>(dst + (uint32_t)(yb + g0 + i) * dstep + xoff) = t.x;
                if (j != i && in[j]) *reinterpret_cast<float*>(dst + (uint32_t)(yb + g0 + j) * dstep + xoff) = t.y;
            }
        } else
#pragma unroll
        for (int i = 0; i < G; i++) {
            if (!in[i]) continue;
            const int y = yb + g0 + i;
            const int ax = Xs[i] & 31, ay = Ys[i] & 31;
            uchar* D = dst + (uint32_t)y * dstep + xoff;
            if (sizeof(T) == 4) {
                const float s32 = 1.f / 32;
                const float fx = ax * s32, fy = ay * s32;
                const float wy0 = 1.f - fy, wx0 = 1.f - fx;
                const float w0 = __fmul_rn(wy0, wx0), w1 = __fmul_rn(wy0, fx), w2 = __fmul_rn(fy, wx0), w3 = __fmul_rn(fy, fx);
                float p0[2 * CN], p1[2 * CN];
#pragma unroll
                for (int q = 0; q < CN; q++) {
                    p0[2 * q] = f0[i * CN + q].x; p0[2 * q + 1] = f0[i * CN + q].y; p1[2 * q] = f1[i * CN + q].x; p1[2 * q + 1] = f1[i * CN + q].y;
                }
#pragma unroll
                for (int c = 0; c < CN; c++) {
                    float t = __fadd_rn(__fmul_rn(p0[c], w0), __fmul_rn(p0[CN + c], w1));
                    t = __fadd_rn(t, __fmul_rn(p1[c], w2));
                    t = __fadd_rn(t, __fmul_rn(p1[CN + c], w3));
                    reinterpret_cast<float*>(D)[c] = t;
                }
            } else {
                const uint2 wl = ltab[ay * 32 + ax];
                short4 wq; wq.x = (short)(wl.x & 0xffff); wq.y = (short)(wl.x >> 16); wq.z = (short)(wl.y & 0xffff); wq.w = (short)(wl.y >> 16);
#pragma unroll
                for (int c = 0; c < CN; c++) {
                    const int v0 = (int)((q0[i] >> (8 * c)) & 255), v1 = (int)((q0[i] >> (8 * (CN + c))) & 255);
                    const int v2 = (int)((q1[i] >> (8 * c)) & 255), v3 = (int)((q1[i] >> (8 * (CN + c))) & 255);
                    const int r = (v0 * wq.x + v1 * wq.y + v2 * wq.z + v3 * wq.w + (1 << 14)) >> 15;
                    D[c] = (uchar)(r < 0 ? 0 : r > 255 ? 255 : r);
                }
            }
        }
    }
    if (slow) {
#pragma unroll 1
        for (int i = 0; i < WROWS; i++) {
            if (!((slow >> i) & 1)) continue;
            const int y = yb + i;
            int X, Y;
            warpXY<KIND>(w, y, ad, bd, xb, x1, X, Y);
            samplePixel(src, sstep, dst + (size_t)y * dstep + xoff, s, satShort(X >> 5), satShort(Y >> 5), X & 31, Y & 31, tab);
        }
    }
}


// ---- CV_32FC1 bilinear warpAffine through an LDS tile (BASELINE config 3c) --------------------------------------------------------------------------
// Why: k_warp_lin<float> gathers.  A 64-lane load of 8-byte tap pairs along a sloped source line touches ~26 cache lines (profiles/r02_warp_pmc.txt:
// TCP_TOTAL_CACHE_ACCESSES 24.2 M for 0.93 M load instructions per 8K frame) and the vector L1 looks up one line per clock: ~95 K cycles per CU and frame, the whole
// kernel time.  Here a workgroup owns a 64 x 32 destination tile: the reference's coordinate sums are a function of x plus a function of y, each monotone
// (saturating double -> int of a linear term), so the four corner terms bound every pixel's source position EXACTLY; the bounding box of the 2 x 2 footprints is
// copied into LDS row by row with consecutive lanes on consecutive floats (a handful of lines per row), and the taps come from LDS.  The per-pixel arithmetic is
// k_warp_lin's (the reference's: imgwarp.cpp:2233-2298 coordinates in 1/1024 rounded to 1/32, remapBilinear<Cast<float,float>> :675-904 weights and
// summation order), so results are bit-identical to it.  Tiles whose box leaves the source or exceeds the LDS allotment take the gather form pixel by pixel
// (taps from global memory; the generic sampler where the footprint is not inside).
constexpr int W32_TW = 64, W32_TH = 32, W32_CAP = 4608;          // tile, LDS floats (18 KB: eight workgroups per CU by LDS and by threads)
struct W32Box { int bx0, by0, bw, bh; bool tiled; };
// the exact source bounding box of the tile [x0, x1] x [y0, y1] from its corner terms (column terms at x0 / x1, row terms at y0 / y1), and whether the LDS kernel takes it
__device__ __forceinline__ W32Box warp32Box(const SampleArgs& s, int adA, int adB, int bdA, int bdB, int XrA, int XrB, int YrA, int YrB)
{
    const long long Xlo = (long long)min(adA, adB) + min(XrA, XrB), Xhi = (long long)max(adA, adB) + max(XrA, XrB);
    const long long Ylo = (long long)min(bdA, bdB) + min(YrA, YrB), Yhi = (long long)max(bdA, bdB) + max(YrA, YrB);
    W32Box b;
    b.tiled = Xlo > -(1ll << 30) && Xhi < (1ll << 30) && Ylo > -(1ll << 30) && Yhi < (1ll << 30);      // the int sums of the reference do not wrap here
    b.bx0 = (int)(Xlo >> 10); b.by0 = (int)(Ylo >> 10);
    const int bx1 = (int)(Xhi >> 10) + 1, by1 = (int)(Yhi >> 10) + 1;
    b.bw = bx1 - b.bx0 + 1; b.bh = by1 - b.by0 + 1;
    b.tiled = b.tiled && b.bx0 >= 0 && b.by0 >= 0 && bx1 <= s.sw - 1 && by1 <= s.sh - 1 && (long long)b.bw * b.bh <= W32_CAP;
    return b;
}
__device__ __forceinline__ int warp32RowX(const WarpArgs& w, int y) { return satIntD(__dmul_rn(__dadd_rn(__dmul_rn(w.M[1], (double)y), w.M[2]), 1024.0)) + 16; }
__device__ __forceinline__ int warp32RowY(const WarpArgs& w, int y) { return satIntD(__dmul_rn(__dadd_rn(__dmul_rn(w.M[4], (double)y), w.M[5]), 1024.0)) + 16; }
__device__ __forceinline__ int warp32ColX(const WarpArgs& w, int x) { return satIntD(__dmul_rn(__dmul_rn(w.M[0], (double)x), 1024.0)); }
__device__ __forceinline__ int warp32ColY(const WarpArgs& w, int x) { return satIntD(__dmul_rn(__dmul_rn(w.M[3], (double)x), 1024.0)); }

// the coordinate terms of every destination column and row, once per call: terms = colX[dw], colY[dw], rowX[dh], rowY[dh]; work[0] = 0 (the rest list's length)
__global__ __launch_bounds__(256) void k_warp32_terms(WarpArgs w, int* __restrict__ terms, uint32_t* __restrict__ work)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) work[0] = 0;
    if (i < w.dw) { terms[i] = warp32ColX(w, i); terms[w.dw + i] = warp32ColY(w, i); }
    if (i < w.dh) { terms[2 * w.dw + i] = warp32RowX(w, i); terms[2 * w.dw + w.dh + i] = warp32RowY(w, i); }
}

struct W32Bx { int bx0, by0, pitch, n4, bh; bool tiled; };       // the box widened to whole float4 columns: first column (multiple of 4), first row, floats per row, float4 elements
__device__ __forceinline__ W32Bx warp32Widen(const SampleArgs& s, const W32Box& e)
{
    W32Bx b;
    b.bx0 = e.bx0 & ~3; b.by0 = e.by0; b.bh = e.bh;
    const int bx1 = (e.bx0 + e.bw - 1) | 3;
    b.pitch = bx1 - b.bx0 + 1;
    b.n4 = (b.pitch >> 2) * e.bh;
    b.tiled = e.tiled && bx1 <= s.sw - 1 && b.n4 <= W32_CAP / 4;
    return b;
}

__global__ __launch_bounds__(256) void k_warp32_tile(const uchar* __restrict__ src, uint32_t sstep, uchar* __restrict__ dst, uint32_t dstep, SampleArgs s, WarpArgs w, int tpw,
                                                     const int* __restrict__ terms, uint32_t* __restrict__ work)
{
    // A workgroup walks `tpw` horizontally adjacent tiles of one tile row.  Two LDS buffers: while tile j is computed from one, the box of tile j + 1 is on its way from
    // L2 into registers (16-byte loads: the box is widened to whole float4 columns, element i = tid + 256 u of the dense box, consecutive lanes on consecutive 16 bytes
    // of a box row), and is parked in the other buffer afterwards -- one barrier per tile, the load latency of a tile under the arithmetic of its predecessor.
    // History (profiles/r04_warp32_ab.txt, 8K frame, 7 degrees; the gather kernel: 75 us): staged row by row and synchronously 154 us; batched dword loads 95;
    // double-buffered 93 -- by then bound by instruction issue (~700 instructions per thread and tile: 18 dword loads with a division each, 45 per pixel); 16-byte
    // staging, scalar row terms and packed float arithmetic 84, of which ~24 were the second kernel evaluating the tile predicate in double arithmetic in every thread:
    // the terms now come from a table built once per call and the tiles left over go through a list.
    __shared__ __attribute__((aligned(16))) float box[2][W32_CAP];
    constexpr int NE = (W32_CAP / 4 + 255) / 256;                    // float4 box elements per thread
    constexpr int RPW = W32_TH / 4;                                  // rows per wave
    typedef float f2 __attribute__((ext_vector_type(2)));
    src += (size_t)blockIdx.z * w.sframe; dst += (size_t)blockIdx.z * w.dframe;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = blockIdx.y, tx0 = blockIdx.x * tpw, ntile = min(tpw, w.gx - tx0);
    const int y0 = ty * W32_TH, y1 = min(y0 + W32_TH, w.dh) - 1;
    const int* colX = terms; const int* colY = terms + w.dw; const int* rowX = terms + 2 * w.dw; const int* rowY = rowX + w.dh;
    int rX[RPW], rY[RPW];                                            // uniform: scalar loads
#pragma unroll
    for (int i = 0; i < RPW; i++) { const int y = min(y0 + wave * RPW + i, w.dh - 1); rX[i] = rowX[y]; rY[i] = rowY[y]; }
    const int XrA = rowX[y0], XrB = rowX[y1], YrA = rowY[y0], YrB = rowY[y1];
    W32Bx cur, nxt;
    int adC = 0, bdC = 0, adN = 0, bdN = 0;                          // this thread's column terms in the current / next tile
    float4 v[NE];
    auto prepare = [&](int j, W32Bx& b, int& ad, int& bd) {          // box of tile j and the thread's column terms; issues the loads of its elements into v
        const int x0 = (tx0 + j) * W32_TW, x1 = min(x0 + W32_TW, w.dw) - 1;
        const int xc = min(x0 + lane, w.dw - 1);
        ad = colX[xc]; bd = colY[xc];
        b = warp32Widen(s, warp32Box(s, colX[x0], colX[x1], colY[x0], colY[x1], XrA, XrB, YrA, YrB));
        if (!b.tiled) {
            if (tid == 0) { const uint32_t k = atomicAdd(work, 1u); work[1 + k] = ((uint32_t)blockIdx.z * (uint32_t)w.gy + (uint32_t)ty) * (uint32_t)w.gx + (uint32_t)(tx0 + j); }
            return;
        }
        const int c4 = b.pitch >> 2;
        const unsigned magic = 0xffffffffu / (unsigned)c4 + (c4 > 1 ? 1u : 0u);          // i / c4 for i < 2^16 by one mul_hi (c4 == 1: i itself)
        const uchar* gb = src + (size_t)b.by0 * sstep + (size_t)b.bx0 * 4;
#pragma unroll
        for (int u = 0; u < NE; u++) {
            const int i = tid + 256 * u;
            const int r = c4 > 1 ? (int)__umulhi((unsigned)i, magic) : i, c = i - r * c4;
            v[u] = i < b.n4 ? *reinterpret_cast<const float4*>(gb + (size_t)r * sstep + (size_t)c * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto park = [&](const W32Bx& b, float* buf) {
        if (!b.tiled) return;
#pragma unroll
        for (int u = 0; u < NE; u++) { const int i = tid + 256 * u; if (i < b.n4) reinterpret_cast<float4*>(buf)[i] = v[u]; }
    };
    prepare(0, cur, adC, bdC);
    park(cur, box[0]);
    __syncthreads();
    for (int j = 0; j < ntile; j++) {
        const bool more = j + 1 < ntile;
        if (more) prepare(j + 1, nxt, adN, bdN);
        const int x0 = (tx0 + j) * W32_TW, x1 = min(x0 + W32_TW, w.dw) - 1, x = x0 + lane;
        if (cur.tiled && x <= x1) {                                  // (tiles the LDS path does not take are on the list for k_warp32_rest)
            const float* bx = box[j & 1];
            const int pitch = cur.pitch, base = -cur.by0 * pitch - cur.bx0;
            uchar* dcol = dst + (size_t)x * 4u;
#pragma unroll
            for (int i = 0; i < RPW; i += 2) {                       // rows i and i + 1 of the wave together: packed float arithmetic
                const int ya = y0 + wave * RPW + i;
                if (ya > y1) break;
                const int Xa = (rX[i] + adC) >> 5, Ya = (rY[i] + bdC) >> 5, Xb = (rX[i + 1] + adC) >> 5, Yb = (rY[i + 1] + bdC) >> 5;
                const float* la = bx + ((Ya >> 5) * pitch + (Xa >> 5) + base);
                const float* lb = ya + 1 <= y1 ? bx + ((Yb >> 5) * pitch + (Xb >> 5) + base) : la;
                const f2 p00 = {la[0], lb[0]}, p01 = {la[1], lb[1]}, p10 = {la[pitch], lb[pitch]}, p11 = {la[pitch + 1], lb[pitch + 1]};
                const f2 s32 = {1.f / 32, 1.f / 32}, one = {1.f, 1.f};
                const f2 fx = f2{(float)(Xa & 31), (float)(Xb & 31)} * s32, fy = f2{(float)(Ya & 31), (float)(Yb & 31)} * s32;
                const f2 wy0 = one - fy, wx0 = one - fx;
                const f2 w0 = wy0 * wx0, w1 = wy0 * fx, w2 = fy * wx0, w3 = fy * fx;
                f2 t = p00 * w0 + p01 * w1;
                t = t + p10 * w2;
                t = t + p11 * w3;
                *reinterpret_cast<float*>(dcol + (size_t)ya * dstep) = t.x;
                if (ya + 1 <= y1) *reinterpret_cast<float*>(dcol + (size_t)(ya + 1) * dstep) = t.y;
            }
        }
        if (more) { park(nxt, box[(j + 1) & 1]); cur = nxt; adC = adN; bdC = bdN; }
        __syncthreads();
    }
}

// the tiles k_warp32_tile left (source box not wholly inside the image, or too large for its LDS allotment), from its list: the gather form, pixel by pixel
__global__ __launch_bounds__(256) void k_warp32_rest(const uchar* __restrict__ src0, uint32_t sstep, uchar* __restrict__ dst0, uint32_t dstep, SampleArgs s, WarpArgs w,
                                                     const short* __restrict__ tab, const int* __restrict__ terms, const uint32_t* __restrict__ work)
{
    const uint32_t count = work[0];
    const int* colX = terms; const int* colY = terms + w.dw; const int* rowX = terms + 2 * w.dw; const int* rowY = rowX + w.dh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t t = blockIdx.x; t < count; t += gridDim.x) {        // (launched with one block per tile of the image: blocks beyond the list's length leave at once)
        const uint32_t id = work[1 + t];
        const int tx = (int)(id % (uint32_t)w.gx), ty = (int)((id / (uint32_t)w.gx) % (uint32_t)w.gy), fr = (int)(id / ((uint32_t)w.gx * (uint32_t)w.gy));
        const uchar* src = src0 + (size_t)fr * w.sframe; uchar* dst = dst0 + (size_t)fr * w.dframe;
        const int x0 = tx * W32_TW, y0 = ty * W32_TH;
        const int x1 = min(x0 + W32_TW, w.dw) - 1, y1 = min(y0 + W32_TH, w.dh) - 1;
        const int x = x0 + lane;
        if (x > x1) continue;
        const int ad = colX[x], bd = colY[x];
        const uint32_t xoff = (uint32_t)x * 4u;
        // BORDER_CONSTANT and the tile's whole source box outside the image (large rotations leave such tiles by the thousand): the border value, no sampling
        const W32Box e = warp32Box(s, colX[x0], colX[x1], colY[x0], colY[x1], rowX[y0], rowX[y1], rowY[y0], rowY[y1]);
        if (s.border == B_CONSTANT && (e.bx0 > s.sw - 1 || e.bx0 + e.bw - 1 < 0 || e.by0 > s.sh - 1 || e.by0 + e.bh - 1 < 0)) {
#pragma unroll
            for (int i = 0; i < W32_TH / 4; i++) { const int y = y0 + wave * (W32_TH / 4) + i; if (y <= y1) *reinterpret_cast<float*>(dst + (size_t)y * dstep + xoff) = s.cval[0]; }
            continue;
        }
        int Xs[W32_TH / 4], Ys[W32_TH / 4];
        float q00[W32_TH / 4], q01[W32_TH / 4], q10[W32_TH / 4], q11[W32_TH / 4];
        unsigned in = 0;
#pragma unroll
        for (int i = 0; i < W32_TH / 4; i++) {                       // all of the thread's taps are requested before the first is used
            const int y = min(y0 + wave * (W32_TH / 4) + i, y1);
            Xs[i] = (rowX[y] + ad) >> 5; Ys[i] = (rowY[y] + bd) >> 5;
            const int sx = Xs[i] >> 5, sy = Ys[i] >> 5;
            const bool inside = (unsigned)sx < (unsigned)(s.sw - 1) && (unsigned)sy < (unsigned)(s.sh - 1);
            in |= inside ? 1u << i : 0u;
            const float* g = reinterpret_cast<const float*>(src + (size_t)(inside ? sy : 0) * sstep) + (inside ? sx : 0);
            const float* g1 = reinterpret_cast<const float*>(reinterpret_cast<const uchar*>(g) + sstep);
            q00[i] = g[0]; q01[i] = g[1]; q10[i] = g1[0]; q11[i] = g1[1];
        }
#pragma unroll
        for (int i = 0; i < W32_TH / 4; i++) {
            const int y = y0 + wave * (W32_TH / 4) + i;
            if (y > y1) break;
            const int ax = Xs[i] & 31, ay = Ys[i] & 31;
            if ((in >> i) & 1) {
                const float s32 = 1.f / 32;
                const float fx = ax * s32, fy = ay * s32;
                const float wy0 = 1.f - fy, wx0 = 1.f - fx;
                const float w0 = __fmul_rn(wy0, wx0), w1 = __fmul_rn(wy0, fx), w2 = __fmul_rn(fy, wx0), w3 = __fmul_rn(fy, fx);
                float tt = __fadd_rn(__fmul_rn(q00[i], w0), __fmul_rn(q01[i], w1));
                tt = __fadd_rn(tt, __fmul_rn(q10[i], w2));
                tt = __fadd_rn(tt, __fmul_rn(q11[i], w3));
                *reinterpret_cast<float*>(dst + (size_t)y * dstep + xoff) = tt;
            } else samplePixel(src, sstep, dst + (size_t)y * dstep + xoff, s, satShort(Xs[i] >> 5), satShort(Ys[i] >> 5), ax, ay, tab);
        }
    }
}

// ---- CV_32FC1 bilinear warpAffine, strip walk through an LDS ring of whole source row pieces (BASELINE config 3c, round 6) ---------------------------------------
// What the probes of round 5 say (profiles/r05_shift_probe.txt, r05_store_geometry_probe.txt): every form that fetches the source as 64-pixel (256-byte) row pieces --
// the gather kernel and the 64 x 32 LDS tiles alike -- stays at 0.50-0.58 of 8 TB/s with no arithmetic at all; 1 KiB row pieces with each source row fetched ONCE and one
// 16-byte store per lane reach 0.69-0.74.  This kernel is that shape for a rotated map:
//   * a workgroup of 8 waves owns a strip of 256 destination columns and walks DOWN a segment of rows, 8 rows per step -- one destination row per wave, 4 consecutive
//     pixels per lane, ONE 16-byte non-temporal store per lane (1 KiB per wave and row);
//   * the source rows the strip touches are staged as pieces of WS_PW = 288 floats (1152 bytes: the strip's horizontal footprint, |M0| 256 pixels, plus the drift
//     over the rows that share a source row) with asynchronous global -> LDS loads (global_load_lds_dwordx4: no staging registers, 1 KiB per wave-instruction) into a ring
//     of WS_NR = 64 row slots (slot = source row mod 64).  For |rotation| up to ~12 degrees one destination row needs |M3| 256 + 2 <= 55 source rows and each new
//     destination row about |M4| new ones: every source byte the strip needs is fetched once per segment;
//   * the row pieces of step j + 1 are requested after the taps of step j have been read (LDS -> registers), and land while step j blends; one s_waitcnt vmcnt(0) +
//     s_barrier per step, BEFORE the step's stores are issued, so that the stores overlap the next step.
// The origin of the piece of source row r is bx(r) = ((r H + C) >> 15) & ~3, a linear bound of the leftmost source column any pixel of the strip reads on that row.  It
// only has to be right in almost every case: a tap whose row is not resident or whose columns fall outside its row's piece (and every pixel whose 2 x 2 footprint is not
// strictly inside the source) is evaluated from global memory by the generic sampler.  The arithmetic per pixel is k_warp_lin's -- the reference's (imgwarp.cpp:2233-2298
// coordinates in 1/1024 rounded to 1/32, remapBilinear<Cast<float, float>> :675-904 weights and summation order) -- so results are bit-identical to it.
constexpr int WS_COLS = 256, WS_NR = 64, WS_PW = 288, WS_WAVES = 8;
struct StripArgs { int pitch /* floats per ring slot */, segRows, H15, dbg /* MI355CV_WARP32_DBG: 1 no row requests after the prologue, 2 no stores, 4 no LDS tap reads (timing decomposition only: wrong pixels) */; double g, cp; /* bx(r) = ((r H15 + C15(strip)) >> 15) & ~3, C15 from min(g x0, g x1) + cp */ };

__global__ __launch_bounds__(64 * WS_WAVES) void k_warp32_strip(const uchar* __restrict__ src, uint32_t sstep, uchar* __restrict__ dst, uint32_t dstep, SampleArgs s, WarpArgs w,
                                                                StripArgs a, const int* __restrict__ terms, uchar* __restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) float ring[];         // WS_NR slots of a.pitch floats
    src += (size_t)blockIdx.z * w.sframe; dst += (size_t)blockIdx.z * w.dframe;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * WS_COLS, xe = min(x0 + WS_COLS, w.dw) - 1;
    const int y0 = blockIdx.y * a.segRows, y1 = min(y0 + a.segRows, w.dh) - 1;
    const int* colX = terms; const int* colY = terms + w.dw; const int* rowX = terms + 2 * w.dw; const int* rowY = rowX + w.dh;
    // the lane's 4 columns and their terms (columns beyond the image repeat the last one: computed, never stored)
    const int xl = x0 + 4 * lane;
    int cx[4], cy[4];
#pragma unroll
    for (int o = 0; o < 4; o++) { const int xc = min(xl + o, w.dw - 1); cx[o] = colX[xc]; cy[o] = colY[xc]; }
    const int cYa = colY[x0], cYb = colY[xe];
    const int cYmin = min(cYa, cYb), cYmax = max(cYa, cYb);
    // bx(r): origin of source row r's piece
    const double gx0 = a.g * (double)x0, gx1 = a.g * (double)xe;
    const int C15 = (int)floor((fmin(gx0, gx1) + a.cp) * 32768.0);
    auto bxOf = [&](int r) { return ((__mul24(r, a.H15) + C15) >> 15) & ~3; };                  // |r| < 2^15, |H15| < 2^14: one v_mad_i32_i24
    const int pitch = a.pitch, pitch4 = 4 * a.pitch;
    // rows [lo, hi] of the source into their ring slots: row lo + i is loaded by wave i % 8, exactly two LDS-DMA instructions per row (64 + 8 lanes x 16 bytes); returns
    // how many this wave issued.  The instructions are inline assembly ON PURPOSE: the compiler does not see them, so it neither drains them before the next LDS read
    // (it cannot tell which slots they write) nor counts them -- the waits below do.  Chunks that would start outside the row are fetched from a clamped position: every
    // lane issues, the count is exact, and what lands in those slots is never read (columns outside the source fail the lane test).
    auto glds16 = [&](const uchar* g, const float* l) {
        unsigned keep;
        const unsigned ldsAddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)l;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(ldsAddr) : "memory");
    };
    // What lies outside the source is put into the ring as the BORDER VALUE (the host sends BORDER_CONSTANT calls only): rows above / below the image and the 16-byte
    // chunks left / right of it are written by the lanes themselves (ds_write) instead of being fetched.  A footprint that crosses the source's rim then reads its
    // taps like any other -- remapBilinear's border branch is the same expression with cval for the missing taps (imgwarp.cpp:836-870) -- and no lane needs a sampler.
    const float cvf = s.cval[0];
    const float4 cv4 = make_float4(cvf, cvf, cvf, cvf);
    auto request = [&](int lo, int hi) -> int {
        int issued = 0;
        for (int r = lo + wave; r <= hi; r += WS_WAVES) {
            const int bx = bxOf(r);
            float* slot = ring + (r & (WS_NR - 1)) * pitch;
            const int xa = bx + 4 * lane, xb = bx + 256 + 4 * lane;
            const bool rowIn = (unsigned)r < (unsigned)s.sh;                                    // wave-uniform
            const bool inA = rowIn && xa >= 0 && xa + 3 < s.sw, inB = rowIn && xb >= 0 && xb + 3 < s.sw, tail = lane < (WS_PW - 256) / 4;
            const uchar* grow = src + (size_t)(rowIn ? r : 0) * sstep;
            if (__ballot(inA)) { if (inA) glds16(grow + (size_t)xa * 4, slot); issued++; }      // (issued <=> some lane is active: the count is exact)
            if (!inA) *reinterpret_cast<float4*>(slot + 4 * lane) = cv4;
            if (__ballot(inB && tail)) { if (inB && tail) glds16(grow + (size_t)xb * 4, slot + 256); issued++; }
            if (!inB && tail) *reinterpret_cast<float4*>(slot + 256 + 4 * lane) = cv4;
        }
        return issued;
    };
    auto topRow = [&](int y) { return ((rowY[min(y, y1)] + cYmax) >> 10) + 1; };               // the last source row destination rows <= y of this segment read (M4 > 0: rows grow with y)
    // prologue: everything steps 0 and 1 need (the loop keeps TWO steps of row pieces in flight: ~33 KB per workgroup -- with one, a step waited out a whole HBM
    // latency with 11 KB in flight per workgroup and the kernel ran at 0.27 of 8 TB/s, profiles/r06_warp32_strip.txt)
    int rlo = (rowY[y0] + cYmin) >> 10;                                                        // first source row of the segment
    int have = topRow(y0 + 2 * WS_WAVES - 1);
    (void)request(rlo, have);
    __builtin_amdgcn_s_waitcnt(0x0070);                                                        // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    const uchar* ringB = reinterpret_cast<const uchar*>(ring);
    // the scalar terms of a step (its row terms, the first source row it reads, the last rows one / two steps ahead read) are fetched ONE STEP EARLY: six scalar loads
    // at the head of every step were a ~500-cycle stall each time (they feed the very first instructions)
    struct StepTerms { int rX, rY, rowMin, top2, top3; };
    auto termsOf = [&](int yb) {
        StepTerms t;
        const int yc = min(yb + wave, y1);
        t.rX = rowX[yc]; t.rY = rowY[yc];
        t.rowMin = (rowY[min(yb, y1)] + cYmin) >> 10;                                          // the first source row the step reads (rows grow with y)
        t.top2 = topRow(yb + 2 * WS_WAVES - 1); t.top3 = topRow(yb + 3 * WS_WAVES - 1);
        return t;
    };
    StepTerms nx = termsOf(y0);
    for (int yb = y0; yb <= y1; yb += WS_WAVES) {
        const int y = yb + wave;                                                               // this wave's destination row (wave-uniform)
        const bool live = y <= y1;
        const StepTerms cur = nx;
        const int rX = cur.rX, rY = cur.rY;
        // requests run up to two steps ahead, but never so far that they would overwrite a row THIS step still reads (slot = row mod 64): the depth adapts to the map
        // (the 7-degree map of BASELINE config 3c needs 60-63 rows for three steps: with a fixed depth of two a few lanes per wave missed their rows in many strips;
        // a second barrier per step after the tap reads, which frees 8 more slots, cost more than it bought: 68 against 64 us)
        const int rowMin = cur.rowMin;
        const int want = max(min(cur.top3, rowMin + (WS_NR - 1)), have);
        const bool late = have < cur.top2;                                                     // rows of the NEXT step are only being requested now: the wait below must cover them
        const int resLo = max(rowMin, rlo);
        nx = termsOf(min(yb + WS_WAVES, y1));                                                  // (scalar loads: they return while this step works)
        // ---- (a) coordinates, addresses, taps LDS -> registers.  The instruction count per pixel is what this kernel is bound by (the first version: 75 per pixel, two
        // 32-bit multiplies at a quarter of the rate among them, 107 us per 8K frame against the gather kernel's 74): the row / column extent of the lane's four pixels is
        // tested once per lane (sx and sy are monotone along a row), the two piece-relative columns of a pixel by ONE comparison (the origins of neighbouring rows differ
        // by 0 or 4), every product is a 24-bit one.
        int X[4], Y[4];
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v pT[4], pB[4];                                                                        // (p00, p01) and (p10, p11) of each pixel: what one ds_read2_b32 returns
#pragma unroll
        for (int o = 0; o < 4; o++) { X[o] = (rX + cx[o]) >> 5; Y[o] = (rY + cy[o]) >> 5; }
        const int sxA = X[0] >> 5, sxB = X[3] >> 5, syA = Y[0] >> 5, syB = Y[3] >> 5;
        const int sxLo = min(sxA, sxB), sxHi = max(sxA, sxB), r0 = min(syA, syB);
        // the lane's pixels read source rows r0 .. r0 + 2 at most (|M3| * 3 < 1): their piece origins and ring bases once per lane
        const int b0 = bxOf(r0), b1 = bxOf(r0 + 1), b2 = bxOf(r0 + 2);
        const int bLo = min(b0, b2), bHi = max(b0, b2);                                         // (bx is monotone in r)
        // a lane is served from the ring when its rows are resident and its columns inside their pieces -- wherever that is relative to the source (see request());
        // lanes whose four footprints are all outside are the border value without a read; what remains (a piece that missed: the origin bound is approximate) is
        // DEFERRED to k_warp32_strip_rest through a flag per row piece.  (Sampling such lanes here was tried three ways -- on the spot, from a list after the walk, taps
        // picked one by one from the ring -- and cost 25-80 us per 8K frame each: a load issued after the row pieces in flight and used before the barrier drains them
        // all, and a wave that does extra work holds its workgroup's other seven at the barrier.  profiles/r06_warp32_strip.txt.)
        const int syHi = max(syA, syB);
        const bool laneOk = r0 >= resLo && syHi < have && syHi - r0 <= 1 && sxLo - bHi >= 0 && sxHi + 1 - bLo <= WS_PW - 1;
        const bool laneOut = sxLo >= s.sw || sxHi + 1 < 0 || r0 >= s.sh || syHi + 1 < 0;
        const bool laneDefer = !laneOk && !laneOut && xl <= xe && live;
        if (laneOk && !(a.dbg & 4)) {
            const unsigned rb0 = __umul24((unsigned)r0 & (WS_NR - 1), (unsigned)pitch4) - 4u * (unsigned)b0;
            const unsigned rb1 = __umul24((unsigned)(r0 + 1) & (WS_NR - 1), (unsigned)pitch4) - 4u * (unsigned)b1;
            const unsigned rb2 = __umul24((unsigned)(r0 + 2) & (WS_NR - 1), (unsigned)pitch4) - 4u * (unsigned)b2;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int sx = X[o] >> 5, sy = Y[o] >> 5;
                const bool up = sy != r0;
                const f2v* la = reinterpret_cast<const f2v*>(ringB + ((up ? rb1 : rb0) + 4u * (unsigned)sx));
                const f2v* lb = reinterpret_cast<const f2v*>(ringB + ((up ? rb2 : rb1) + 4u * (unsigned)sx));
                typedef float f2u4 __attribute__((ext_vector_type(2), aligned(4)));
                pT[o] = *reinterpret_cast<const f2u4*>(la); pB[o] = *reinterpret_cast<const f2u4*>(lb);
            }
        }
        // ---- (b) the row pieces of the next step, straight into LDS (they overwrite rows below resLo only)
        int issued = 0;
        if (want > have && !(a.dbg & 1)) { issued = request(have + 1, want); have = want; }
        // ---- (c) weights and blend: the tap PAIRS of a pixel go through packed multiplies as they came from LDS (no register shuffling); the products and the order of
        // the sums are the reference's: t = p00 w0 + p01 w1; t += p10 w2; t += p11 w3
        float out[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const float fx = (float)(X[o] & 31) * (1.f / 32), fy = (float)(Y[o] & 31) * (1.f / 32);
            const float wy0 = 1.f - fy, wx0 = 1.f - fx;
            const f2v wx = {wx0, fx};
            const f2v wT = f2v{wy0, wy0} * wx, wBm = f2v{fy, fy} * wx;                        // (w0, w1), (w2, w3)
            const f2v a = pT[o] * wT, c = pB[o] * wBm;
            float t = a.x + a.y;
            t = t + c.x;
            t = t + c.y;
            out[o] = t;
        }
        // ---- (d) all taps of this step are in registers everywhere, the next step's pieces have landed.  (sched_barrier: the blend stays ABOVE the wait -- left to itself
        // the scheduler sinks it below the barrier, where nothing is in flight any more)
        asm volatile("" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]));       // (... and the optimiser does not sink it either: the four results exist here)
        __builtin_amdgcn_sched_barrier(0);
        // the pieces of step j + 1 (requested one step ago) must have landed; the ones just requested (for step j + 2) may stay in flight: loads complete in order, so
        // "at most `issued` operations outstanding" says exactly that (a store of the previous step still in flight only makes the wait longer)
        switch (late ? 0 : issued) {
        case 1:  __builtin_amdgcn_s_waitcnt(0x0071); break;                                     // vmcnt(n) lgkmcnt(0): the border values the lanes wrote themselves are in as well
        case 2:  __builtin_amdgcn_s_waitcnt(0x0072); break;
        case 3:  __builtin_amdgcn_s_waitcnt(0x0073); break;
        case 4:  __builtin_amdgcn_s_waitcnt(0x0074); break;
        case 5:  __builtin_amdgcn_s_waitcnt(0x0075); break;
        case 6:  __builtin_amdgcn_s_waitcnt(0x0076); break;
        default: __builtin_amdgcn_s_waitcnt(0x0070); break;
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- (e) stores.  Deferred lanes are left out: the wave marks its row piece (one flag byte per wave and row, written every time: no clearing pass) and
        // k_warp32_strip_rest redoes the marked pieces with the generic sampler.  (Sampling them here -- on the spot, or from a list after the walk -- was tried: a load
        // issued after the row pieces in flight and used before the barrier drains them all, and a wave that samples a few lanes serially holds its workgroup's other
        // seven: 93 and 134 us per 8K frame against 56 with those lanes left out, profiles/r06_warp32_strip.txt.)
        const bool anyDefer = __ballot(laneDefer && !(a.dbg & 8)) != 0ull;
        if (live && lane == 0) flags[((size_t)blockIdx.z * w.dh + y) * gridDim.x + blockIdx.x] = anyDefer ? 1 : 0;
        if (live && xl <= xe && !laneDefer && !(a.dbg & 2)) {
            uchar* drow = dst + (size_t)y * dstep;
            if (laneOut) { out[0] = out[1] = out[2] = out[3] = s.cval[0]; }
            if (xl + 3 <= xe) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(f4{out[0], out[1], out[2], out[3]}, reinterpret_cast<f4*>(drow + (size_t)xl * 4));
            } else {
                for (int o = 0; o < 4 && xl + o <= xe; o++) reinterpret_cast<float*>(drow)[xl + o] = out[o];
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                                        // nothing may still be on its way into this workgroup's LDS when it ends
}

// the row pieces k_warp32_strip marked (256 destination pixels of one row each): every pixel again, the footprints strictly inside the source by k_warp_lin's expression
// (the same products in the same order), the others by the generic sampler.  A wave reads 64 flags and walks the marked ones.
__global__ __launch_bounds__(256) void k_warp32_strip_rest(const uchar* __restrict__ src0, uint32_t sstep, uchar* __restrict__ dst0, uint32_t dstep, SampleArgs s, WarpArgs w,
                                                           const int* __restrict__ terms, const short* __restrict__ tab, const uchar* __restrict__ flags, int nstrips, int nframes)
{
    const int lane = threadIdx.x & 63;
    const size_t total = (size_t)nframes * w.dh * nstrips;
    const size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (base >= total) return;
    const size_t mine = base + lane;
    unsigned long long m = __ballot(mine < total && flags[mine] != 0);
    const int* colX = terms; const int* colY = terms + w.dw; const int* rowX = terms + 2 * w.dw; const int* rowY = rowX + w.dh;
    while (m) {
        const int k = __builtin_ctzll(m);
        m &= m - 1;
        const size_t id = base + k;
        const int strip = (int)(id % (size_t)nstrips), y = (int)((id / (size_t)nstrips) % (size_t)w.dh), fr = (int)(id / ((size_t)nstrips * w.dh));
        const uchar* src = src0 + (size_t)fr * w.sframe;
        uchar* drow = dst0 + (size_t)fr * w.dframe + (size_t)y * dstep;
        const int rX = rowX[y], rY = rowY[y];
        for (int q = 0; q < 4; q++) {
            const int x = strip * WS_COLS + 64 * q + lane;                                     // consecutive lanes on consecutive pixels
            if (x >= w.dw) break;
            const int X = (rX + colX[x]) >> 5, Y = (rY + colY[x]) >> 5;
            const int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
            if ((unsigned)sx < (unsigned)(s.sw - 1) && (unsigned)sy < (unsigned)(s.sh - 1)) {
                const float* g = reinterpret_cast<const float*>(src + (size_t)sy * sstep) + sx;
                const float* g1 = reinterpret_cast<const float*>(reinterpret_cast<const uchar*>(g) + sstep);
                const float s32 = 1.f / 32, fx = ax * s32, fy = ay * s32, wy0 = 1.f - fy, wx0 = 1.f - fx;
                const float w0 = __fmul_rn(wy0, wx0), w1 = __fmul_rn(wy0, fx), w2 = __fmul_rn(fy, wx0), w3 = __fmul_rn(fy, fx);
                float t = __fadd_rn(__fmul_rn(g[0], w0), __fmul_rn(g[1], w1));
                t = __fadd_rn(t, __fmul_rn(g1[0], w2));
                t = __fadd_rn(t, __fmul_rn(g1[1], w3));
                reinterpret_cast<float*>(drow)[x] = t;
            } else samplePixel(src, sstep, drow + (size_t)x * 4, s, satShort(sx), satShort(sy), ax, ay, tab);
        }
    }
}


// the affine coordinate terms of every destination column and row, once per call (k_warp8_tile reads them instead of redoing the double arithmetic per tile)
__global__ __launch_bounds__(256) void k_warp8_terms(warp8::Args a, int* __restrict__ colT, int* __restrict__ rowT, uint32_t* __restrict__ work)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && work) work[0] = 0;                                                     // the list of tiles k_warp8_lean1 leaves to k_warp8_tile_list
    if (i < a.dw) { colT[i] = warp8::affColX(a, i); colT[a.dw + i] = warp8::affColY(a, i); }
    if (i < a.dh) { rowT[i] = warp8::affRowX(a, i); rowT[a.dh + i] = warp8::affRowY(a, i); }
}

// ---- CV_8U bilinear warpAffine / warpPerspective through an LDS tile (warp8.h has the why and every phase; this is the launch geometry) ------------------
// A workgroup walks TPW horizontally adjacent 128 x th tiles; per tile: box terms by
// the first lanes -> barrier -> the source box into LDS + row / column terms -> barrier -> 16 (th = 32) or 8 destination pixels per thread.
template <int CN, int KIND, int FETCH>
__device__ __forceinline__ void warp8Tile(const uchar* __restrict__ src, uchar* __restrict__ dst, const SampleArgs& s, const warp8::Args& a, const short* __restrict__ tab,
                                          uchar* w8lds, int x0, int y0, int tid)
{
    warp8::phaseA<KIND>(a, x0, y0, w8lds, tid);
    __syncthreads();
    int terms[12];                                                                   // wave-uniform: the box arithmetic runs on the scalar unit
#pragma unroll
    for (int k = 0; k < (KIND == 0 ? 8 : 12); k++) terms[k] = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(w8lds + warp8::OFF_TERMS)[k]);
    const warp8::Box b = warp8::boxFromTerms<CN, KIND>(a, terms);
    warp8::phaseB<CN, KIND>(a, b, x0, y0, src, w8lds, tid);
    __syncthreads();
    const unsigned redo = warp8::phaseC<CN, KIND, FETCH>(a, b, x0, y0, w8lds, dst, tid);
    if (redo) {
        // what the tile path left: partial footprints on the source's rim, the border rules other than CONSTANT, BORDER_TRANSPARENT -- the generic sampler's
        warp8::redoGroups<CN, KIND>(a, b, redo, x0, y0, w8lds, tid, [&](int x, int y, int X, int Y) {
            samplePixel(src, a.sstep, dst + (size_t)y * a.dstep + (size_t)x * CN, s, satShort(X >> 5), satShort(Y >> 5), X & 31, Y & 31, tab);
        });
    }
}

template <int CN, int KIND, int FETCH>
__global__ __launch_bounds__(256) void k_warp8_tile(const uchar* __restrict__ src, uchar* __restrict__ dst, SampleArgs s, warp8::Args a, const short* __restrict__ tab, int tpw)
{
    extern __shared__ __align__(16) uchar w8lds[];
    src += (size_t)blockIdx.z * a.sframe; dst += (size_t)blockIdx.z * a.dframe;
    const int tid = threadIdx.x, y0 = blockIdx.y * a.th;
    for (int t = 0; t < tpw; t++) {
        const int tx = blockIdx.x * tpw + t;
        if (tx >= a.gx) break;                                                           // uniform
        if (t) __syncthreads();                                                          // the previous tile's terms and pixels are no longer read
        warp8Tile<CN, KIND, FETCH>(src, dst, s, a, tab, w8lds, tx * warp8::TW, y0, tid);
    }
}

// the same tile code over a list of tiles (work[0] = count, work[1..] = (frame * gy + ty) * gx + tx): what k_warp8_lean1 left -- the tiles on the source's rim
template <int CN, int KIND, int FETCH>
__global__ __launch_bounds__(256) void k_warp8_tile_list(const uchar* __restrict__ src, uchar* __restrict__ dst, SampleArgs s, warp8::Args a, const short* __restrict__ tab,
                                                         const uint32_t* __restrict__ work)
{
    extern __shared__ __align__(16) uchar w8lds[];
    const uint32_t count = work[0];
    const int tid = threadIdx.x;
    bool first = true;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {                           // uniform
        const uint32_t id = work[1 + i];
        const uint32_t tx = id % (uint32_t)a.gx, rest = id / (uint32_t)a.gx, ty = rest % (uint32_t)a.gy, f = rest / (uint32_t)a.gy;
        if (!first) __syncthreads();
        first = false;
        warp8Tile<CN, KIND, FETCH>(src + (size_t)f * a.sframe, dst + (size_t)f * a.dframe, s, a, tab, w8lds, (int)tx * warp8::TW, (int)ty * a.th, tid);
    }
}

// bicubic warpAffine of CV_8U images on the tile path of warp8.h (boxFromTerms4 / phaseC4): persistent workgroups -- the Q15 weight table is copied into LDS
// once, 12 dwords per entry (8 used: the 48-byte pitch spreads the entries over the banks) --, then tile after tile: box terms, the source box staged with aligned
// dwords, four destination pixels per lane from LDS, what is left (the source's rim, border rules) through samplePixelN.  The default for one channel since round 5
// (see runWarp; tests/hostemu also runs its phases on the CPU against the restatement).
template <int CN>
__global__ __launch_bounds__(256) void k_warp8_cubic(const uchar* __restrict__ src, uchar* __restrict__ dst, SampleArgs s, warp8::Args a, const short* __restrict__ tabI,
                                                     const float* __restrict__ tab1, uint32_t tileBytes, int nframes)
{
    extern __shared__ __align__(16) uchar w8lds[];
    uint32_t* wq = reinterpret_cast<uint32_t*>(w8lds + tileBytes);
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024 * 8; i += 256) wq[(i >> 3) * 12 + (i & 7)] = reinterpret_cast<const uint32_t*>(tabI)[i];
    const uint32_t perFrame = (uint32_t)a.gx * (uint32_t)a.gy, total = perFrame * (uint32_t)nframes;
    for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {                           // uniform
        const uint32_t f = t / perFrame, tt = t - f * perFrame, ty = tt / (uint32_t)a.gx, tx = tt - ty * (uint32_t)a.gx;
        const uchar* S = src + (size_t)f * a.sframe; uchar* D = dst + (size_t)f * a.dframe;
        const int x0 = (int)tx * warp8::TW, y0 = (int)ty * a.th;
        __syncthreads();                                                                 // the table is there / the previous tile's terms and pixels are no longer read
        warp8::phaseA<0>(a, x0, y0, w8lds, tid);
        __syncthreads();
        int terms[8];
#pragma unroll
        for (int k = 0; k < 8; k++) terms[k] = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(w8lds + warp8::OFF_TERMS)[k]);
        const warp8::Box b = warp8::boxFromTerms4<CN>(a, terms);
        warp8::phaseB<CN, 0>(a, b, x0, y0, S, w8lds, tid);
        __syncthreads();
        const unsigned redo = warp8::phaseC4<CN>(a, b, x0, y0, w8lds, wq, 12, D, tid);
        if (redo)
            warp8::redoGroups<CN, 0>(a, b, redo, x0, y0, w8lds, tid, [&](int x, int y, int X, int Y) {
                samplePixelN<4>(S, a.sstep, D + (size_t)y * a.dstep + (size_t)x * CN, s, satShort(X >> 5), satShort(Y >> 5), X & 31, Y & 31, tabI, tab1);
            });
    }
}

// the lean path of warp8.h: one or three channels, affine.  A workgroup walks `tpw` tiles of one tile row with the NEXT tile's box in flight in registers while
// it samples the current one (two LDS buffers, one barrier per tile); the tiles it cannot take (the source's rim under a border rule other than CONSTANT, boxes
// beyond its staging geometry) go on a list for k_warp8_tile_list.
template <int CN, int LW, int NR>
__global__ __launch_bounds__(256) void k_warp8_lean(const uchar* __restrict__ src, uchar* __restrict__ dst, warp8::Args a, int tpw, uint32_t* __restrict__ work)
{
    extern __shared__ __align__(16) uchar w8lds[];
    src += (size_t)blockIdx.z * a.sframe; dst += (size_t)blockIdx.z * a.dframe;
    const int tid = threadIdx.x, y0 = blockIdx.y * a.th, tx0 = blockIdx.x * tpw, n = min(tpw, a.gx - tx0);
    warp8::LeanRowT rt;
    warp8::leanRowTerms<CN>(a, y0, tid, rt);
    uint32_t v[NR];
    auto load = [&](const warp8::LBox& b) {
        if (b.kind == warp8::LEAN_INSIDE) warp8::leanLoad<CN, LW, NR, false>(a, b, src, tid, v);
        else if (b.kind == warp8::LEAN_RIM) warp8::leanLoad<CN, LW, NR, true>(a, b, src, tid, v);
    };
    warp8::LBox b = warp8::leanClassify<CN>(a, tx0 * warp8::TW, y0);
    load(b);
    for (int t = 0; t < n; t++) {                                                        // uniform
        uchar* buf = w8lds + (t & 1) * a.leanBuf;
        const int x0 = (tx0 + t) * warp8::TW;
        if (b.kind == warp8::LEAN_INSIDE || b.kind == warp8::LEAN_RIM) warp8::leanStore<CN, LW, NR>(a, b, buf, tid, v);
        else if (b.kind == warp8::LEAN_NO && tid == 0) work[1 + atomicAdd(work, 1u)] = ((uint32_t)blockIdx.z * (uint32_t)a.gy + blockIdx.y) * (uint32_t)a.gx + (uint32_t)(tx0 + t);
        warp8::LBox bn = b; bn.kind = warp8::LEAN_NO;
        if (t + 1 < n) { bn = warp8::leanClassify<CN>(a, x0 + warp8::TW, y0); load(bn); }  // in flight across the barrier and the sampling below
        __syncthreads();                                                                 // tile t is in buf; tile t - 1's readers are past their sampling
        if (b.kind == warp8::LEAN_INSIDE) warp8::leanRows<CN, false>(a, b, x0, y0, buf, dst, tid, rt);
        else if (b.kind == warp8::LEAN_RIM) warp8::leanRows<CN, true>(a, b, x0, y0, buf, dst, tid, rt);
        else if (b.kind == warp8::LEAN_OUTSIDE) warp8::leanFill<CN>(a, x0, y0, dst, tid);
        b = bn;
    }
}

// ---- CV_8U bilinear resize through the lean kernel's LDS pipeline (warp8.h "bilinear resize ... on the lean kernel's machinery") ------------------------------
__global__ __launch_bounds__(256) void k_resize8_terms(warp8::Args a, int* __restrict__ colT, int* __restrict__ rowT)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.dw) { int sx; uint32_t a01; warp8::rzColTerm(a, i, sx, a01); colT[i] = sx; colT[a.dw + i] = (int)a01; }
    if (i < a.dh) { int y0, y1; uint32_t b0, b1; warp8::rzRowTerm(a, i, y0, y1, b0, b1); rowT[i] = y0; rowT[a.dh + i] = y1; rowT[2 * a.dh + i] = (int)b0; rowT[3 * a.dh + i] = (int)b1; }
}

template <int CN, int LW, int NR>
__global__ __launch_bounds__(256) void k_resize8_lean(const uchar* __restrict__ src, uchar* __restrict__ dst, warp8::Args a, int tpw)
{
    extern __shared__ __align__(16) uchar w8lds[];
    src += (size_t)blockIdx.z * a.sframe; dst += (size_t)blockIdx.z * a.dframe;
    const int tid = threadIdx.x, y0 = blockIdx.y * a.th, tx0 = blockIdx.x * tpw, n = min(tpw, a.gx - tx0);
    warp8::RzRowT rt;
    warp8::rzRowTerms<CN>(a, y0, tid, rt);
    uint32_t v[NR];
    auto load = [&](const warp8::LBox& b) {
        if (b.kind == warp8::LEAN_INSIDE) warp8::leanLoad<CN, LW, NR, false>(a, b, src, tid, v);
        else if (b.kind == warp8::LEAN_RIM) warp8::leanLoad<CN, LW, NR, true>(a, b, src, tid, v);
    };
    warp8::LBox b = warp8::rzClassify<CN>(a, tx0 * warp8::TW, y0);
    load(b);
    for (int t = 0; t < n; t++) {                                                        // uniform
        uchar* buf = w8lds + (t & 1) * a.leanBuf;
        const int x0 = (tx0 + t) * warp8::TW;
        if (b.kind != warp8::LEAN_NO) warp8::leanStore<CN, LW, NR>(a, b, buf, tid, v);
        warp8::LBox bn = b; bn.kind = warp8::LEAN_NO;
        if (t + 1 < n) { bn = warp8::rzClassify<CN>(a, x0 + warp8::TW, y0); load(bn); }
        __syncthreads();
        if (b.kind != warp8::LEAN_NO) warp8::rzRows<CN>(a, b, x0, y0, buf, dst, tid, rt);
        b = bn;
    }
}

bool depthOk(int d) { return d == D8U || d == D16U || d == D16S || d == D32F; }

int runWarp(const char* entry, int src_type, const uchar* src, size_t sstep, int sw, int sh, uchar* dst, size_t dstep, int dw, int dh,
            const double* M, int kind, int interpolation, int borderType, const double* bv,
            const float* mapx, size_t mxstep, const float* mapy, size_t mystep, int nframes = 1, size_t sframe = 0, size_t dframe = 0)
{
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    const int depth = MI355CV_MAT_DEPTH(src_type), cn = MI355CV_MAT_CN(src_type);
    const bool is64 = depth == MI355CV_64F;                                                      // CV_64F images: the per-pixel kernel k_warp64
    // channels: the samplers loop over them (border value of channel k = borderValue[k & 3], imgwarp.cpp:340 / :692); the reference itself asserts <= 4 for bicubic / Lanczos
    // (imgwarp.cpp:2795), and so does this entry further down
    if (!(depthOk(depth) || is64) || cn < 1 || cn > 512 || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return mi355::declined(__func__, __LINE__, "depth is none of 8U / 16U / 16S / 32F / 64F || cn < 1 || cn > 512 || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0");
    if (is64 && (kind == 6 || kind == 7)) return mi355::declined(__func__, __LINE__, "warpPolar on CV_64F images");
    const bool relative = (interpolation & 32) != 0 && kind >= 2 && kind <= 5;              // WARP_RELATIVE_MAP (cv::remap only, imgwarp.cpp:1724)
    if (relative) interpolation &= ~32;
    if (interpolation == MI355CV_INTER_AREA) interpolation = MI355CV_INTER_LINEAR;          // imgwarp.cpp:2818
    if (interpolation != MI355CV_INTER_NEAREST && interpolation != MI355CV_INTER_LINEAR && interpolation != MI355CV_INTER_CUBIC && interpolation != MI355CV_INTER_LANCZOS4)
        return mi355::declined(__func__, __LINE__, "interpolation is none of NEAREST, LINEAR, CUBIC, AREA, LANCZOS4");
    const bool taps = interpolation == MI355CV_INTER_CUBIC || interpolation == MI355CV_INTER_LANCZOS4;
    if (taps && cn > 4) return mi355::declined(__func__, __LINE__, "bicubic / Lanczos sampling of more than 4 channels (the reference asserts on it)");
    if (taps && (kind == 5 || kind == 6 || kind == 7)) return mi355::declined(__func__, __LINE__, "bicubic / Lanczos sampling with a CV_16SC2 map alone, or in warpPolar");
    if (borderType < 0 || borderType > B_TRANSPARENT) return mi355::declined(__func__, __LINE__, "borderType < 0 || borderType > B_TRANSPARENT");
    if (sw > 32767 || sh > 32767) return mi355::declined(__func__, __LINE__, "sw > 32767 || sh > 32767");                         // coordinates saturate to short in the reference
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src, (size_t)dw * dh, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src, (size_t)dw * dh, minPixels(HOST_HEAVY))");
    if (nframes < 1 || (nframes > 1 && (!isDevicePtr(src) || !isDevicePtr(dst))))
        return setError(MI355CV_NOT_IMPLEMENTED, "%s: batch entry needs device-resident frames", entry);
    const short* g_tabDev = deviceTab();
    if (!g_tabDev) return mi355::declined(__func__, __LINE__, "!g_tabDev");
    const int e = is64 ? 8 : eszOf(depth);
    size_t dss, dds, mxs = mxstep, mys = mystep;
    const uchar* ds = stg.in(src, sstep, (size_t)sw * cn * e, sh, &dss);
    uchar* dd;
    if (borderType == B_TRANSPARENT && !isDevicePtr(dst)) {
        // untouched pixels must keep their previous contents: stage dst in as well
        const uchar* din = stg.in(dst, dstep, (size_t)dw * cn * e, dh, &dds);
        size_t dds2; dd = stg.out(dst, dstep, (size_t)dw * cn * e, dh, &dds2);
        if (!din || !dd) return mi355::declined(__func__, __LINE__, "!din || !dd");
        if (hipMemcpy2DAsync(dd, dds2, din, dds, (size_t)dw * cn * e, dh, hipMemcpyDeviceToDevice, stream()) != hipSuccess) return mi355::declined(__func__, __LINE__, "hipMemcpy2DAsync(dd, dds2, din, dds, (size_t)dw * cn * e, dh, hipMemcpyDeviceToDevice, stream()) != hipSuccess");
        dds = dds2;
    } else dd = stg.out(dst, dstep, (size_t)dw * cn * e, dh, &dds);
    const uchar* dmx = nullptr; const uchar* dmy = nullptr;
    if (kind == 2) {
        dmx = stg.in((const uchar*)mapx, mxstep, (size_t)dw * 4, dh, &mxs);
        dmy = stg.in((const uchar*)mapy, mystep, (size_t)dw * 4, dh, &mys);
        if (!dmx || !dmy) return mi355::declined(__func__, __LINE__, "!dmx || !dmy");
    } else if (kind == 3) {                                                                  // one CV_32FC2 map
        dmx = stg.in((const uchar*)mapx, mxstep, (size_t)dw * 8, dh, &mxs);
        if (!dmx) return mi355::declined(__func__, __LINE__, "!dmx");
    } else if (kind == 4 || kind == 5) {                                                     // CV_16SC2 (+ CV_16UC1 fractions)
        dmx = stg.in((const uchar*)mapx, mxstep, (size_t)dw * 4, dh, &mxs);
        if (kind == 4) dmy = stg.in((const uchar*)mapy, mystep, (size_t)dw * 2, dh, &mys);
        if (!dmx || (kind == 4 && !dmy)) return mi355::declined(__func__, __LINE__, "!dmx || (kind == 4 && !dmy)");
    } else if (kind == 7) {                                                                  // inverse warpPolar: the 512-float log table (host array)
        dmx = (const uchar*)stg.param(mapx, 512 * sizeof(float));
        if (!dmx) return mi355::declined(__func__, __LINE__, "!dmx");
    } else if (kind == 6) {                                                                  // warpPolar tables (host arrays: dw radii, dh (cos, sin) pairs)
        dmx = (const uchar*)stg.param(mapx, (size_t)dw * 4);
        dmy = (const uchar*)stg.param(mapy, (size_t)dh * 16);
        if (!dmx || !dmy) return mi355::declined(__func__, __LINE__, "!dmx || !dmy");
    }
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    SampleArgs s; s.sw = sw; s.sh = sh; s.depth = depth; s.cn = cn; s.linear = interpolation == MI355CV_INTER_LINEAR ? 1 : taps ? interpolation : 0; s.border = borderType;
    for (int k = 0; k < 4; k++) { s.cval[k] = bv ? (float)bv[k] : 0.f; s.cvalD[k] = bv ? bv[k] : 0.0; }
    WarpArgs w; memset(&w, 0, sizeof w);
    w.dw = dw; w.dh = dh; w.kind = kind; w.rel = relative ? 1 : 0;
    w.sframe = nframes > 1 ? sframe : 0; w.dframe = nframes > 1 ? dframe : 0;
    if (M) for (int i = 0; i < (kind == 0 ? 6 : kind == 6 ? 2 : kind == 7 ? 5 : 9); i++) w.M[i] = M[i];
    int bh0 = dh < 16 ? dh : 16;
    w.bw0 = 1024 / bh0 < dw ? 1024 / bh0 : dw;                                              // WarpPerspectiveInvoker :3182-3184
    { static const int pe = [] { const char* v = getenv("MI355CV_PERSP_EXACT"); return v ? atoi(v) : 0; }(); w.pexact = pe; }
    if (is64) {
        const TapTabs* tt = deviceTapTabs();
        if (!tt) return mi355::declined(__func__, __LINE__, "the bicubic / Lanczos weight tables could not be placed on the device");
        dim3 grid(divUp(dw, 64), divUp(dh, 4), nframes);
        hipLaunchKernelGGL(k_warp64, grid, dim3(256), 0, stream(), ds, dss, dd, dds, s, w, tt->cubic1, tt->lanczos1, dmx, mxs, dmy, mys);
        noteKernel("k_warp64 grid=%ux%ux%u x256 kind=%d mode=%d", grid.x, grid.y, grid.z, kind, s.linear);
        return stg.finish(entry);
    }
    if (taps) {
        // bicubic / Lanczos: the per-pixel kernel over the shared coordinate generation
        const TapTabs* tt = deviceTapTabs();
        if (!tt) return mi355::declined(__func__, __LINE__, "the bicubic / Lanczos weight tables could not be placed on the device");
        // the LDS form (MI355CV_WARP_TAPS_LDS=0 keeps the plain per-pixel kernel for A/B runs): 1-, 3- and 4-channel images whose rows can be read as unaligned dwords
        // affine maps: the column / row terms once per call
        int* terms = nullptr;
        if (kind == 0) {
            terms = (int*)stg.scratch((size_t)(2 * dw + 2 * dh) * sizeof(int));
            uint32_t* work0 = (uint32_t*)stg.scratch(sizeof(uint32_t));
            if (!terms || !work0) return mi355::declined(__func__, __LINE__, "scratch for the coordinate terms");
            hipLaunchKernelGGL(k_warp32_terms, dim3(divUp(std::max(dw, dh), 256)), dim3(256), 0, stream(), w, terms, work0);
        }
        // CV_8U bicubic on the tile path of warp8.h.  Round 5, first run on the GPU (profiles/r05_warp_taps_tile_call1.txt, 4K, 7 degrees): one channel 42.7 us against the
        // tap-row kernel's 52.8-54.1 -- the default from here --, three channels 113.9 against 85.6 -- stays on the tap-row kernel.  MI355CV_WARP_TAPS_TILE=0 / 1: never / both.
        static const int tapsTile = [] { const char* v = getenv("MI355CV_WARP_TAPS_TILE"); return v ? atoi(v) : -1; }();
        if ((tapsTile < 0 ? cn == 1 : tapsTile != 0) && kind == 0 && depth == D8U && interpolation == MI355CV_INTER_CUBIC && (cn == 1 || cn == 3) && sw >= 4 && sh >= 4 && terms) {
            warp8::Args a8; size_t lds8 = 0;
            if (warp8::plan(a8, cn, 0, M, sw, sh, dw, dh, dss, dds, ds, dd, w.bw0, &lds8, 3)) {
                a8.sframe = w.sframe; a8.dframe = w.dframe;
                a8.constBorder = borderType == B_CONSTANT;
                for (int k = 0; k < cn; k++) a8.cval |= (uint32_t)fminf(fmaxf(rintf(s.cval[k]), 0.f), 255.f) << (8 * k);
                a8.colT = terms; a8.rowT = terms + 2 * dw;
                a8.leanLW = 0;
                const uint32_t tileBytes = (uint32_t)((lds8 + 15) & ~(size_t)15);
                const size_t ldsAll = (size_t)tileBytes + 1024 * 12 * 4;
                const long long ntiles = (long long)a8.gx * a8.gy * nframes;
                const unsigned gridN = (unsigned)std::min<long long>(ntiles, 256LL * (ldsAll <= 80 * 1024 ? 2 : 1));
                if (cn == 1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_warp8_cubic<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsAll);
                               hipLaunchKernelGGL(k_warp8_cubic<1>, dim3(gridN), dim3(256), ldsAll, stream(), ds, dd, s, a8, tt->cubicI, tt->cubic1, tileBytes, nframes); }
                else         { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_warp8_cubic<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsAll);
                               hipLaunchKernelGGL(k_warp8_cubic<3>, dim3(gridN), dim3(256), ldsAll, stream(), ds, dd, s, a8, tt->cubicI, tt->cubic1, tileBytes, nframes); }
                noteKernel("k_warp8_cubic<%d> grid=%u x256 lds=%zu (tile %u + table) tiles=%dx%dx%d box<=%dx%d", cn, gridN, ldsAll, tileBytes, a8.gx, a8.gy, nframes,
                           (a8.ldsPitch - 8) / cn, a8.ldsRows);
                return stg.finish(entry);
            }
        }
        static const bool tapsLds = [] { const char* v = getenv("MI355CV_WARP_TAPS_LDS"); return !v || atoi(v) != 0; }();
        const bool lanc = interpolation == MI355CV_INTER_LANCZOS4;
        if (tapsLds && (cn == 1 || cn == 3 || cn == 4) && sw >= (lanc ? 8 : 4) && sh >= (lanc ? 8 : 4)) {
            const int ks = lanc ? 8 : 4;
            const bool u8 = depth == D8U;
            const int block = (u8 && lanc) ? 512 : 256, rows = block / 64;
            const size_t lds = (size_t)32 * ks * 4 + (u8 ? (size_t)1024 * (ks * ks * 2 + 16) : 0);      // the Q15 entries padded by 16 bytes (bank spread)
            // pixels per thread (affine maps only: their coordinates are two table reads): by the registers the tap rows take -- 4 where a pixel's rows are <= 10 dwords
            // (bicubic CV_8UC1), 2 up to 20 (bicubic CV_8UC3 / CV_8UC4 / CV_32FC1, Lanczos CV_8UC1)
            const int esz = u8 ? 1 : depth == D32F ? 4 : 2, ksnb = ks * (ks * cn * esz / 4);
            const int ppt = (!terms || (depth != D8U && depth != D32F)) ? 1 : ksnb <= 10 ? 4 : ksnb <= 20 ? 2 : 1;
            const int tilesX = divUp(dw, 64), tilesY = divUp(dh, rows * ppt);
            const long long total = (long long)tilesX * tilesY * nframes;
            const int perCU = u8 ? (lanc ? 1 : 3) : 8;                                  // workgroups a CU holds (LDS for CV_8U, waves otherwise); 256 CUs
            const unsigned gridN = (unsigned)std::min<long long>(total, 256LL * perCU);
            const short* tI = lanc ? tt->lanczosI : tt->cubicI; const float* t1 = lanc ? tt->lanczos1 : tt->cubic1;
            // one flag byte per 64-pixel row strip: set by the LDS kernel where pixels next to the border were left out
            const size_t strips = (size_t)tilesX * dh * nframes;
            uchar* stripFlag = (uchar*)stg.scratch(strips);
            if (!stripFlag || hipMemsetAsync(stripFlag, 0, strips, stream()) != hipSuccess) return mi355::declined(__func__, __LINE__, "scratch for the border-strip flags");
#define WTA(KS_, DEP_, CN_, BLK_, PP_) do { \
                if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_warp_taps_lds<KS_, DEP_, CN_, BLK_, PP_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                hipLaunchKernelGGL((k_warp_taps_lds<KS_, DEP_, CN_, BLK_, PP_>), dim3(gridN), dim3(BLK_), lds, stream(), ds, dss, dd, dds, s, w, tI, t1, dmx, mxs, dmy, mys, tilesX, tilesY, nframes, \
                                   terms, stripFlag); } while (0)
#define WT1(KS_, DEP_, BLK_) do { if (cn == 1) WTA(KS_, DEP_, 1, BLK_, 1); else if (cn == 3) WTA(KS_, DEP_, 3, BLK_, 1); else WTA(KS_, DEP_, 4, BLK_, 1); } while (0)
            if (ppt == 4)      WTA(4, D8U, 1, 256, 4);                                                                  // bicubic CV_8UC1
            else if (ppt == 2) {
                if (lanc) WTA(8, D8U, 1, 512, 2);                                                                       // Lanczos CV_8UC1
                else if (u8) { if (cn == 3) WTA(4, D8U, 3, 256, 2); else WTA(4, D8U, 4, 256, 2); }                      // bicubic CV_8UC3 / CV_8UC4
                else WTA(4, D32F, 1, 256, 2);                                                                           // bicubic CV_32FC1
            }
            else if (lanc) { if (u8) WT1(8, D8U, 512); else if (depth == D16U) WT1(8, D16U, 256); else if (depth == D16S) WT1(8, D16S, 256); else WT1(8, D32F, 256); }
            else           { if (u8) WT1(4, D8U, 256); else if (depth == D16U) WT1(4, D16U, 256); else if (depth == D16S) WT1(4, D16S, 256); else WT1(4, D32F, 256); }
#undef WT1
#undef WTA
            const unsigned gridL = (unsigned)((strips + 15) / 16);
            if (lanc) hipLaunchKernelGGL(k_warp_taps_strips<8>, dim3(gridL), dim3(256), 0, stream(), ds, dss, dd, dds, s, w, tI, t1, dmx, mxs, dmy, mys, tilesX, nframes, terms, stripFlag);
            else      hipLaunchKernelGGL(k_warp_taps_strips<4>, dim3(gridL), dim3(256), 0, stream(), ds, dss, dd, dds, s, w, tI, t1, dmx, mxs, dmy, mys, tilesX, nframes, terms, stripFlag);
            noteKernel("k_warp_taps_lds<%d,depth %d,cn %d,block %d,%d px per thread> grid=%u lds=%zu tiles=%dx%dx%d kind=%d + k_warp_taps_strips<%d> grid=%u (row strips next to the border)",
                       ks, depth, cn, block, ppt, gridN, lds, tilesX, tilesY, nframes, kind, ks, gridL);
            return stg.finish(entry);
        }
        dim3 grid(divUp(dw, 64), divUp(dh, 4), nframes);
        if (interpolation == MI355CV_INTER_CUBIC) hipLaunchKernelGGL(k_warp_taps<4>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, s, w, tt->cubicI, tt->cubic1, dmx, mxs, dmy, mys, 0, terms);
        else                                      hipLaunchKernelGGL(k_warp_taps<8>, grid, dim3(256), 0, stream(), ds, dss, dd, dds, s, w, tt->lanczosI, tt->lanczos1, dmx, mxs, dmy, mys, 0, terms);
        noteKernel("k_warp_taps<%d> grid=%ux%ux%u x256 kind=%d", interpolation == MI355CV_INTER_CUBIC ? 4 : 8, grid.x, grid.y, grid.z, kind);
        return stg.finish(entry);
    }
    if ((kind == 0 || kind == 1) && s.linear && (cn == 1 || cn == 3 || cn == 4) && (depth == D32F || depth == D8U) && sw >= 3 && sh >= 2 && (dss % e) == 0 &&
        (unsigned long long)sh * dss < (1ull << 32) && (unsigned long long)dh * dds < (1ull << 32) && dss < (1u << 24) &&
        ((uintptr_t)ds % e) == 0) {
        // CV_8U: the LDS-tile kernel where it applies (4-byte aligned images, a source box that fits its LDS allotment); MI355CV_WARP8=0 keeps the
        // thread-per-column kernel (A/B runs, tools/warp_probe.py)
        static const bool warp8On = [] { const char* v = getenv("MI355CV_WARP8"); return !v || atoi(v) != 0; }();
        warp8::Args a8; size_t lds8 = 0;
        // measured (profiles/r03_warp8.txt): the tile kernel wins for affine maps of 1- and 3-channel images (4K 8UC1 7 degrees 20.5 -> ~13 us per frame, 90 degrees
        // 27 -> 20; 8UC3 33 -> 31 / 45 -> 27); 4-channel images and perspective maps (a double division per pixel either way) stay on the gather kernel
        static const int warp8All = [] { const char* v = getenv("MI355CV_WARP8"); return v ? atoi(v) : 1; }();      // 2: every case the plan accepts (A/B runs)
        if (depth == D8U && warp8On && (warp8All == 2 || (kind == 0 && cn != 4)) && warp8::plan(a8, cn, kind, M, sw, sh, dw, dh, dss, dds, ds, dd, w.bw0, &lds8)) {
            a8.sframe = w.sframe; a8.dframe = w.dframe;
            a8.constBorder = borderType == B_CONSTANT;
            static const bool leanOn = [] { const char* v = getenv("MI355CV_WARP8_LEAN"); return !v || atoi(v) != 0; }();
            uint32_t* work = nullptr;                                                        // tiles the lean kernel leaves to the general one: [0] = count, then ids
            if (kind == 0) {
                int* tt = (int*)stg.scratch((size_t)(2 * dw + 2 * dh) * sizeof(int));
                if (tt && leanOn && (cn == 1 || cn == 3) && a8.leanLW) work = (uint32_t*)stg.scratch(((size_t)a8.gx * a8.gy * nframes + 1) * sizeof(uint32_t));
                if (tt) {
                    hipLaunchKernelGGL(k_warp8_terms, dim3(divUp(std::max(dw, dh), 256)), dim3(256), 0, stream(), a8, tt, tt + 2 * dw, work);
                    a8.colT = tt; a8.rowT = tt + 2 * dw;
                }
            }
            for (int k = 0; k < cn; k++) a8.cval |= (uint32_t)fminf(fmaxf(rintf(s.cval[k]), 0.f), 255.f) << (8 * k);
            static const int tpw = [] { const char* v = getenv("MI355CV_WARP8_TPW"); const int t = v ? atoi(v) : 1; return t < 1 ? 1 : t > 64 ? 64 : t; }();
            static const int fetch = [] { const char* v = getenv("MI355CV_WARP8_FETCH"); return v ? atoi(v) : 1; }();       // tap fetch form (warp8.h bilinearAt), A/B runs
            dim3 g8(divUp(a8.gx, tpw), a8.gy, nframes);
            constexpr int leanTpw = 6;                                 // tiles a workgroup walks (2 .. 12 measured within 2 %, profiles/r03_resize8_lean.txt)
            if (work) {
                dim3 gl(divUp(a8.gx, leanTpw), a8.gy, nframes);
                const size_t ldsL = 2 * (size_t)a8.leanBuf;
#define WLN(CN_, LW_, NR_) if (cn == CN_ && a8.leanLW == LW_ && a8.leanNR == NR_) hipLaunchKernelGGL((k_warp8_lean<CN_, LW_, NR_>), gl, dim3(256), ldsL, stream(), ds, dd, a8, leanTpw, work)
#define WLS(CN_, LW_) WLN(CN_, LW_, 6); WLN(CN_, LW_, 10); WLN(CN_, LW_, 14); WLN(CN_, LW_, 20)
                WLS(1, 16); WLS(1, 32); WLS(1, 64); WLS(3, 32); WLS(3, 64); WLS(3, 128); WLS(3, 256);
#undef WLS
#undef WLN
                const unsigned nl = (unsigned)std::min<size_t>((size_t)a8.gx * a8.gy * nframes, 256 * 5);
#define WTL(CN_, F_) hipLaunchKernelGGL((k_warp8_tile_list<CN_, 0, F_>), dim3(nl), dim3(256), lds8, stream(), ds, dd, s, a8, g_tabDev, work)
                if (cn == 1) { if (fetch) WTL(1, 1); else WTL(1, 0); } else { if (fetch) WTL(3, 1); else WTL(3, 0); }
#undef WTL
                noteKernel("k_warp8_lean<%d,%d,%d> grid=%ux%ux%u x256 tpw=%d lds=%zu (BORDER_CONSTANT: every tile; else the tiles inside the source) + k_warp8_tile_list<%d,0,%d> grid=%u (what it left) box<=%dx%d",
                           cn, a8.leanLW, a8.leanNR, gl.x, gl.y, gl.z, leanTpw, ldsL, cn, fetch, nl, (a8.ldsPitch - 8) / cn, a8.ldsRows);
                return stg.finish(entry);
            }
#define W8(CN_, K_, F_) hipLaunchKernelGGL((k_warp8_tile<CN_, K_, F_>), g8, dim3(256), lds8, stream(), ds, dd, s, a8, g_tabDev, tpw)
            if (kind == 0) { if (cn == 1) { if (fetch) W8(1, 0, 1); else W8(1, 0, 0); } else if (cn == 3) { if (fetch) W8(3, 0, 1); else W8(3, 0, 0); } else W8(4, 0, 0); }
            else           { if (cn == 1) { if (fetch) W8(1, 1, 1); else W8(1, 1, 0); } else if (cn == 3) { if (fetch) W8(3, 1, 1); else W8(3, 1, 0); } else W8(4, 1, 0); }
#undef W8
            noteKernel("k_warp8_tile<%d,%d,%d> grid=%ux%ux%u x256 tpw=%d lds=%zu box<=%dx%d", cn, kind, cn != 4 ? fetch : 0, g8.x, g8.y, g8.z, tpw, lds8, (a8.ldsPitch - 8) / cn, a8.ldsRows);
            return stg.finish(entry);
        }
        // XCD-banded tile order: off by default.  It paid 3 % on CV_32F while the kernel was bound by its own instruction count; with the lean
        // kernel the plain order is faster for small rotations (8K 32F, 7 degrees: 63.5 vs 71.0 us) and within 3 % otherwise.
        // MI355CV_WARP_BAND=1 turns it on (tools/warp_probe.py).
        const char* ve = getenv("MI355CV_WARP_BAND");
        w.band = ve ? atoi(ve) : 0;
        // CV_32FC1 affine: the LDS-tile kernel (MI355CV_WARP32=0 keeps the gather kernel for A/B runs)
        // Measured on 8K frames (profiles/r04_warp32_ab.txt): the LDS-tile kernel wins where a wave's 64 destination pixels spread over many source rows (33 degrees: 76 us
        // against the gather kernel's 90; 90 degrees likewise), the gather kernel where they stay within a few rows (7 degrees, shifts: 72-75 against 79) and wherever the
        // source box of a tile exceeds the LDS allotment (minification by ~1.5 and more: every tile would go to the list).  MI355CV_WARP32 = 0 / 1 force either kernel.
        // CV_32FC1 affine maps that are gentle enough (|rotation| up to ~12 degrees at scale ~1): the strip walk over an LDS ring of whole source row pieces (k_warp32_strip).
        // MI355CV_WARP32_STRIP=0 keeps the kernels below (A/B runs); the conditions are what the ring can hold -- everything else is decided per pixel inside the kernel
        static const int stripEnv = [] { const char* v = getenv("MI355CV_WARP32_STRIP"); return v ? atoi(v) : 1; }();
        if (stripEnv && kind == 0 && depth == D32F && cn == 1 && borderType == B_CONSTANT && M[4] > 0.25 && (sw & 3) == 0 && dw >= 64 && dh >= 16 &&
            ((((uintptr_t)ds) | dss | w.sframe | ((uintptr_t)dd) | dds | w.dframe) & 15) == 0) {
            const double ma = M[0], mb = M[1], md = M[3], me = M[4];
            const double h = mb / me, gcoef = ma - mb * md / me;
            // source rows one step reads + the next step's, and the width of a row piece: |M3| 255 + 1 rows across the strip, M4 rows per destination row (16 of them), the second
            // tap row; |g| 255 columns across the strip + the drift 2 |h| between the rows that share a source row + the second tap column + alignment + slack
            const double rowsNeeded = std::fabs(md) * 255 + me * (2 * WS_WAVES) + 4, colsNeeded = std::fabs(gcoef) * 255 + 2 * std::fabs(h) + 2 + 3 + 3;
            if (rowsNeeded <= WS_NR - 2 && colsNeeded <= WS_PW && std::fabs(h) < 0.5 && std::fabs(gcoef) * sw < 1e6 && std::fabs(M[2] - mb * M[5] / me) < 1e6) {
                StripArgs a;
                static const int pitchEnv = [] { const char* v = getenv("MI355CV_WARP32_PITCH"); const int p = v ? atoi(v) : 292; return p < WS_PW ? WS_PW : (p + 3) & ~3; }();
                a.pitch = pitchEnv;
                static const int dbgEnv = [] { const char* v = getenv("MI355CV_WARP32_DBG"); return v ? atoi(v) : 0; }();
                a.dbg = dbgEnv;
                a.H15 = (int)std::lrint(h * 32768.0);
                a.g = gcoef;
                a.cp = (M[2] - mb * M[5] / me) - std::fabs(h) - 2.0;            // sx >= g x + h sy + cp on every pixel of the strip that reads source row sy (2 = rounding of the 1/32 grid, of H15, slack)
                const int strips = divUp(dw, WS_COLS);
                const long long per = (long long)strips * nframes;
                // segments per strip: ~272 rows each.  Measured on 8 x 8K frames (profiles/r06_warp32_strip.txt): 2 / 4 / 6 / 8 / 16 / 24 / 32 / 48 segments per strip =
                // 62.9 / 61.2 / 60.4 / 60.9 / 58.3 / 60.8 / 61.9 / 64.9 us per frame -- short segments balance the workgroups (the ones over the frame's outside corners
                // finish early) and that outweighs the ~50 source rows each segment fetches again at its top until ~270 rows
                int segs = std::max(1, (dh + 136) / 272);
                (void)per;
                {
                    static const int segsEnv = [] { const char* v = getenv("MI355CV_WARP32_SEGS"); return v ? atoi(v) : 0; }();
                    if (segsEnv > 0) segs = segsEnv;
                }
                a.segRows = divUp(divUp(dh, segs), WS_WAVES) * WS_WAVES;
                segs = divUp(dh, a.segRows);
                int* terms = (int*)stg.scratch((size_t)(2 * dw + 2 * dh) * sizeof(int));
                uint32_t* work = (uint32_t*)stg.scratch(16);
                if (!terms || !work) return mi355::declined(__func__, __LINE__, "scratch for the coordinate terms");
                const size_t lds = (size_t)WS_NR * a.pitch * sizeof(float);
                const size_t nflags = (size_t)nframes * dh * strips;                       // one byte per wave and row: "this row piece has lanes left to the generic sampler"
                uchar* flags = (uchar*)stg.scratch(nflags);
                if (!flags) return mi355::declined(__func__, __LINE__, "scratch for the row-piece flags");
                static bool attr[64] = {}; const int dv = activeDevice() & 63;
                if (!attr[dv]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_warp32_strip), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[dv] = true; }
                hipLaunchKernelGGL(k_warp32_terms, dim3(divUp(std::max(dw, dh), 256)), dim3(256), 0, stream(), w, terms, work);
                hipLaunchKernelGGL(k_warp32_strip, dim3(strips, segs, nframes), dim3(64 * WS_WAVES), lds, stream(), ds, (uint32_t)dss, dd, (uint32_t)dds, s, w, a, terms, flags);
                hipLaunchKernelGGL(k_warp32_strip_rest, dim3((unsigned)((nflags + 255) / 256)), dim3(256), 0, stream(), ds, (uint32_t)dss, dd, (uint32_t)dds, s, w, terms, g_tabDev, flags, strips, nframes);
                noteKernel("k_warp32_strip grid=%dx%dx%d x%d strips of %d columns, %d rows per segment, ring %d x %d floats (lds %zu)", strips, segs, nframes, 64 * WS_WAVES, WS_COLS, a.segRows, WS_NR, a.pitch, lds);
                return stg.finish(entry);
            }
        }
        static const int warp32Env = [] { const char* v = getenv("MI355CV_WARP32"); return v ? atoi(v) : -1; }();
        const bool spreadRows = std::fabs(M[3]) * 64 >= 12.0, fitsLds = (std::fabs(M[0]) * 64 + std::fabs(M[1]) * 32 + 6) * (std::fabs(M[3]) * 64 + std::fabs(M[4]) * 32 + 2) <= W32_CAP;
        const bool warp32On = warp32Env < 0 ? (spreadRows && fitsLds) : warp32Env != 0;
        if (warp32On && kind == 0 && depth == D32F && cn == 1 && ((((uintptr_t)ds) | dss | w.sframe) & 15) == 0) {       // (16-byte box loads)
            dim3 g3(divUp(dw, W32_TW), divUp(dh, W32_TH), nframes);
            w.gx = g3.x; w.gy = g3.y;
            constexpr int tpw32 = 6;                                   // tiles a workgroup walks (3 / 6 / 12 / 20 within 3 %, profiles/r05_warp32_quad_ab.txt)
            // coordinate terms of every destination column and row (double arithmetic, once per call) + the list of tiles the LDS kernel leaves to k_warp32_rest
            const size_t ntiles = (size_t)g3.x * g3.y * nframes;
            int* terms = (int*)stg.scratch((size_t)(2 * dw + 2 * dh) * sizeof(int));
            uint32_t* work = (uint32_t*)stg.scratch((ntiles + 1) * sizeof(uint32_t));
            if (!terms || !work) return mi355::declined(__func__, __LINE__, "scratch for the coordinate terms / the tile list");
            hipLaunchKernelGGL(k_warp32_terms, dim3(divUp(std::max(dw, dh), 256)), dim3(256), 0, stream(), w, terms, work);
            hipLaunchKernelGGL(k_warp32_tile, dim3(divUp((int)g3.x, tpw32), g3.y, nframes), dim3(256), 0, stream(), ds, (uint32_t)dss, dd, (uint32_t)dds, s, w, tpw32, terms, work);
            hipLaunchKernelGGL(k_warp32_rest, dim3((unsigned)std::min<size_t>(ntiles, 1u << 22)), dim3(256), 0, stream(), ds, (uint32_t)dss, dd, (uint32_t)dds, s, w, g_tabDev, terms, work);
            noteKernel("k_warp32_tile (+ k_warp32_rest for what it leaves) grid=%ux%ux%u x256 tile %dx%d lds=%d floats band=%d", g3.x, g3.y, g3.z, W32_TW, W32_TH, W32_CAP, w.band);
            return stg.finish(entry);
        }
        dim3 g2(divUp(dw, 64), divUp(dh, 4 * (depth == D32F ? warpRows<float>() : warpRows<uchar>())), nframes);
        w.gx = g2.x; w.gy = g2.y;
#define WL(T_, CN_, K_) hipLaunchKernelGGL((k_warp_lin<T_, CN_, K_>), g2, dim3(256), 0, stream(), ds, (uint32_t)dss, dd, (uint32_t)dds, s, w, g_tabDev)
#define WLC(T_, K_) do { if (cn == 1) WL(T_, 1, K_); else if (cn == 3) WL(T_, 3, K_); else WL(T_, 4, K_); } while (0)
        if (kind == 0) { if (depth == D32F) WLC(float, 0); else WLC(uchar, 0); }
        else           { if (depth == D32F) WLC(float, 1); else WLC(uchar, 1); }
#undef WLC
#undef WL
        return stg.finish(entry);
    }
    dim3 grid(divUp(dw, 64), divUp(dh, 4), nframes);
    hipLaunchKernelGGL(k_warp, grid, dim3(256), 0, stream(), ds, dss, dd, dds, s, w, g_tabDev, dmx, mxs, dmy, mys);
    return stg.finish(entry);
}

} // namespace

static int runResize(const char* entry, int src_type, const uchar* src_data, size_t src_step, size_t src_frame, int src_width, int src_height,
                     uchar* dst_data, size_t dst_step, size_t dst_frame, int dst_width, int dst_height, int nframes, double inv_scale_x, double inv_scale_y,
                     int interpolation)
{
    if (disabled() || nframes < 1) return mi355::declined(__func__, __LINE__, "disabled() || nframes < 1");
    const int depth = MI355CV_MAT_DEPTH(src_type), cn = MI355CV_MAT_CN(src_type);
    if (!depthOk(depth) || cn < 1 || cn > 512 || src_width <= 0 || src_height <= 0 || dst_width <= 0 || dst_height <= 0) return mi355::declined(__func__, __LINE__, "!depthOk(depth) || cn < 1 || cn > 512 || src_width <= 0 || src_height <= 0 || dst_width <= 0 || dst_height <= 0");
    if (inv_scale_x < 2.220446049250313e-16 || inv_scale_y < 2.220446049250313e-16) {        // resize.cpp:3834-3838
        inv_scale_x = (double)dst_width / src_width; inv_scale_y = (double)dst_height / src_height;
    }
    ResizeArgs a; memset(&a, 0, sizeof a);
    a.sw = src_width; a.sh = src_height; a.dw = dst_width; a.dh = dst_height; a.depth = depth; a.cn = cn;
    a.inv_x = inv_scale_x; a.inv_y = inv_scale_y; a.scale_x = 1. / inv_scale_x; a.scale_y = 1. / inv_scale_y;
    a.isx = (int)nearbyint(a.scale_x); a.isy = (int)nearbyint(a.scale_y);
    const bool areaFast = std::fabs(a.scale_x - a.isx) < 2.220446049250313e-16 && std::fabs(a.scale_y - a.isy) < 2.220446049250313e-16;
    if (interpolation == MI355CV_INTER_NEAREST) a.mode = 0;
    else if (interpolation == MI355CV_INTER_NEAREST_EXACT) {
        // resizeNN_bitexact (resize.cpp:1267-1289): source pixel = (ifx * x + ifx0) >> 16 in int arithmetic, steps rounded to 16.16, pixel centres aligned
        if (src_width >= 32768 || src_height >= 32768) return mi355::declined(__func__, __LINE__, "src_width >= 32768 || src_height >= 32768");          // (size << 16) must stay an int, as in the reference
        a.mode = 0; a.nnExact = 1;
        a.ifx = ((src_width << 16) + dst_width / 2) / dst_width; a.ifx0 = a.ifx / 2 - src_width % 2;
        a.ify = ((src_height << 16) + dst_height / 2) / dst_height; a.ify0 = a.ify / 2 - src_height % 2;
    } else {
        if (interpolation == MI355CV_INTER_LINEAR && areaFast && a.isx == 2 && a.isy == 2) interpolation = MI355CV_INTER_AREA;   // :4011
        if (interpolation == MI355CV_INTER_AREA && a.scale_x >= 1 && a.scale_y >= 1) {
            a.mode = areaFast ? 3 : 4;                                                       // 4: true area (resizeArea_)
        } else if (interpolation == MI355CV_INTER_LINEAR) a.mode = 1;
        else if (interpolation == MI355CV_INTER_AREA) a.mode = 2;
        else if (interpolation == MI355CV_INTER_LINEAR_EXACT && (depth == D8U || depth == D16U || depth == D16S)) {
            // resize.cpp:3976-3990: exactly-half sizes are the (bit-exact) 2x2 area mean, except for 2 channels
            if (areaFast && a.isx == 2 && a.isy == 2 && cn != 2) a.mode = 3; else a.mode = 7;
        }
        else if (interpolation == 2 /*INTER_CUBIC*/) a.mode = 5;
        else if (interpolation == 4 /*INTER_LANCZOS4*/) a.mode = 6;
        else return mi355::declined(__func__, __LINE__, nullptr);                                                // LINEAR_EXACT on other depths
    }
    if (a.mode == 4 && cn > 4) return mi355::declined(__func__, __LINE__, "true INTER_AREA of more than 4 channels (the reference asserts on it, resize.cpp:4045)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)dst_width * dst_height, minPixels(a.mode >= 3 ? HOST_HEAVY : HOST_CHEAP))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)dst_width * dst_height, minPixels(a.mode >= 3 ? HOST_HEAVY : HOST_CHEAP))");
    const int e = eszOf(depth);
    if (nframes > 1) {
        // batches are an HBM-resident construct; nearest / bilinear / area-fast run as ONE launch (grid z = frame), the table-driven modes frame by frame
        if (!isDevicePtr(src_data) || !isDevicePtr(dst_data)) return setError(MI355CV_NOT_IMPLEMENTED, "%s: batch entry needs device-resident frames", entry);
        if (a.mode >= 4) {
            for (int f = 0; f < nframes; f++) {
                const int rc = runResize(entry, src_type, src_data + (size_t)f * src_frame, src_step, 0, src_width, src_height, dst_data + (size_t)f * dst_frame, dst_step, 0,
                                         dst_width, dst_height, 1, inv_scale_x, inv_scale_y, interpolation);
                if (rc != MI355CV_OK) return rc;
            }
            return MI355CV_OK;
        }
        a.sframe = src_frame; a.dframe = dst_frame;
    }
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)src_width * cn * e, src_height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)dst_width * cn * e, dst_height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    if (a.mode == 5 || a.mode == 6) {
        dim3 gt(divUp(dst_width * cn, 64), divUp(dst_height, 16)), g1(divUp(dst_width * cn, 64), divUp(dst_height, 4));
        int rows = 0, unused = 0;
        DevRef keepX, keepY;                                        // the tables stay alive until the launches below are enqueued
        // CV_8U: tiles of 256 x 16 elements with the source bytes staged in LDS (resize_tab8.h) where a tile's rows fit.  Measured against the 64 x 16 kernel
        // (profiles/r03_resize_tab8_ab.txt): 1080p -> 4K cubic 8UC3 46.7 against 59.9 us, 8UC1 20.1 / 21.5, Lanczos 68.5 / 66.7, a 1.5 x cubic downscale
        // 53.2 / 52.1 -- so it serves cubic upscales.  MI355CV_RESIZE_TAB8=1: wherever it is eligible, =0: never.
        static const int tab8Env = getenv("MI355CV_RESIZE_TAB8") ? atoi(getenv("MI355CV_RESIZE_TAB8")) : -1;
        rt8::Geom geom8 = {src_width, src_height, dst_width, dst_height, cn, 0};
        size_t lds8 = 0;
        const dim3 g8(divUp(dst_width * cn, rt8::TW), divUp(dst_height, rt8::TH));
        auto tab8 = [&](int nt) {
            if (tab8Env == 0 || (tab8Env < 0 && !(nt == 4 && a.scale_x <= 1.0 && a.scale_y <= 1.0))) return false;
            geom8.sp = rt8::stagePitch(cn, a.scale_x, nt);
            lds8 = (size_t)rows * ((size_t)rt8::TW * 4 + (size_t)geom8.sp);
            return lds8 <= 48 * 1024;                                 // within what a workgroup gets without asking for more
        };
        if (a.mode == 5) {
            const CubicTap *dxt, *dyt;
            if (!cachedTab<CubicTap>(5, dst_width, a.scale_x, 4, buildCubicTab, &dxt, &unused, &keepX) || !cachedTab<CubicTap>(5, dst_height, a.scale_y, 4, buildCubicTab, &dyt, &rows, &keepY))
                return mi355::declined(__func__, __LINE__, "!cachedTab<CubicTap>(5, dst_width, a.scale_x, 4, buildCubicTab, &dxt, &unused, &keepX) || !cachedTab<CubicTap>(5, dst_height, a.scale_y, 4, buildCubicTab, &dyt, &rows, &keepY)");
            const TapT<4>* tx = reinterpret_cast<const TapT<4>*>(dxt); const TapT<4>* ty = reinterpret_cast<const TapT<4>*>(dyt);
#define RZ_TILED(T_, NT_) hipLaunchKernelGGL((k_resize_tiled<T_, NT_>), gt, dim3(256), (size_t)rows * 256, stream(), ds, dss, dd, dds, src_width, src_height, dst_width, dst_height, cn, tx, ty)
#define RZ_BY_DEPTH(M_) do { if (depth == D8U) M_(uchar); else if (depth == D16U) M_(unsigned short); else if (depth == D16S) M_(short); else M_(float); } while (0)
#define RZ_T4(T_) RZ_TILED(T_, 4)
#define RZ_C(T_) hipLaunchKernelGGL(k_resize_cubic<T_>, g1, dim3(256), 0, stream(), ds, dss, dd, dds, src_width, src_height, dst_width, dst_height, cn, dxt, dyt)
            if (rows && depth == D8U && tab8(4)) {
                hipLaunchKernelGGL(k_resize_tab8<4>, g8, dim3(256), lds8, stream(), ds, dss, dd, dds, geom8, tx, ty, rows);
                noteKernel("k_resize_tab8<4> grid=%ux%u x256 lds=%zu", g8.x, g8.y, lds8);
            } else if (rows) RZ_BY_DEPTH(RZ_T4); else RZ_BY_DEPTH(RZ_C);
        } else {
            const LanczosTap *dxt, *dyt;
            if (!cachedTab<LanczosTap>(6, dst_width, a.scale_x, 8, buildLanczosTab, &dxt, &unused, &keepX) || !cachedTab<LanczosTap>(6, dst_height, a.scale_y, 8, buildLanczosTab, &dyt, &rows, &keepY))
                return mi355::declined(__func__, __LINE__, "!cachedTab<LanczosTap>(6, dst_width, a.scale_x, 8, buildLanczosTab, &dxt, &unused, &keepX) || !cachedTab<LanczosTap>(6, dst_height, a.scale_y, 8, buildLanczosTab, &dyt, &rows, &keepY)");
            const TapT<8>* tx = reinterpret_cast<const TapT<8>*>(dxt); const TapT<8>* ty = reinterpret_cast<const TapT<8>*>(dyt);
#define RZ_T8(T_) RZ_TILED(T_, 8)
#define RZ_L(T_) hipLaunchKernelGGL(k_resize_lanczos<T_>, g1, dim3(256), 0, stream(), ds, dss, dd, dds, src_width, src_height, dst_width, dst_height, cn, dxt, dyt)
            if (rows && depth == D8U && tab8(8)) {
                hipLaunchKernelGGL(k_resize_tab8<8>, g8, dim3(256), lds8, stream(), ds, dss, dd, dds, geom8, tx, ty, rows);
                noteKernel("k_resize_tab8<8> grid=%ux%u x256 lds=%zu", g8.x, g8.y, lds8);
            } else if (rows) RZ_BY_DEPTH(RZ_T8); else RZ_BY_DEPTH(RZ_L);
#undef RZ_T8
#undef RZ_L
#undef RZ_T4
#undef RZ_C
#undef RZ_BY_DEPTH
#undef RZ_TILED
        }
        return stg.finish(entry);
    }
    if (a.mode == 7) {
        const int shift = depth == D8U ? 8 : 16;
        const ExactTap *dx, *dy;
        DevRef keepX, keepY;                                        // the tables stay alive until the launch below is enqueued
        if (!cachedExactTab(inv_scale_x, src_width, dst_width, shift, &dx, &keepX) || !cachedExactTab(inv_scale_y, src_height, dst_height, shift, &dy, &keepY))
            return mi355::declined(__func__, __LINE__, "!cachedExactTab(inv_scale_x, src_width, dst_width, shift, &dx, &keepX) || !cachedExactTab(inv_scale_y, src_height, dst_height, shift, &dy, &keepY)");
        dim3 g7(divUp(dst_width * cn, 64), divUp(dst_height, 4));
        if (depth == D8U) hipLaunchKernelGGL((k_resize_exact<uchar, 8>), g7, dim3(256), 0, stream(), ds, dss, dd, dds, dst_width, dst_height, cn, dx, dy);
        else if (depth == D16U) hipLaunchKernelGGL((k_resize_exact<unsigned short, 16>), g7, dim3(256), 0, stream(), ds, dss, dd, dds, dst_width, dst_height, cn, dx, dy);
        else hipLaunchKernelGGL((k_resize_exact<short, 16>), g7, dim3(256), 0, stream(), ds, dss, dd, dds, dst_width, dst_height, cn, dx, dy);
        return stg.finish(entry);
    }
    if (a.mode == 3 && depth == D8U && a.isx == 2 && a.isy == 2 && (cn == 1 || cn == 3 || cn == 4) && src_width == 2 * dst_width && src_height == 2 * dst_height &&
        ((((uintptr_t)ds) | dss | a.sframe) & 7) == 0 && ((((uintptr_t)dd) | dds | a.dframe) & 3) == 0) {
        dim3 g3(divUp(divUp(dst_width, 4), 64), divUp(dst_height, 4), nframes);
        if (cn == 1) hipLaunchKernelGGL(k_area2x2_u8<1>, g3, dim3(256), 0, stream(), ds, dss, a.sframe, dd, dds, a.dframe, dst_width, dst_height);
        else if (cn == 3) hipLaunchKernelGGL(k_area2x2_u8<3>, g3, dim3(256), 0, stream(), ds, dss, a.sframe, dd, dds, a.dframe, dst_width, dst_height);
        else hipLaunchKernelGGL(k_area2x2_u8<4>, g3, dim3(256), 0, stream(), ds, dss, a.sframe, dd, dds, a.dframe, dst_width, dst_height);
        return stg.finish(entry);
    }
    if (a.mode == 3 && depth == D32F && cn == 1 && a.isx == 2 && a.isy == 2 && src_width == 2 * dst_width && src_height == 2 * dst_height &&
        ((((uintptr_t)ds) | dss | a.sframe) & 15) == 0 && ((((uintptr_t)dd) | dds | a.dframe) & 15) == 0) {
        dim3 g3(divUp(divUp(dst_width, 4), 64), divUp(dst_height, 4), nframes);
        hipLaunchKernelGGL(k_area2x2_f32, g3, dim3(256), 0, stream(), ds, dss, a.sframe, dd, dds, a.dframe, dst_width, dst_height);
        noteKernel("k_area2x2_f32 grid=%ux%ux%u x256", g3.x, g3.y, g3.z);
        return stg.finish(entry);
    }
    if (a.mode == 4) {
        AreaDev ax, ay;
        if (!cachedAreaTab(src_width, dst_width, a.scale_x, &ax) || !cachedAreaTab(src_height, dst_height, a.scale_y, &ay)) return mi355::declined(__func__, __LINE__, "!cachedAreaTab(src_width, dst_width, a.scale_x, &ax) || !cachedAreaTab(src_height, dst_height, a.scale_y, &ay)");
        const AreaTap* dxt = ax.tab; const int* dxo = ax.ofs; const AreaTap* dyt = ay.tab; const int* dyo = ay.ofs;
        dim3 g4(divUp(dst_width * cn, 64), divUp(dst_height, 4));
#define RA(T_) hipLaunchKernelGGL(k_resize_area<T_>, g4, dim3(256), 0, stream(), ds, dss, dd, dds, dst_width, dst_height, cn, depth, dxt, dxo, dyt, dyo)
        switch (depth) { case D8U: RA(uchar); break; case D16U: RA(unsigned short); break; case D16S: RA(short); break; default: RA(float); }
#undef RA
        return stg.finish(entry);
    }
    if ((a.mode == 1 || a.mode == 2) && depth == D8U && (cn == 1 || cn == 3)) {
        // the LDS pipeline of the lean warp kernel (four pixels per lane, dword stores, taps out of a staged tile); the plan bounds every tile's source box from the
        // scale factors, so a planned call has no tile the kernel cannot take.  MI355CV_RESIZE8_LEAN=0 keeps the column-owning kernels (A/B runs)
        static const int rzOn = [] { const char* v = getenv("MI355CV_RESIZE8_LEAN"); return v ? atoi(v) : 1; }();
        warp8::Args a8; size_t ldsR = 0;
        if (rzOn && warp8::planResize(a8, cn, src_width, src_height, dst_width, dst_height, dss, dds, ds, dd, a.scale_x, a.inv_x, a.scale_y, a.inv_y, a.mode == 2, &ldsR)) {
            int* tt = (int*)stg.scratch((size_t)(2 * dst_width + 4 * dst_height) * sizeof(int));
            if (tt) {
                a8.sframe = a.sframe; a8.dframe = a.dframe;
                hipLaunchKernelGGL(k_resize8_terms, dim3(divUp(std::max(dst_width, dst_height), 256)), dim3(256), 0, stream(), a8, tt, tt + 2 * dst_width);
                a8.colT = tt; a8.rowT = tt + 2 * dst_width;
                const int tpw = 6;
                dim3 gl(divUp(a8.gx, tpw), a8.gy, nframes);
#define RZN(CN_, LW_, NR_) if (cn == CN_ && a8.leanLW == LW_ && a8.leanNR == NR_) hipLaunchKernelGGL((k_resize8_lean<CN_, LW_, NR_>), gl, dim3(256), ldsR, stream(), ds, dd, a8, tpw)
#define RZS(CN_, LW_) RZN(CN_, LW_, 6); RZN(CN_, LW_, 10); RZN(CN_, LW_, 14); RZN(CN_, LW_, 20)
                RZS(1, 16); RZS(1, 32); RZS(1, 64); RZS(3, 32); RZS(3, 64); RZS(3, 128); RZS(3, 256);
#undef RZS
#undef RZN
                noteKernel("k_resize8_lean<%d,%d,%d> grid=%ux%ux%u x256 tpw=%d lds=%zu box<=%dx%d", cn, a8.leanLW, a8.leanNR, gl.x, gl.y, gl.z, tpw, ldsR, (a8.ldsPitch - 8) / cn, a8.ldsRows);
                return stg.finish(entry);
            }
        }
    }
    if ((a.mode == 1 || a.mode == 2) && cn == 1 && (depth == D32F || depth == D8U) && src_width >= 2 && (dss % e) == 0 && ((uintptr_t)ds % e) == 0) {
        dim3 g2(divUp(dst_width, 64), divUp(dst_height, 4 * RROWS), nframes);
        if (depth == D32F) hipLaunchKernelGGL(k_resize_lin1<float>, g2, dim3(256), 0, stream(), ds, dss, dd, dds, a);
        else hipLaunchKernelGGL(k_resize_lin1<uchar>, g2, dim3(256), 0, stream(), ds, dss, dd, dds, a);
        return stg.finish(entry);
    }
    if ((a.mode == 1 || a.mode == 2) && (cn == 3 || cn == 4) && (depth == D32F || depth == D8U) && (dss % e) == 0 && ((uintptr_t)ds % e) == 0) {
        dim3 g2(divUp(dst_width, 64), divUp(dst_height, 4 * RROWS), nframes);
#define RLC(T_, CN_) hipLaunchKernelGGL((k_resize_linC<T_, CN_>), g2, dim3(256), 0, stream(), ds, dss, dd, dds, a)
        if (depth == D32F) { if (cn == 3) RLC(float, 3); else RLC(float, 4); }
        else { if (cn == 3) RLC(uchar, 3); else RLC(uchar, 4); }
#undef RLC
        return stg.finish(entry);
    }
    dim3 grid(divUp(dst_width, 64), divUp(dst_height, 4), nframes);
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, stream(), ds, dss, dd, dds, a);
    return stg.finish(entry);
}

extern "C" {

MI355CV_API int mi355cv_resize(int src_type, const uchar* src_data, size_t src_step, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, int dst_width, int dst_height, double inv_scale_x, double inv_scale_y, int interpolation)
{
    mi355::EntryGuard entry_(__func__);
    return runResize("resize", src_type, src_data, src_step, 0, src_width, src_height, dst_data, dst_step, 0, dst_width, dst_height, 1, inv_scale_x, inv_scale_y, interpolation);
}

// batches of device-resident frames (frames are the unit that shards): every frame through the same transform, one launch where
// the kernel takes a frame index (nearest / bilinear / area-fast resize, every warp), frame strides in bytes
MI355CV_API int mi355cv_resizeBatch(int src_type, const uchar* src_data, size_t src_step, size_t src_frame_stride, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int dst_width, int dst_height, int nframes, double inv_scale_x, double inv_scale_y, int interpolation)
{
    mi355::EntryGuard entry_(__func__);
    if (src_width > 0 && src_height > 0 && dst_width > 0 && dst_height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {        // frames in host memory
        const size_t pix = (size_t)MI355CV_MAT_CN(src_type) * depthBytes(MI355CV_MAT_DEPTH(src_type));
        const HostBatch hb = {src_data, src_step, src_frame_stride, pix * src_width, src_height, dst_data, dst_step, dst_frame_stride, pix * dst_width, dst_height, nframes};
        return runHostBatch("resizeBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_resizeBatch(src_type, s, ss, sf, src_width, src_height, d, ds, df, dst_width, dst_height, nf, inv_scale_x, inv_scale_y, interpolation); });
    }
    return runResize("resizeBatch", src_type, src_data, src_step, src_frame_stride, src_width, src_height, dst_data, dst_step, dst_frame_stride, dst_width, dst_height,
                     nframes, inv_scale_x, inv_scale_y, interpolation);
}

MI355CV_API int mi355cv_warpAffineBatch(int src_type, const uchar* src_data, size_t src_step, size_t src_frame_stride, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int dst_width, int dst_height, int nframes, const double M[6], int interpolation, int borderType,
        const double borderValue[4])
{
    mi355::EntryGuard entry_(__func__);
    if (!M) return mi355::declined(__func__, __LINE__, "!M");
    if (src_width > 0 && src_height > 0 && dst_width > 0 && dst_height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {        // frames in host memory
        // BORDER_TRANSPARENT keeps the caller's pixels where the map leaves the source: the pipeline's device buffers do not hold them, so
        // such a batch goes frame by frame through the single-image path, which stages dst in as well
        if (borderType == B_TRANSPARENT) {
            for (int f = 0; f < nframes; f++) {
                const int rc = runWarp("warpAffineBatch", src_type, src_data + (size_t)f * src_frame_stride, src_step, src_width, src_height,
                                       dst_data + (size_t)f * dst_frame_stride, dst_step, dst_width, dst_height, M, 0, interpolation, borderType, borderValue,
                                       nullptr, 0, nullptr, 0);
                if (rc != MI355CV_OK) return rc;
            }
            return MI355CV_OK;
        }
        const size_t pix = (size_t)MI355CV_MAT_CN(src_type) * depthBytes(MI355CV_MAT_DEPTH(src_type));
        const HostBatch hb = {src_data, src_step, src_frame_stride, pix * src_width, src_height, dst_data, dst_step, dst_frame_stride, pix * dst_width, dst_height, nframes};
        return runHostBatch("warpAffineBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_warpAffineBatch(src_type, s, ss, sf, src_width, src_height, d, ds, df, dst_width, dst_height, nf, M, interpolation, borderType, borderValue); });
    }
    return runWarp("warpAffineBatch", src_type, src_data, src_step, src_width, src_height, dst_data, dst_step, dst_width, dst_height,
                   M, 0, interpolation, borderType, borderValue, nullptr, 0, nullptr, 0, nframes, src_frame_stride, dst_frame_stride);
}

MI355CV_API int mi355cv_warpPerspectiveBatch(int src_type, const uchar* src_data, size_t src_step, size_t src_frame_stride, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int dst_width, int dst_height, int nframes, const double M[9], int interpolation, int borderType,
        const double borderValue[4])
{
    mi355::EntryGuard entry_(__func__);
    if (!M) return mi355::declined(__func__, __LINE__, "!M");
    if (src_width > 0 && src_height > 0 && dst_width > 0 && dst_height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {        // frames in host memory
        // BORDER_TRANSPARENT keeps the caller's pixels where the map leaves the source: the pipeline's device buffers do not hold them, so
        // such a batch goes frame by frame through the single-image path, which stages dst in as well
        if (borderType == B_TRANSPARENT) {
            for (int f = 0; f < nframes; f++) {
                const int rc = runWarp("warpPerspectiveBatch", src_type, src_data + (size_t)f * src_frame_stride, src_step, src_width, src_height,
                                       dst_data + (size_t)f * dst_frame_stride, dst_step, dst_width, dst_height, M, 1, interpolation, borderType, borderValue,
                                       nullptr, 0, nullptr, 0);
                if (rc != MI355CV_OK) return rc;
            }
            return MI355CV_OK;
        }
        const size_t pix = (size_t)MI355CV_MAT_CN(src_type) * depthBytes(MI355CV_MAT_DEPTH(src_type));
        const HostBatch hb = {src_data, src_step, src_frame_stride, pix * src_width, src_height, dst_data, dst_step, dst_frame_stride, pix * dst_width, dst_height, nframes};
        return runHostBatch("warpPerspectiveBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_warpPerspectiveBatch(src_type, s, ss, sf, src_width, src_height, d, ds, df, dst_width, dst_height, nf, M, interpolation, borderType, borderValue); });
    }
    return runWarp("warpPerspectiveBatch", src_type, src_data, src_step, src_width, src_height, dst_data, dst_step, dst_width, dst_height,
                   M, 1, interpolation, borderType, borderValue, nullptr, 0, nullptr, 0, nframes, src_frame_stride, dst_frame_stride);
}

MI355CV_API int mi355cv_warpAffine(int src_type, const uchar* src_data, size_t src_step, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, int dst_width, int dst_height, const double M[6], int interpolation, int borderType,
        const double borderValue[4])
{
    mi355::EntryGuard entry_(__func__);
    if (!M) return mi355::declined(__func__, __LINE__, "!M");
    return runWarp("warpAffine", src_type, src_data, src_step, src_width, src_height, dst_data, dst_step, dst_width, dst_height,
                   M, 0, interpolation, borderType, borderValue, nullptr, 0, nullptr, 0);
}

MI355CV_API int mi355cv_warpPerspective(int src_type, const uchar* src_data, size_t src_step, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, int dst_width, int dst_height, const double M[9], int interpolation, int borderType,
        const double borderValue[4])
{
    mi355::EntryGuard entry_(__func__);
    if (!M) return mi355::declined(__func__, __LINE__, "!M");
    return runWarp("warpPerspective", src_type, src_data, src_step, src_width, src_height, dst_data, dst_step, dst_width, dst_height,
                   M, 1, interpolation, borderType, borderValue, nullptr, 0, nullptr, 0);
}

MI355CV_API int mi355cv_remap32f(int src_type, const uchar* src_data, size_t src_step, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, int dst_width, int dst_height, float* mapx, size_t mapx_step, float* mapy, size_t mapy_step,
        int interpolation, int border_type, const double border_value[4])
{
    mi355::EntryGuard entry_(__func__);
    if (!mapx || !mapy) return mi355::declined(__func__, __LINE__, "!mapx || !mapy");
    return runWarp("remap32f", src_type, src_data, src_step, src_width, src_height, dst_data, dst_step, dst_width, dst_height,
                   nullptr, 2, interpolation, border_type, border_value, mapx, mapx_step, mapy, mapy_step);
}

// cv::remap for every map representation it accepts (imgwarp.cpp:1718-1921; no HAL hook beyond remap32f): a pair of CV_32FC1 planes, one
// CV_32FC2 map, or the fixed-point form convertMaps produces -- CV_16SC2 integer coordinates with CV_16UC1 / CV_16SC1 fractions (bilinear or
// nearest) or CV_16SC2 alone (nearest).  NEAREST / LINEAR (AREA = LINEAR), every border mode incl. TRANSPARENT.  map types are cv type codes.
MI355CV_API int mi355cv_remap(int src_type, const uchar* src_data, size_t src_step, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, int dst_width, int dst_height, const void* map1, size_t map1_step, int map1_type,
        const void* map2, size_t map2_step, int map2_type, int interpolation, int border_type, const double border_value[4])
{
    mi355::EntryGuard entry_(__func__);
    if (!map1) return mi355::declined(__func__, __LINE__, "!map1");
    const int t32fc1 = MI355CV_MAKETYPE(MI355CV_32F, 1), t32fc2 = MI355CV_MAKETYPE(MI355CV_32F, 2), t16sc2 = MI355CV_MAKETYPE(MI355CV_16S, 2);
    const int t16uc1 = MI355CV_MAKETYPE(MI355CV_16U, 1), t16sc1 = MI355CV_MAKETYPE(MI355CV_16S, 1);
    int interp = interpolation & 7;
    if (interp == MI355CV_INTER_AREA) interp = MI355CV_INTER_LINEAR;
    interp |= interpolation & 32;                                                              // WARP_RELATIVE_MAP travels to runWarp
    if (map2 && map2_type == t16sc2 && (map1_type == t16uc1 || map1_type == t16sc1)) {        // either order is accepted (:1905-1909)
        std::swap(map1, map2); std::swap(map1_step, map2_step); std::swap(map1_type, map2_type);
    }
    int kind;
    if (map1_type == t32fc1 && map2 && map2_type == t32fc1) kind = 2;
    else if (map1_type == t32fc2 && !map2) kind = 3;
    else if (map1_type == t16sc2 && map2 && (map2_type == t16uc1 || map2_type == t16sc1)) kind = 4;
    else if (map1_type == t16sc2 && !map2 && (interp & 7) == MI355CV_INTER_NEAREST) kind = 5;
    else return mi355::declined(__func__, __LINE__, nullptr);
    return runWarp("remap", src_type, src_data, src_step, src_width, src_height, dst_data, dst_step, dst_width, dst_height,
                   nullptr, kind, interp, border_type, border_value, (const float*)map1, map1_step, (const float*)map2, map2_step);
}

// cv::convertMaps (imgwarp.cpp:1925-2260) between the float and the fixed-point map representations, device-resident or host maps:
//   (CV_32FC1, CV_32FC1) or (CV_32FC2) -> CV_16SC2 + CV_16UC1 (nninterpolate: CV_16SC2 alone);  CV_16SC2 (+ CV_16UC1) -> CV_32FC1 pair or CV_32FC2
MI355CV_API int mi355cv_convertMaps(const void* map1, size_t map1_step, int map1_type, const void* map2, size_t map2_step, int map2_type,
        void* dstmap1, size_t dstmap1_step, int dstmap1_type, void* dstmap2, size_t dstmap2_step, int width, int height, int nninterpolate)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || !map1 || !dstmap1 || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || !map1 || !dstmap1 || width <= 0 || height <= 0");
    const int t32fc1 = MI355CV_MAKETYPE(MI355CV_32F, 1), t32fc2 = MI355CV_MAKETYPE(MI355CV_32F, 2), t16sc2 = MI355CV_MAKETYPE(MI355CV_16S, 2);
    const int t16uc1 = MI355CV_MAKETYPE(MI355CV_16U, 1), t16sc1 = MI355CV_MAKETYPE(MI355CV_16S, 1);
    const bool toFixed = dstmap1_type == t16sc2 && ((map1_type == t32fc1 && map2 && map2_type == t32fc1) || (map1_type == t32fc2 && !map2));
    const bool toFloat = map1_type == t16sc2 && (!map2 || map2_type == t16uc1 || map2_type == t16sc1) && (dstmap1_type == t32fc1 || dstmap1_type == t32fc2);
    if (!toFixed && !toFloat) return mi355::declined(__func__, __LINE__, "!toFixed && !toFloat");
    if (toFixed && !nninterpolate && !dstmap2) return mi355::declined(__func__, __LINE__, "toFixed && !nninterpolate && !dstmap2");
    if (toFloat && dstmap1_type == t32fc1 && !dstmap2) return mi355::declined(__func__, __LINE__, "toFloat && dstmap1_type == t32fc1 && !dstmap2");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    size_t s1 = 0, s2 = 0, o1 = 0, o2 = 0;
    const size_t e1 = map1_type == t32fc1 ? 4 : map1_type == t32fc2 ? 8 : 4;
    const uchar* a = stg.in((const uchar*)map1, map1_step, (size_t)width * e1, height, &s1);
    const uchar* b = map2 ? stg.in((const uchar*)map2, map2_step, (size_t)width * (toFixed ? 4 : 2), height, &s2) : nullptr;
    const size_t f1 = dstmap1_type == t32fc1 ? 4 : dstmap1_type == t32fc2 ? 8 : 4;
    uchar* c = stg.out((uchar*)dstmap1, dstmap1_step, (size_t)width * f1, height, &o1);
    const bool needD2 = toFixed ? !nninterpolate : dstmap1_type == t32fc1;
    uchar* d = needD2 ? stg.out((uchar*)dstmap2, dstmap2_step, (size_t)width * (toFixed ? 2 : 4), height, &o2) : nullptr;
    if (!a || (map2 && !b) || !c || (needD2 && !d)) return mi355::declined(__func__, __LINE__, "!a || (map2 && !b) || !c || (needD2 && !d)");
    dim3 grid(divUp(width, 64), divUp(height, 4));
    if (toFixed) hipLaunchKernelGGL(k_convert_maps_to_fixed, grid, dim3(256), 0, stream(), a, s1, b, s2, map1_type == t32fc2 ? 1 : 0, c, o1, d, o2, width, height, nninterpolate ? 1 : 0);
    else hipLaunchKernelGGL(k_convert_maps_to_float, grid, dim3(256), 0, stream(), a, s1, b, s2, c, o1, d, o2, dstmap1_type == t32fc2 ? 1 : 0, width, height);
    return stg.finish("convertMaps");
}

// cv::warpPolar, forward direction (imgwarp.cpp:3731-3793: Cartesian image -> polar / semilog-polar image), as ONE kernel: the maps the
// reference materialises (two CV_32F images) are evaluated per output pixel from dsize.width radii and dsize.height (cos, sin) pairs, which the
// host computes exactly as the reference does (libm in double), followed by remap's sampling.  WARP_INVERSE_MAP is left to the reference
// (its map needs cartToPolar / log in the CPU's own approximations; the remap it ends in is served by the remap32f hook).
MI355CV_API int mi355cv_warpPolar(int src_type, const uchar* src_data, size_t src_step, int src_width, int src_height,
        uchar* dst_data, size_t dst_step, int dst_width, int dst_height, float center_x, float center_y, double maxRadius, int flags)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || dst_width <= 0 || dst_height <= 0 || src_width <= 0 || src_height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || dst_width <= 0 || dst_height <= 0 || src_width <= 0 || src_height <= 0");
    const bool semiLog = (flags & 256) != 0;                                                  // WARP_POLAR_LOG
    const double bv[4] = {0, 0, 0, 0};
    if (flags & MI355CV_WARP_INVERSE_MAP) {
        // polar / semi-log polar image -> Cartesian image (imgwarp.cpp:3795-3845): the source gets one wrapped row above and below (copyMakeBorder BORDER_WRAP),
        // the map is evaluated per destination pixel in the kernel (k_warp kind 7), then cv::remap's sampling
        const int depth = MI355CV_MAT_DEPTH(src_type), cn = MI355CV_MAT_CN(src_type);
        if (!depthOk(depth) || cn < 1 || cn > 4 || src_height + 2 > 32767) return mi355::declined(__func__, __LINE__, "!depthOk(depth) || cn < 1 || cn > 4 || src_height + 2 > 32767");
        Stager outer;                                  // first: a declined call must also put the host's device back (~Stager)
        if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
        if (hostImageTooSmall(src_data, (size_t)dst_width * dst_height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)dst_width * dst_height, minPixels(HOST_HEAVY))");
        const size_t rowB = (size_t)src_width * cn * eszOf(depth), bstep = (rowB + 255) & ~(size_t)255;
        size_t dss;
        const uchar* ds = outer.in(src_data, src_step, rowB, src_height, &dss);
        uchar* bordered = (uchar*)outer.scratch(bstep * (size_t)(src_height + 2));
        if (!ds || !bordered) return mi355::declined(__func__, __LINE__, "!ds || !bordered");
        hipStream_t st = stream();
        if (hipMemcpy2DAsync(bordered + bstep, bstep, ds, dss, rowB, src_height, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(bordered, ds + (size_t)(src_height - 1) * dss, rowB, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(bordered + bstep * (size_t)(src_height + 1), ds, rowB, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return setError(MI355CV_ERROR_UNKNOWN, "warpPolar: %s", hipGetErrorString(hipGetLastError()));
        static float logTab[512]; static std::once_flag once;
        std::call_once(once, [] {                                                           // the reference's logTab_f (mathfuncs_core: log(1 + i/256) and 1 / (1 + i/256) as floats)
            for (int i = 0; i < 256; i++) { const double t = 1.0 + i / 256.0; logTab[2 * i] = (float)std::log(t); logTab[2 * i + 1] = (float)(1.0 / t); }
            logTab[510] = (float)0.69314718055994530941723212145818; logTab[511] = 0.5f;       // the last cell is measured from 2.0
        });
        const double Kangle = 6.283185307179586476925286766559 / src_height;
        const double Kmag = semiLog ? std::log(maxRadius) / src_width : maxRadius / src_width;
        const double M[5] = {(double)center_x, (double)center_y, Kmag, Kangle, semiLog ? 1.0 : 0.0};
        const int rc = runWarp("warpPolar", src_type, bordered, bstep, src_width, src_height + 2, dst_data, dst_step, dst_width, dst_height,
                               M, 7, flags & 7, (flags & 8) ? B_CONSTANT : B_TRANSPARENT, bv, logTab, 0, nullptr, 0);
        if (rc != MI355CV_OK) return rc;
        if (!asyncMode() && hipStreamSynchronize(st) != hipSuccess) return setError(MI355CV_ERROR_UNKNOWN, "warpPolar: %s", hipGetErrorString(hipGetLastError()));
        return MI355CV_OK;
    }
    std::vector<float> rhos((size_t)dst_width);
    std::vector<double> cs(2 * (size_t)dst_height);
    if (semiLog) { const double Kmag = std::log(maxRadius) / dst_width; for (int r = 0; r < dst_width; r++) rhos[(size_t)r] = (float)(std::exp(r * Kmag) - 1.0); }
    else { const double Kmag = maxRadius / dst_width; for (int r = 0; r < dst_width; r++) rhos[(size_t)r] = (float)(r * Kmag); }
    const double Kangle = 6.283185307179586476925286766559 / dst_height;                      // CV_2PI
    for (int p = 0; p < dst_height; p++) { const double k = Kangle * p; cs[2 * (size_t)p] = std::cos(k); cs[2 * (size_t)p + 1] = std::sin(k); }
    const double M[2] = {(double)center_x, (double)center_y};
    return runWarp("warpPolar", src_type, src_data, src_step, src_width, src_height, dst_data, dst_step, dst_width, dst_height,
                   M, 6, flags & 7, (flags & 8) ? B_CONSTANT : B_TRANSPARENT, bv, rhos.data(), 0, (const float*)cs.data(), 0);   // 8 = WARP_FILL_OUTLIERS
}

} // extern "C"
