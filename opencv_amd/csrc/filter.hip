// filter.hip -- rows a3/a4 of SURVEY.md §8: cv::filter2D, cv::sepFilter2D, cv::Sobel, cv::Scharr.
//
// Reference semantics restated here:
//  * filter2D (filter.simd.hpp:3103 Filter2D, :2146 FilterVec_8u, filter.dispatch.cpp:390 preprocess2DKernel):
//      kernel -> float; s = delta; for every NON-ZERO tap in raster order: s = fma(float(px), k, s);
//      dst = saturate_cast<DT>(s) (round-half-even for integer DT).  Correlation (kernel not flipped), anchor,
//      borders by borderInterpolate over the FULL image when the ROI has real neighbours, BORDER_CONSTANT = 0.
//  * sepFilter2D (filter.dispatch.cpp:305 createSeparableLinearFilter):
//      - 8U->8U with both kernels smooth+symmetrical: bit-exact integer mode, taps*256 -> int32,
//        dst = sat_u8((sum_j ky[j]*sum_i kx[i]*p + delta*2^16 + 2^15) >> 16)         (:326-352, FixedPtCastEx :2937)
//      - 8U->16S with integer (anti)symmetrical kernels (Sobel/Scharr): exact int32, dst = sat_s16(sum + delta)
//      - otherwise float: row sums s = k0*p0, s = fma(kk, pk, s) (RowFilter :2386), column in the
//        (anti)symmetric pair form s = k0*r0 + delta, s += kk*(r[+k] +- r[-k]) (SymmColumnFilter :2679-2751) or the
//        plain chain (ColumnFilter :2609), then saturate_cast<DT>.  Float results agree with the CPU to rounding
//        (the CPU picks among several SIMD association orders); the parity bar there is 1e-4 relative.
//  * Sobel / Scharr (deriv.cpp:87 getSobelKernels, :55 getScharrKernels, :414 cv::Sobel): kernels generated as the
//    reference does, `scale` folded into the smoothing kernel, then the separable engine.
//
// The kernels in this file are the GENERAL path: one thread per output element, taps read through L1/L2.  They are
// correct for every depth/border/anchor/ROI combination above.  The 4K 8U 3x3 configuration of BASELINE.json has
// its own fast path (TODO next round: register-rolling 3x3).
#include "rt.h"
#include <climits>
#include "roll.h"
#include "seproll.h"
#include "seplong.h"
#include "sepmx.h"
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

using namespace mi355;

namespace {

enum { D8U = MI355CV_8U, D16U = MI355CV_16U, D16S = MI355CV_16S, D32F = MI355CV_32F, D64F = MI355CV_64F };

__device__ __forceinline__ float ldF(const uchar* row, int idx, int depth)
{
    switch (depth) {
    case D8U:  return (float)row[idx];
    case D16U: return (float)reinterpret_cast<const unsigned short*>(row)[idx];
    case D16S: return (float)reinterpret_cast<const short*>(row)[idx];
    default:   return reinterpret_cast<const float*>(row)[idx];
    }
}

// saturate_cast<DT>(float): cvRound (round-half-even) then clamp (core/saturate.hpp:103-142)
__device__ __forceinline__ void stF(uchar* row, int idx, int depth, float s)
{
    switch (depth) {
    case D8U:  { float r = rintf(s); r = fminf(fmaxf(r, 0.f), 255.f); row[idx] = (uchar)(int)r; break; }
    case D16U: { float r = rintf(s); r = fminf(fmaxf(r, 0.f), 65535.f); reinterpret_cast<unsigned short*>(row)[idx] = (unsigned short)(int)r; break; }
    case D16S: { float r = rintf(s); r = fminf(fmaxf(r, -32768.f), 32767.f); reinterpret_cast<short*>(row)[idx] = (short)(int)r; break; }
    default:   reinterpret_cast<float*>(row)[idx] = s;
    }
}

struct Tap2D { float k; int dx, dy; };

__global__ __launch_bounds__(256) void k_filter2d_generic(
    const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
    int W, int H, int cn, int sdepth, int ddepth, int fullW, int fullH, int offX, int offY,
    const Tap2D* __restrict__ taps, int ntaps, int ax, int ay, int kw, int kh, float delta, int border)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, ch = e - x * cn;
    const int fx = x + offX - ax, fy = y + offY - ay;              // full-image coordinates of tap (0,0)
    float s = delta;
    if (fx >= 0 && fy >= 0 && fx + kw <= fullW && fy + kh <= fullH) {   // interior: no border logic
        const uchar* base = src + (ptrdiff_t)(y - ay) * (ptrdiff_t)sstep;
        const int e0 = (x - ax) * cn + ch;
        for (int t = 0; t < ntaps; t++) {
            const Tap2D tp = taps[t];
            s = __builtin_fmaf(ldF(base + (ptrdiff_t)tp.dy * (ptrdiff_t)sstep, e0 + tp.dx * cn, sdepth), tp.k, s);
        }
    } else {
        for (int t = 0; t < ntaps; t++) {
            const Tap2D tp = taps[t];
            const int yy = mi355_borderInterpolate(fy + tp.dy, fullH, border);
            const int xx = mi355_borderInterpolate(fx + tp.dx, fullW, border);
            float v = 0.f;
            if (yy >= 0 && xx >= 0)
                v = ldF(src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep, (xx - offX) * cn + ch, sdepth);
            s = __builtin_fmaf(v, tp.k, s);
        }
    }
    stF(dst + (size_t)y * dstep, e, ddepth, s);
}

// ---------------------------------------------------------------------------------- filter2D, LDS tile (kernels the rolling path does not take)
// The reference's non-DFT engine (Filter2D<ST, Cast<float, DT>, FilterVec_*>, filter.simd.hpp:2146-2330, 3027-3075): s = delta, then s = fma(p, k, s) over the NON-ZERO taps
// in raster order, one rounding to the destination depth.  k_filter2d_generic does that with one gather per tap and output (~8 instructions per tap: 11 x 11 on a 4K frame
// 777 us); here a workgroup stages the source box of 64 x 32 outputs of ONE channel ((32 + kh - 1) rows x (64 + kw - 1) columns, border rule resolved, converted to float) in
// LDS once.  A lane owns 4 neighbouring outputs in each of the rows ly and ly + 16, and the box is stored as PAIRS (row r, row r + 16): every multiply-add is one half of a
// v_pk_fma_f32 whose other half is the same tap of the partner row -- the operands arrive paired from one ds_read_b128, no register shuffling.  Along a kernel row the lane
// slides a window of pairs; the taps of the row are scalar operands, fetched one kernel row ahead (4 NC dwords, NC = ceil(kw / 4) a template parameter so that they stay in
// scalar registers); the row's non-zero mask lets dense rows run without per-tap tests and keeps zero taps out of the chain exactly as preprocess2DKernel does.  Per output
// the chain is the generic kernel's: the same products in the same order.
constexpr int FT_W = 64, FT_H = 32, FT_HH = FT_H / 2;
struct TileArgs { int W, H, cn, ddepth, fullW, fullH, offX, offY, kw, kh, ax, ay, border, pitch /* pairs per LDS row */, ncols; float delta; };
typedef float f32x2t __attribute__((ext_vector_type(2)));

template <typename ST> __device__ __forceinline__ float ldT(const uchar* row, int idx) { return (float)reinterpret_cast<const ST*>(row)[idx]; }

template <typename ST, int NC>
__global__ __launch_bounds__(256) void k_filter2d_tile(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                       TileArgs a, const float* __restrict__ kd /* kh x 4 NC, zero padded */, const unsigned* __restrict__ km /* kh row masks */)
{
    extern __shared__ __attribute__((aligned(16))) float ftile[];                      // (FT_HH + kh - 1) rows of a.pitch pairs, then the kernel (kh x 4 NC floats)
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int f = blockIdx.z / a.cn, ch = blockIdx.z - f * a.cn;
    src += (size_t)f * sframe; dst += (size_t)f * dframe;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    // ---- stage: box (r, j) = pixel (x0 - ax + j, y0 - ay + r) of channel ch as a float (outside the parent image by the border rule; BORDER_CONSTANT: 0); it is the
    // first half of pair row r and the second half of pair row r - 16.  A wave takes every fourth row (row address and its border rule on the scalar unit), a lane the
    // columns lane and lane + 64, whose element offsets -- border rule applied -- it computes once.
    const int rows = FT_H + a.kh - 1, prow = FT_HH + a.kh - 1;
    const int bx = x0 - a.ax + a.offX, by = y0 - a.ay + a.offY;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    float* kl = ftile + 2 * (size_t)prow * a.pitch;
    for (int i = tid; i < a.kh * 4 * NC; i += 256) kl[i] = kd[i];
    int xo[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int xx = bx + lane + 64 * q;
        if ((unsigned)xx >= (unsigned)a.fullW) xx = mi355_borderInterpolate(xx, a.fullW, a.border);
        xo[q] = (xx >= 0 && lane + 64 * q < a.ncols) ? (xx - a.offX) * a.cn + ch : INT_MIN;      // (columns left of a ROI window have negative offsets: INT_MIN = none)
    }
    const bool two = a.ncols > 64;                                                      // (uniform)
    for (int r = wave; r < rows; r += 4) {
        int yy = by + r;
        if ((unsigned)yy >= (unsigned)a.fullH) yy = mi355_borderInterpolate(yy, a.fullH, a.border);
        yy = __builtin_amdgcn_readfirstlane(yy);
        const uchar* srow = src + (ptrdiff_t)(max(yy, 0) - a.offY) * (ptrdiff_t)sstep;
        float v0 = 0.f, v1 = 0.f;
        if (yy >= 0) {
            if (xo[0] != INT_MIN) v0 = ldT<ST>(srow, xo[0]);
            if (two && xo[1] != INT_MIN) v1 = ldT<ST>(srow, xo[1]);
        }
        if (r < prow) {
            float* p0 = ftile + 2 * (size_t)r * a.pitch;
            p0[2 * lane] = v0;
            if (two && lane + 64 < a.ncols) p0[2 * (lane + 64)] = v1;
        }
        if (r >= FT_HH) {
            float* p1 = ftile + 2 * (size_t)(r - FT_HH) * a.pitch + 1;
            p1[2 * lane] = v0;
            if (two && lane + 64 < a.ncols) p1[2 * (lane + 64)] = v1;
        }
    }
    __syncthreads();
    const f32x2t d2 = {a.delta, a.delta};
    f32x2t acc[4] = {d2, d2, d2, d2};
    typedef float f4 __attribute__((ext_vector_type(4)));
    const unsigned lead = (1u << (4 * (NC - 1))) - 1u;                                 // the taps of the whole chunks before the last one
    for (int dy = 0; dy < a.kh; dy++) {
        const unsigned mc = km[dy];                                                      // (uniform: a scalar load)
        const f4* rp = reinterpret_cast<const f4*>(ftile + 2 * ((size_t)(ly + dy) * a.pitch + 4 * lx));
        const f4* kp = reinterpret_cast<const f4*>(kl + (size_t)dy * 4 * NC);          // the row's taps: every lane reads the same address (one broadcast per ds_read_b128)
        f32x2t w[4 + 4 * NC];
        float kc[4 * NC];
#pragma unroll
        for (int q = 0; q < 2 + 2 * NC; q++) { const f4 v = rp[q]; w[2 * q] = f32x2t{v.x, v.y}; w[2 * q + 1] = f32x2t{v.z, v.w}; }
#pragma unroll
        for (int q = 0; q < NC; q++) { const f4 v = kp[q]; kc[4 * q] = v.x; kc[4 * q + 1] = v.y; kc[4 * q + 2] = v.z; kc[4 * q + 3] = v.w; }
        // a zero tap stays out of the chain (its product could be NaN, or flip a -0): REAL uniform branches -- the empty asm keeps the compiler from turning them into
        // eight selects per tap; a dense row runs its whole chunks without any test
        auto tap = [&](int t) {
            const f32x2t k2 = {kc[t], kc[t]};
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_elementwise_fma(w[t + i], k2, acc[i]);
        };
        if ((mc & lead) == lead) {
#pragma unroll
            for (int t = 0; t < 4 * (NC - 1); t++) tap(t);
        } else {
#pragma unroll
            for (int t = 0; t < 4 * (NC - 1); t++) if ((mc >> t) & 1u) { asm volatile(""); tap(t); }
        }
#pragma unroll
        for (int t = 4 * (NC - 1); t < 4 * NC; t++) if ((mc >> t) & 1u) { asm volatile(""); tap(t); }
    }
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
        const int y = y0 + ly + FT_HH * hh;
        if (y >= a.H) continue;
        uchar* drow = dst + (size_t)y * dstep;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = x0 + 4 * lx + i;
            if (x < a.W) stF(drow, x * a.cn + ch, a.ddepth, hh ? acc[i].y : acc[i].x);
        }
    }
}

template <typename ST>
static void launchFilterTile(int nc, dim3 grid, size_t lds, hipStream_t st, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe,
                             const TileArgs& a, const float* dk, const unsigned* dm)
{
    switch (nc) {
#define FT_CASE(N_) case N_: hipLaunchKernelGGL((k_filter2d_tile<ST, N_>), grid, dim3(256), lds, st, src, sstep, sframe, dst, dstep, dframe, a, dk, dm); break
    FT_CASE(1); FT_CASE(2); FT_CASE(3); FT_CASE(4); FT_CASE(5); FT_CASE(6); FT_CASE(7); FT_CASE(8);
#undef FT_CASE
    }
}

// ---------------------------------------------------------------------------------- filter2D, register-rolling fast path
// CV_8U -> CV_8U, K x K taps (K = 3 or 5), centred anchor: the skeleton of roll.h with the last K source rows kept as
// floats in registers.  Per output: s = delta; s = fma(float(p), k, s) over ALL K*K taps in raster order (a zero tap
// leaves s unchanged bit for bit, so skipping them as the reference does changes nothing), cvRound, saturate --
// FilterVec_8u, filter.simd.hpp:2146-2210.  HBM-bound: 2*cn bytes per pixel.
struct DenseTaps { float k[25]; float delta; };

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The float window of a row is held as pairs Q[m] = (p[m], p[m+8]) so that outputs i and i+8 of the lane advance together
// through v_pk_fma_f32 (two FMAs per issue, tap broadcast by op_sel): the kernel would otherwise be VALU-bound at half the
// HBM rate.  Each half keeps its own chain, so the accumulation order per pixel is still the raster order of the taps.
// How a row reaches the lane: PlainRows loads the lane's 16 bytes of a CV_8U image as they are; GrayRows<SCN> loads the 16 x SCN bytes of a
// BGR(A) / RGB(A) image and turns them into the 16 gray bytes cv::cvtColor would have written (RGB2Gray<uchar>, color_rgb.simd.hpp:660-748:
// (c0 k0 + c1 k1 + c2 k2 + 2^14) >> 15) before anything else sees them -- the fused cvtColor -> filter2D pass of SURVEY §8d, which reads the
// colour image once and never writes the gray one.
template <int K, int CN> struct PlainRows {
    typedef roll::Ctx<K / 2, K / 2, CN> Cx;
    typedef typename Cx::RawT Raw;
    __device__ __forceinline__ void issue(const Cx& cx, Raw& r, int j, int& valid) const { cx.issue(r, j, valid); }
    __device__ __forceinline__ const typename Cx::RawT& bytes(const Raw& r) const { return r; }
};
template <int K, int SCN> struct GrayRows {
    typedef roll::Ctx<K / 2, K / 2, 1> Cx;
    static_assert(Cx::HD == 1 && Cx::MD == 4, "one side dword, 16-byte chunks");
    struct Raw { uint32_t w[4 * SCN]; uint32_t s[SCN]; };
    uint32_t k0, k1, k2;
    __device__ __forceinline__ void issue(const Cx& cx, Raw& r, int j, int& valid) const
    {
        const int ry = cx.rowIdx(cx.gy(min(j, cx.nrows - 1 + K / 2)));
        valid = ry >= 0;
        const uchar* row = cx.src + (size_t)max(ry, 0) * cx.sstep;                    // cx.src / sstep describe the COLOUR image
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
        for (int i = 0; i < SCN; i++) {
            const u32x4u v = *reinterpret_cast<const u32x4u*>(row + (size_t)SCN * cx.mainOff + 16 * i);
            r.w[4 * i] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < SCN; i++) r.s[i] = *reinterpret_cast<const uint32_t*>(row + (size_t)SCN * cx.sideOff + 4 * i);
    }
    // four gray bytes from pixels first .. first+3 (first a multiple of 4).  c0 k0 + c1 k1 + c2 k2 = 256 (c . khi) + (c . klo) with the coefficients split into
    // bytes, so each pixel costs two v_dot4_u32_u8 (the rounding constant rides in the second one's accumulator), one shift-add and one shift; a 3-byte
    // pixel is brought into one dword by v_alignbyte where it straddles two, and the unused fourth byte meets a zero coefficient -- about half the
    // instructions of extracting the bytes and chaining multiply-adds, in a kernel whose bound is the VALU
    template <int NPX> __device__ __forceinline__ uint32_t gray4(const uint32_t* w, int first) const
    {
        const uint32_t KH = (k0 >> 8) | ((k1 >> 8) << 8) | ((k2 >> 8) << 16), KL = (k0 & 255u) | ((k1 & 255u) << 8) | ((k2 & 255u) << 16);
        uint32_t px[4], acc = 0;
        bool hiPos[4] = {false, false, false, false};                                 // pixel bytes at positions 1..3 of the dword instead of 0..2
        if (SCN == 4) {
#pragma unroll
            for (int p = 0; p < 4; p++) px[p] = w[first + p];
        } else {
            const uint32_t* g = w + (first / 4) * 3;
            px[0] = g[0];
            px[1] = __builtin_amdgcn_alignbyte(g[1], g[0], 3);
            px[2] = __builtin_amdgcn_alignbyte(g[2], g[1], 2);
            px[3] = g[2]; hiPos[3] = true;
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t kh = hiPos[p] ? KH << 8 : KH, kl = hiPos[p] ? KL << 8 : KL;
            const uint32_t hi = __builtin_amdgcn_udot4(px[p], kh, 0u, false);
            const uint32_t lo = __builtin_amdgcn_udot4(px[p], kl, 1u << 14, false);
            acc |= (((hi << 8) + lo) >> 15) << (8 * p);
        }
        return acc;
    }
    __device__ __forceinline__ typename Cx::RawT bytes(const Raw& r) const
    {
        typename Cx::RawT g;
#pragma unroll
        for (int q = 0; q < 4; q++) g.m[q] = gray4<16>(r.w, 4 * q);
        g.side[0] = gray4<4>(r.s, 0);
        return g;
    }
};

template <int K, int CN, bool UP, typename Rows>
__device__ __forceinline__ void filterRows(roll::Ctx<K / 2, K / 2, CN>& cx, uchar* __restrict__ dst, size_t dstep, const DenseTaps& t, const Rows& rows)
{
    constexpr int R = K / 2, HB = R * CN, HD = roll::Cfg<R, CN>::HD, NW = roll::Cfg<R, CN>::NW, NQ = 8 + 2 * HB;
    f32x2 Q[K][NQ];
    auto toFloat = [&](f32x2 (&q)[NQ], const typename Rows::Raw& rr, int valid) {
        if (!valid) {
#pragma unroll
            for (int i = 0; i < NQ; i++) q[i] = f32x2{0.f, 0.f};
            return;
        }
        uint32_t X[NW];
        const typename roll::Ctx<K / 2, K / 2, CN>::RawT r = rows.bytes(rr);
        cx.window(X, r);
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const int b0 = 4 * HD - HB + i, b1 = b0 + 8;
            q[i].x = (float)((X[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu);
            q[i].y = (float)((X[b1 >> 2] >> (8 * (b1 & 3))) & 0xffu);
        }
    };
#pragma unroll
    for (int i = 0; i < K - 1; i++) {                      // prologue: logical rows -R .. R-1
        typename Rows::Raw pre; int v;
        rows.issue(cx, pre, i - R, v);
        toFloat(Q[i], pre, v);
    }
    typename Rows::Raw raw[K]; int rv[K];
#pragma unroll
    for (int u = 0; u < K; u++) rows.issue(cx, raw[u], u + R, rv[u]);
    for (int y = 0; y < cx.nrows; y += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            if (y + u < cx.nrows) {
                toFloat(Q[(K - 1 + u) % K], raw[u], rv[u]);
                rows.issue(cx, raw[u], y + u + K + R, rv[u]);
                uint32_t o[4] = {0, 0, 0, 0};
                if constexpr (K == 3) {
                    // taps outermost: a zero tap is skipped by a scalar branch (the taps are wave-uniform), as the reference skips it -- the sums are the
                    // same bit for bit either way, and a 3x3 sharpen / Laplacian then costs 5 of the 9 packed FMAs per output pair (the kernel is
                    // VALU-bound at 3x3 already); the chain of every output still runs in raster order of the taps
                    f32x2 s[8];
    #pragma unroll
                    for (int i = 0; i < 8; i++) s[i] = f32x2{t.delta, t.delta};
    #pragma unroll
                    for (int dy = 0; dy < K; dy++) {
                        // image row (y - R + dy) sits in slot (u + dy) when walking down, (u + K-1-dy) when walking up
                        const f32x2* qr = Q[(u + (UP ? K - 1 - dy : dy)) % K];
    #pragma unroll
                        for (int dx = 0; dx < K; dx++) {
                            const float kv = t.k[dy * K + dx];
                            if (__builtin_amdgcn_readfirstlane(__float_as_int(kv)) != 0) {
    #pragma unroll
                                for (int i = 0; i < 8; i++) s[i] = __builtin_elementwise_fma(qr[i + dx * CN], f32x2{kv, kv}, s[i]);
                            }
                        }
                    }
    #pragma unroll
                    for (int i = 0; i < 8; i++) {
                        // cvRound + saturate_cast<uchar>: round half-even, then the saturating byte conversion
                        o[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(s[i].x), i & 3, o[i >> 2]);
                        o[2 + (i >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(s[i].y), i & 3, o[2 + (i >> 2)]);
                    }
                } else {
                    // 5x5: outputs outermost (the eight accumulator pairs of the other order cost a wave per SIMD here)
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        f32x2 s = {t.delta, t.delta};
#pragma unroll
                        for (int dy = 0; dy < K; dy++) {
                            const f32x2* qr = Q[(u + (UP ? K - 1 - dy : dy)) % K];
#pragma unroll
                            for (int dx = 0; dx < K; dx++) {
                                const float kv = t.k[dy * K + dx];
                                s = __builtin_elementwise_fma(qr[i + dx * CN], f32x2{kv, kv}, s);
                            }
                        }
                        o[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(s.x), i & 3, o[i >> 2]);
                        o[2 + (i >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(s.y), i & 3, o[2 + (i >> 2)]);
                    }
                }
                cx.template store<1>(dst, dstep, cx.gy(y + u), o);
            }
        }
    }
}

template <int K, int CN>
__global__ __launch_bounds__(256) void k_filter2d_roll(const uchar* __restrict__ src, size_t sstep, size_t sframe,
                                                       uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                       int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes, int border, int alt, DenseTaps t)
{
    roll::Ctx<K / 2, K / 2, CN> cx;
    if (!cx.init(src, sstep, sframe, W, H, nchunks, nstrips, segRows, nseg, nframes, border, alt)) return;
    dst += (size_t)cx.frame * dframe;
    const PlainRows<K, CN> rows;
    if (cx.up) filterRows<K, CN, true>(cx, dst, dstep, t, rows);
    else       filterRows<K, CN, false>(cx, dst, dstep, t, rows);
}


// cv::filter2D on CV_32FC1 -> CV_32FC1, 3x3 / 5x5, on the rolling skeleton: a float is a pixel of four "channels" of the byte skeleton (roll.h), so halos
// are whole floats and the window's dwords are the elements.  The K row windows stay in registers; an output is delta + the taps in raster order, one
// FMA each (Filter2D<float, Cast<float, float>, FilterVec_32f>, filter.simd.hpp:3103-3190: s0 = delta, s0 = fma(kf[k], src[k], s0) over the non-zero taps
// -- a zero tap leaves the sum unchanged bit for bit).
template <int K, bool UP>
__device__ __forceinline__ void filterRowsF32(roll::Ctx<K / 2, K / 2, 4>& cx, uchar* __restrict__ dst, size_t dstep, const DenseTaps& t)
{
    typedef roll::Ctx<K / 2, K / 2, 4> Cx;
    typedef typename Cx::RawT RawT;
    constexpr int R = K / 2, NW = Cx::NW;
    uint32_t Q[K][NW];
    auto win = [&](uint32_t (&q)[NW], RawT raw, int valid) {
        if (!valid) {
#pragma unroll
            for (int d = 0; d < Cx::MD; d++) raw.m[d] = 0;
#pragma unroll
            for (int d = 0; d < Cx::HD; d++) raw.side[d] = 0;
        }
        cx.window(q, raw);
    };
#pragma unroll
    for (int i = 0; i < K - 1; i++) { RawT pre; int v; cx.issue(pre, i - R, v); win(Q[i], pre, v); }
    RawT raw[K]; int rv[K];
#pragma unroll
    for (int u = 0; u < K; u++) cx.issue(raw[u], u + R, rv[u]);
    for (int y = 0; y < cx.nrows; y += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            if (y + u < cx.nrows) {
                win(Q[(K - 1 + u) % K], raw[u], rv[u]);
                cx.issue(raw[u], y + u + K + R, rv[u]);
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float s = t.delta;
#pragma unroll
                    for (int dy = 0; dy < K; dy++) {
                        const uint32_t* qr = Q[(u + (UP ? K - 1 - dy : dy)) % K];     // image row y - R + dy
#pragma unroll
                        for (int dx = 0; dx < K; dx++) s = __builtin_fmaf(t.k[dy * K + dx], __uint_as_float(qr[k + dx]), s);
                    }
                    o[k] = __float_as_uint(s);
                }
                cx.template store<1>(dst, dstep, cx.gy(y + u), o);
            }
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_filter2d_roll_f32(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                           int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes, int border, roll::Win win, DenseTaps t)
{
    roll::Ctx<K / 2, K / 2, 4> cx;
    if (!cx.init(src, sstep, sframe, W, H, nchunks, nstrips, segRows, nseg, nframes, border, 1, win)) return;
    dst += (size_t)cx.frame * dframe;
    if (cx.up) filterRowsF32<K, true>(cx, dst, dstep, t);
    else       filterRowsF32<K, false>(cx, dst, dstep, t);
}

// cvtColor(BGR2GRAY / RGB2GRAY / BGRA2GRAY / RGBA2GRAY) followed by filter2D, in one pass: W is the width in pixels, the work split is that of
// the gray image (a lane = 16 gray pixels = 16 SCN colour bytes)
template <int K, int SCN>
__global__ __launch_bounds__(256) void k_gray_filter2d_roll(const uchar* __restrict__ src, size_t sstep, size_t sframe,
                                                            uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                            int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes, int border, int alt, DenseTaps t,
                                                            int k0, int k1, int k2)
{
    roll::Ctx<K / 2, K / 2, 1> cx;
    if (!cx.init(src, sstep, sframe, W, H, nchunks, nstrips, segRows, nseg, nframes, border, alt)) return;
    dst += (size_t)cx.frame * dframe;
    GrayRows<K, SCN> rows; rows.k0 = (uint32_t)k0; rows.k1 = (uint32_t)k1; rows.k2 = (uint32_t)k2;
    if (cx.up) filterRows<K, 1, true>(cx, dst, dstep, t, rows);
    else       filterRows<K, 1, false>(cx, dst, dstep, t, rows);
}

// ---------------------------------------------------------------------------------- separable
template <int MAXK> struct SepParamsT {
    float kxf[MAXK], kyf[MAXK];
    int   kxi[MAXK], kyi[MAXK];
    int nx, ny, ax, ay;
    int mode;        // 0 float, 1 int Q8 x Q8 (8U->8U), 2 int exact (8U->16S)
    int symY;        // 1 symmetrical pair form, 2 anti-symmetrical pair form, 0 plain chain
    float deltaF;
    int deltaI;
};
typedef SepParamsT<33> SepParams;            // what every kernel up to 33 taps per axis uses

template <int MAXK>
__global__ __launch_bounds__(256) void k_sepfilter_generic(
    const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
    int W, int H, int cn, int sdepth, int ddepth, int fullW, int fullH, int offX, int offY, int border, SepParamsT<MAXK> p)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, ch = e - x * cn;
    const int fx0 = x + offX - p.ax, fy0 = y + offY - p.ay;
    int xs[MAXK];
    for (int i = 0; i < p.nx; i++) {
        int xx = mi355_borderInterpolate(fx0 + i, fullW, border);
        xs[i] = xx < 0 ? INT_MIN : (xx - offX) * cn + ch;      // offsets are relative to the ROI: negative ones are real pixels of the parent
    }
    if (p.mode != 0) {
        // integer modes: order of summation is irrelevant
        long long acc = p.deltaI;
        int ri[MAXK];
        for (int j = 0; j < p.ny; j++) {
            const int yy = mi355_borderInterpolate(fy0 + j, fullH, border);
            ri[j] = 0;
            if (yy < 0) continue;
            const uchar* row = src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep;
            int rs = 0;
            for (int i = 0; i < p.nx; i++) if (xs[i] != INT_MIN) rs += p.kxi[i] * (int)row[xs[i]];
            ri[j] = rs;
            acc += (long long)p.kyi[j] * rs;
        }
        int a = (int)acc;                                      // the reference accumulates in int32 (wraps identically)
        if (p.mode == 1 && p.ny > 1 && e < ((W * cn) & ~15)) {
            // What the reference's AVX2 build computes for every element its 16-lane loop reaches
            // (SymmColumnVec_32s8u, filter.simd.hpp:1011-1085): int32 row sums combined in FLOAT (taps * 2^-16, FMA
            // chain, round-half-even).  Only its scalar tail (below) uses the integer (v + 2^15) >> 16 form.
            const int c = p.ay;
            float sF = __builtin_fmaf((float)ri[c], (float)p.kyi[c] * (1.0f / 65536.0f), p.deltaF);
            for (int k = 1; k <= p.ny / 2; k++)
                sF = __builtin_fmaf((float)(ri[c + k] + ri[c - k]), (float)p.kyi[c + k] * (1.0f / 65536.0f), sF);
            float r = rintf(sF);
            dst[(size_t)y * dstep + e] = (uchar)(int)fminf(fmaxf(r, 0.f), 255.f);
        } else if (p.mode == 1) {
            int r = (a + (1 << 15)) >> 16;
            dst[(size_t)y * dstep + e] = (uchar)(r < 0 ? 0 : r > 255 ? 255 : r);
        } else {
            reinterpret_cast<short*>(dst + (size_t)y * dstep)[e] = (short)(a < -32768 ? -32768 : a > 32767 ? 32767 : a);
        }
        return;
    }
    // float mode: row sums as RowFilter does, for the ny rows this output needs
    auto rowSum = [&](int j) -> float {
        const int yy = mi355_borderInterpolate(fy0 + j, fullH, border);
        const uchar* row = src + (ptrdiff_t)((yy < 0 ? offY : yy) - offY) * (ptrdiff_t)sstep;
        float s = 0.f;
        for (int i = 0; i < p.nx; i++) {
            const float v = (yy < 0 || xs[i] == INT_MIN) ? 0.f : ldF(row, xs[i], sdepth);
            s = i == 0 ? p.kxf[0] * v : __builtin_fmaf(p.kxf[i], v, s);
        }
        return s;
    };
    float s;
    if (p.symY) {
        const int c = p.ay;                                    // centred anchor is part of the symmetry test
        s = __builtin_fmaf(p.kyf[c], rowSum(c), p.deltaF);
        if (p.symY == 2) s = p.deltaF;
        for (int k = 1; k <= p.ny / 2; k++) {
            const float a = rowSum(c + k), b = rowSum(c - k);
            s = __builtin_fmaf(p.kyf[c + k], p.symY == 1 ? a + b : a - b, s);
        }
    } else {
        s = __builtin_fmaf(p.kyf[0], rowSum(0), p.deltaF);
        for (int j = 1; j < p.ny; j++) s = __builtin_fmaf(p.kyf[j], rowSum(j), s);
    }
    stF(dst + (size_t)y * dstep, e, ddepth, s);
}


// ---------------------------------------------------------------------------------- CV_64F destinations (and sources)
// cv::filter2D / cv::sepFilter2D with a CV_64F source or destination run the reference's engines with double kernels and double intermediate rows (kdepth / bdepth =
// CV_64F, filter.simd.hpp:3192-3210, filter.dispatch.cpp:318-330): Filter2D<ST, Cast<double, double>, FilterNoVec> -- s = delta, s += k * src over the non-zero taps in
// raster order --, RowFilter<ST, double, RowNoVec> -- kx[0] * S[0], then s += kx[i] * S[i] -- and ColumnFilter / SymmColumnFilter<Cast<double, double>, ColumnNoVec>
// (the pair forms for odd (anti)symmetric kernels).  The running copy of those loops is the AVX2 + FMA one (filter.simd.hpp is a dispatched file), where the compiler
// fuses every `s += a * b`: fma() here, as fmaf() in the float kernels above.  One thread per output element; double arithmetic at half rate on a path nobody
// benchmarks -- what matters is that a CV_64F Mat does not fall off the GPU.
__device__ __forceinline__ double ldD(const uchar* row, int idx, int depth)
{
    switch (depth) {
    case D8U:  return (double)row[idx];
    case D16U: return (double)reinterpret_cast<const unsigned short*>(row)[idx];
    case D16S: return (double)reinterpret_cast<const short*>(row)[idx];
    case D32F: return (double)reinterpret_cast<const float*>(row)[idx];
    default:   return reinterpret_cast<const double*>(row)[idx];
    }
}

struct Tap2D64 { double k; int dx, dy; };

__global__ __launch_bounds__(256) void k_filter2d_generic64(
    const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
    int W, int H, int cn, int sdepth, int fullW, int fullH, int offX, int offY,
    const Tap2D64* __restrict__ taps, int ntaps, int ax, int ay, double delta, int border)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, ch = e - x * cn;
    const int fx = x + offX - ax, fy = y + offY - ay;              // full-image coordinates of tap (0,0)
    double s = delta;
    for (int t = 0; t < ntaps; t++) {
        const Tap2D64 tp = taps[t];
        const int yy = mi355_borderInterpolate(fy + tp.dy, fullH, border);
        const int xx = mi355_borderInterpolate(fx + tp.dx, fullW, border);
        double v = 0.0;
        if (yy >= 0 && xx >= 0) v = ldD(src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep, (xx - offX) * cn + ch, sdepth);
        s = __builtin_fma(tp.k, v, s);
    }
    reinterpret_cast<double*>(dst + (size_t)y * dstep)[e] = s;
}

struct SepParams64 { double kx[33], ky[33]; int nx, ny, ax, ay, symY; double delta; };

__global__ __launch_bounds__(256) void k_sepfilter_generic64(
    const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
    int W, int H, int cn, int sdepth, int fullW, int fullH, int offX, int offY, int border, SepParams64 p)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, ch = e - x * cn;
    const int fx0 = x + offX - p.ax, fy0 = y + offY - p.ay;
    int xs[33];
    for (int i = 0; i < p.nx; i++) {
        const int xx = mi355_borderInterpolate(fx0 + i, fullW, border);
        xs[i] = xx < 0 ? INT_MIN : (xx - offX) * cn + ch;
    }
    auto rowSum = [&](int j) -> double {
        const int yy = mi355_borderInterpolate(fy0 + j, fullH, border);
        const uchar* row = src + (ptrdiff_t)((yy < 0 ? offY : yy) - offY) * (ptrdiff_t)sstep;
        double s = 0.0;
        for (int i = 0; i < p.nx; i++) {
            const double v = (yy < 0 || xs[i] == INT_MIN) ? 0.0 : ldD(row, xs[i], sdepth);
            s = i == 0 ? p.kx[0] * v : __builtin_fma(p.kx[i], v, s);
        }
        return s;
    };
    double s;
    if (p.symY) {
        const int c = p.ay;
        s = p.symY == 1 ? __builtin_fma(p.ky[c], rowSum(c), p.delta) : p.delta;
        for (int k = 1; k <= p.ny / 2; k++) {
            const double a = rowSum(c + k), b = rowSum(c - k);
            s = __builtin_fma(p.ky[c + k], p.symY == 1 ? a + b : a - b, s);
        }
    } else {
        s = __builtin_fma(p.ky[0], rowSum(0), p.delta);
        for (int j = 1; j < p.ny; j++) s = __builtin_fma(p.ky[j], rowSum(j), s);
    }
    reinterpret_cast<double*>(dst + (size_t)y * dstep)[e] = s;
}

// ---------------------------------------------------------------------------------- box filter (row a5)
// cv::boxFilter (box_filter.dispatch.cpp:440, createBoxFilter box_filter.simd.hpp:1250):
//   8U->8U, area <= 256 : u16 sums; normalised result = ((s + dd) * ds) >> 23 with the reciprocal pair (ds, dd) of
//                         ColumnSum<ushort,uchar> (:429-455); un-normalised = saturate_u8(s)
//   other integer inputs: int32 sums; normalised = cvRound(float(s) * float(1/area)) (the SIMD body of
//                         ColumnSum<int,uchar> :340-372), un-normalised = saturate(s)
//   float input         : double sums (RowSum<float,double>, ColumnSum<double,float>), result = float(s * scale)
struct BoxParams { int kw, kh, ax, ay, normalize, mode /*0 u16, 1 int, 2 double*/, divScale, divDelta; float scaleF; double scaleD; };

__global__ __launch_bounds__(256) void k_box_generic(
    const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
    int W, int H, int cn, int sdepth, int ddepth, int fullW, int fullH, int offX, int offY, int border, BoxParams p)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, ch = e - x * cn;
    const int fx0 = x + offX - p.ax, fy0 = y + offY - p.ay;
    uchar* drow = dst + (size_t)y * dstep;
    if (p.mode == 2) {
        double s = 0.0;
        for (int j = 0; j < p.kh; j++) {
            const int yy = mi355_borderInterpolate(fy0 + j, fullH, border);
            if (yy < 0) continue;
            const uchar* row = src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep;
            double rs = 0.0;
            for (int i = 0; i < p.kw; i++) {
                const int xx = mi355_borderInterpolate(fx0 + i, fullW, border);
                if (xx >= 0) rs += sdepth == D64F ? reinterpret_cast<const double*>(row)[(xx - offX) * cn + ch] : (double)reinterpret_cast<const float*>(row)[(xx - offX) * cn + ch];
            }
            s += rs;
        }
        if (ddepth == D64F) reinterpret_cast<double*>(drow)[e] = p.normalize ? s * p.scaleD : s;
        else reinterpret_cast<float*>(drow)[e] = (float)(p.normalize ? s * p.scaleD : s);
        return;
    }
    int s = 0;
    for (int j = 0; j < p.kh; j++) {
        const int yy = mi355_borderInterpolate(fy0 + j, fullH, border);
        if (yy < 0) continue;
        const uchar* row = src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep;
        for (int i = 0; i < p.kw; i++) {
            const int xx = mi355_borderInterpolate(fx0 + i, fullW, border);
            if (xx < 0) continue;
            const int idx = (xx - offX) * cn + ch;
            s += sdepth == D8U ? (int)row[idx] : sdepth == D16U ? (int)reinterpret_cast<const unsigned short*>(row)[idx]
                                                              : (int)reinterpret_cast<const short*>(row)[idx];
        }
    }
    if (p.mode == 0) {
        unsigned r = p.normalize ? (((unsigned)s + (unsigned)p.divDelta) * (unsigned)p.divScale) >> 23 : (unsigned)s;
        drow[e] = (uchar)(p.normalize ? r : (r > 255u ? 255u : r));
        return;
    }
    if (ddepth == D64F) {           // ColumnSum<int, double> (box_filter.simd.hpp:1195-1240): the exact int sum, one multiply in double
        reinterpret_cast<double*>(drow)[e] = p.normalize ? (double)s * p.scaleD : (double)s;
        return;
    }
    if (ddepth == D32F) {
        // ColumnSum<int, float> (box_filter.simd.hpp:1109-1130): the vector body multiplies in float, the last (W*cn) % 4 elements of a row in double
        reinterpret_cast<float*>(drow)[e] = !p.normalize ? (float)s : e < ((W * cn) & ~3) ? __fmul_rn((float)s, p.scaleF) : (float)((double)s * p.scaleD);
        return;
    }
    if (p.normalize && e >= ((W * cn) & ~7)) {
        // the reference's scalar tail (the last (W*cn) % 8 elements of a row, box_filter.simd.hpp:380-385) multiplies in double
        const double r = rint((double)s * p.scaleD);
        if (ddepth == D8U) drow[e] = (uchar)(int)fmin(fmax(r, 0.0), 255.0);
        else if (ddepth == D16U) reinterpret_cast<unsigned short*>(drow)[e] = (unsigned short)(int)fmin(fmax(r, 0.0), 65535.0);
        else reinterpret_cast<short*>(drow)[e] = (short)(int)fmin(fmax(r, -32768.0), 32767.0);
        return;
    }
    float v = p.normalize ? rintf((float)s * p.scaleF) : (float)s;
    stF(drow, e, ddepth, v);
}

// ---------------------------------------------------------------------------------- box filter in two passes (what neither the rolling kernels nor k_sepmx take)
// RowSum then ColumnSum like the reference (box_filter.simd.hpp:60-180, 182-1240) instead of k_box_generic's kw * kh gathers per output (a 121 x 121 window of a guided
// filter on a 4K float frame: 14 641 loads per element, slower than the CPU's running sums):
//   k_box_rows   window sums along x of every parent row the call touches, into a scratch image of int (integer sources: exact) or double (float sources: the reference's
//                sum type) -- a workgroup stages 1024 + kw - 1 values of one channel of one row in LDS (border rule resolved there), a lane takes 4 neighbouring windows:
//                one sum of kw values, three slides;
//   k_box_cols   a lane owns an output column and walks down a segment with the running column sum (add the row entering, subtract the row leaving -- ColumnSum's own
//                recurrence), and finishes every sum exactly as k_box_generic does (the reference's normalisations per depth pair).
// Integer results are those of k_box_generic bit for bit; double sums differ from it in the last bits of the DOUBLE (the order of the additions), i.e. not at all after
// the rounding to float in all but isolated elements -- the float bar of the path is 1e-4.
template <typename SUM> struct BoxStage { typedef int L; };
template <> struct BoxStage<double> { typedef float L; };

template <typename ST, typename SUM>
__global__ __launch_bounds__(256) void k_box_rows(const uchar* __restrict__ src, size_t sstep, SUM* __restrict__ R, size_t rpitch, int W, int cn, int fullW, int offX, int offY,
                                                  int r0, int kw, int ax, int border)
{
    typedef typename BoxStage<SUM>::L LT;
    extern __shared__ __attribute__((aligned(16))) uchar boxlds_[];
    LT* L = reinterpret_cast<LT*>(boxlds_);
    const int tid = threadIdx.x, ch = blockIdx.z, prow = r0 + blockIdx.y, xb = blockIdx.x * 1024;
    const ST* row = reinterpret_cast<const ST*>(src + (ptrdiff_t)(prow - offY) * (ptrdiff_t)sstep);
    const int nout = min(1024, W - xb), nst = nout + kw + 3;
    for (int j = tid; j < nst; j += 256) {
        int fx = xb + j + offX - ax;
        if ((unsigned)fx >= (unsigned)fullW) fx = mi355_borderInterpolate(fx, fullW, border);
        L[j] = (fx >= 0 && j < nout + kw - 1) ? (LT)row[(fx - offX) * cn + ch] : (LT)0;
    }
    __syncthreads();
    const int x = 4 * tid;
    if (x >= nout) return;
    typedef LT l4 __attribute__((ext_vector_type(4)));
    const LT* q = L + x;
    SUM s = 0;
    int t = 0;
    for (; t + 4 <= kw; t += 4) { const l4 v = *reinterpret_cast<const l4*>(q + t); s += (SUM)v.x; s += (SUM)v.y; s += (SUM)v.z; s += (SUM)v.w; }
    for (; t < kw; t++) s += (SUM)q[t];
    SUM* out = R + (size_t)blockIdx.y * rpitch + (size_t)(xb + x) * cn + ch;
    out[0] = s;
#pragma unroll
    for (int i = 1; i < 4; i++) {
        if (x + i >= nout) break;
        s += (SUM)q[kw + i - 1]; s -= (SUM)q[i - 1];
        out[(size_t)i * cn] = s;
    }
}

template <typename SUM>
__device__ __forceinline__ void boxFinish(uchar* drow, int e, SUM s, int W, int cn, int ddepth, const BoxParams& p);
template <>
__device__ __forceinline__ void boxFinish<double>(uchar* drow, int e, double s, int W, int cn, int ddepth, const BoxParams& p)
{
    if (ddepth == D64F) reinterpret_cast<double*>(drow)[e] = p.normalize ? s * p.scaleD : s;
    else reinterpret_cast<float*>(drow)[e] = (float)(p.normalize ? s * p.scaleD : s);
}
template <>
__device__ __forceinline__ void boxFinish<int>(uchar* drow, int e, int s, int W, int cn, int ddepth, const BoxParams& p)
{
    if (p.mode == 0) {
        unsigned r = p.normalize ? (((unsigned)s + (unsigned)p.divDelta) * (unsigned)p.divScale) >> 23 : (unsigned)s;
        drow[e] = (uchar)(p.normalize ? r : (r > 255u ? 255u : r));
        return;
    }
    if (ddepth == D64F) { reinterpret_cast<double*>(drow)[e] = p.normalize ? (double)s * p.scaleD : (double)s; return; }
    if (ddepth == D32F) { reinterpret_cast<float*>(drow)[e] = !p.normalize ? (float)s : e < ((W * cn) & ~3) ? __fmul_rn((float)s, p.scaleF) : (float)((double)s * p.scaleD); return; }
    if (p.normalize && e >= ((W * cn) & ~7)) {
        const double r = rint((double)s * p.scaleD);
        if (ddepth == D8U) drow[e] = (uchar)(int)fmin(fmax(r, 0.0), 255.0);
        else if (ddepth == D16U) reinterpret_cast<unsigned short*>(drow)[e] = (unsigned short)(int)fmin(fmax(r, 0.0), 65535.0);
        else reinterpret_cast<short*>(drow)[e] = (short)(int)fmin(fmax(r, -32768.0), 32767.0);
        return;
    }
    const float v = p.normalize ? rintf((float)s * p.scaleF) : (float)s;
    stF(drow, e, ddepth, v);
}

template <typename SUM>
__global__ __launch_bounds__(256) void k_box_cols(const SUM* __restrict__ R, size_t rpitch, int r0, uchar* __restrict__ dst, size_t dstep, int W, int H, int cn, int ddepth,
                                                  int fullH, int offY, int border, BoxParams p, int seg)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W * cn) return;
    const int y0 = blockIdx.y * seg, y1 = min(H, y0 + seg);
    auto rowSum = [&](int fy) -> SUM {                                                 // (fy is uniform: the row's address is scalar work)
        if ((unsigned)fy >= (unsigned)fullH) fy = mi355_borderInterpolate(fy, fullH, border);
        return fy < 0 ? (SUM)0 : R[(size_t)(fy - r0) * rpitch + e];
    };
    SUM s = 0;
    for (int j = 0; j < p.kh; j++) s += rowSum(y0 + offY - p.ay + j);
    for (int y = y0; y < y1; y++) {
        boxFinish<SUM>(dst + (size_t)y * dstep, e, s, W, cn, ddepth, p);
        if (y + 1 < y1) { s += rowSum(y + 1 + offY - p.ay + p.kh - 1); s -= rowSum(y + offY - p.ay); }
    }
}

// the two-pass box filter (k_box_rows / k_box_cols) for one image; false: not its case (the caller falls back to k_box_generic)
// parent rows the windows can touch: the window's own rows, and what a mirroring border rule folds back into them; BORDER_WRAP reaches across the image
static void boxRowRange(int H, int kh, int fullH, int offY, int border, int* r0, int* r1)
{
    *r0 = std::max(0, offY - kh); *r1 = std::min(fullH, offY + H + kh);
    if (border == B_WRAP) { *r0 = 0; *r1 = fullH; }
}

template <typename ST, typename SUM>
static bool boxTwoPassT(void* scratch, const BoxParams& p, const uchar* src, size_t sstep, uchar* dst, size_t dstep, int W, int H, int cn, int ddepth,
                        int fullW, int fullH, int offX, int offY, int border, hipStream_t st)
{
    int r0, r1;
    boxRowRange(H, p.kh, fullH, offY, border, &r0, &r1);
    const int nr = r1 - r0;
    const size_t rpitch = ((size_t)W * cn + 3) & ~(size_t)3;
    SUM* R = (SUM*)scratch;
    const size_t lds = (size_t)(1024 + p.kw + 8) * 4;
    hipLaunchKernelGGL((k_box_rows<ST, SUM>), dim3(divUp(W, 1024), nr, cn), dim3(256), lds, st, src, sstep, R, rpitch, W, cn, fullW, offX, offY, r0, p.kw, p.ax, border);
    const int seg = 32;
    hipLaunchKernelGGL((k_box_cols<SUM>), dim3(divUp(W * cn, 256), divUp(H, seg)), dim3(256), 0, st, R, rpitch, r0, dst, dstep, W, H, cn, ddepth, fullH, offY, border, p, seg);
    return true;
}

// ---------------------------------------------------------------------------------- host: contexts
int depthSize(int d) { return d == D8U ? 1 : (d == D16U || d == D16S) ? 2 : d == D32F ? 4 : d == D64F ? 8 : 0; }

double kernelAt(const uchar* data, size_t step, int type, int r, int c)
{
    const uchar* p = data + (size_t)r * step;
    switch (MI355CV_MAT_DEPTH(type)) {
    case MI355CV_8U:  return p[c];
    case MI355CV_8S:  return ((const signed char*)p)[c];
    case MI355CV_16U: return ((const unsigned short*)p)[c];
    case MI355CV_16S: return ((const short*)p)[c];
    case MI355CV_32S: return ((const int*)p)[c];
    case MI355CV_32F: return ((const float*)p)[c];
    default:          return ((const double*)p)[c];
    }
}

enum { K_GENERAL = 0, K_SYMMETRICAL = 1, K_ASYMMETRICAL = 2, K_SMOOTH = 4, K_INTEGER = 8 };

// cv::getKernelType (filter.dispatch.cpp:225-259)
int kernelType(const std::vector<double>& k, int anchor)
{
    const int sz = (int)k.size();
    int type = K_SMOOTH + K_INTEGER;
    if (anchor * 2 + 1 == sz) type |= K_SYMMETRICAL + K_ASYMMETRICAL;
    double sum = 0;
    for (int i = 0; i < sz; i++) {
        const double a = k[i], b = k[sz - i - 1];
        if (a != b) type &= ~K_SYMMETRICAL;
        if (a != -b) type &= ~K_ASYMMETRICAL;
        if (a < 0) type &= ~K_SMOOTH;
        if (a != (double)(int)nearbyint(a)) type &= ~K_INTEGER;
        sum += a;
    }
    if (std::fabs(sum - 1) > 1.1920928955078125e-7 * (std::fabs(sum) + 1)) type &= ~K_SMOOTH;
    return type;
}

// createBitExactKernel_32S (filter.dispatch.cpp:288-303)
bool bitExactKernel(const std::vector<double>& k, int bits, std::vector<int>& out)
{
    out.resize(k.size());
    const double eps = 10 * 1.1920928955078125e-7 * (1 << bits);
    for (size_t i = 0; i < k.size(); i++) {
        const double v = k[i] * (1 << bits);
        const int q = (int)nearbyint(v);
        out[i] = q;
        if (std::fabs(v - q) > eps) return false;
    }
    return true;
}

struct FilterCtx {
    int kind;                       // 1 filter2D, 2 separable
    int sdepth, ddepth, cn, border;
    int ax, ay, kw, kh;
    float delta;
    std::vector<Tap2D> taps;
    SepParams sp;                   // taps for the rolling kernels and k_sepfilter_generic<33> (at most 33 per axis; nx = 0 when the kernel is longer)
    // the same taps at any length up to lim::SEP_MAX_TAPS, for the LDS-ring kernel (seplong.hip)
    std::vector<float> lkxf, lkyf;
    std::vector<int> lkxi, lkyi;
    int lmode = 0, lsymY = 0, ldeltaI = 0, lnx = 0, lny = 0;
    float ldeltaF = 0.f;
    bool big = false;               // more than 33 taps on an axis: c.sp is not filled
    FilterCtx() = default;
    FilterCtx(const FilterCtx&) = delete;
    FilterCtx& operator=(const FilterCtx&) = delete;
    bool wide = false;              // CV_64F destination: double kernels and sums (k_filter2d_generic64 / k_sepfilter_generic64)
    double delta64 = 0;
    std::vector<Tap2D64> taps64;
    SepParams64 sp64;
};

// the CV_64F engines of the reference: filter2D from 8U / 16U / 16S / 64F (getLinearFilter, filter.simd.hpp:3230-3250), sepFilter2D also from 32F
bool widePairOk(int sd, int dd, bool separable)
{
    return dd == D64F && (sd == D8U || sd == D16U || sd == D16S || sd == D64F || (separable && sd == D32F));
}

bool depthPairOk(int sd, int dd)
{
    if (sd == D8U) return dd == D8U || dd == D16U || dd == D16S || dd == D32F;
    if (sd == D16U) return dd == D16U || dd == D32F;
    if (sd == D16S) return dd == D16S || dd == D32F;
    if (sd == D32F) return dd == D32F;
    return false;
}

int sepInit(FilterCtx& c, int stype, int dtype, const std::vector<double>& kx, const std::vector<double>& ky,
            int ax, int ay, double delta, int border)
{
    c.kind = 2;
    c.sdepth = MI355CV_MAT_DEPTH(stype); c.ddepth = MI355CV_MAT_DEPTH(dtype);
    c.cn = MI355CV_MAT_CN(stype);
    c.wide = c.cn == MI355CV_MAT_CN(dtype) && widePairOk(c.sdepth, c.ddepth, true);
    if (c.cn != MI355CV_MAT_CN(dtype) || !(c.wide || depthPairOk(c.sdepth, c.ddepth)))
        return setError(MI355CV_NOT_IMPLEMENTED, "sepFilter: depth pair %d -> %d (channels %d -> %d) outside the GPU path", c.sdepth, c.ddepth, c.cn, MI355CV_MAT_CN(dtype));
    const int nx = (int)kx.size(), ny = (int)ky.size();
    const bool large = nx > 33 || ny > 33;
    if (nx < 1 || ny < 1 || nx > lim::SEP_MAX_TAPS || ny > lim::SEP_MAX_TAPS || (c.wide && (nx > lim::SEP_MAX_TAPS_64F || ny > lim::SEP_MAX_TAPS_64F)))
        return mi355::declined(__func__, __LINE__, "nx < 1 || ny < 1 || nx or ny > lim::SEP_MAX_TAPS (lim::SEP_MAX_TAPS_64F into CV_64F)");
    if (large && c.cn > 4) return mi355::declined(__func__, __LINE__, "more than 33 taps on more than 4 channels (seplong.hip covers 1-4)");
    if (ax < 0) ax = nx / 2;
    if (ay < 0) ay = ny / 2;
    if (ax >= nx || ay >= ny) return mi355::declined(__func__, __LINE__, "ax >= nx || ay >= ny");
    c.border = border & ~MI355CV_BORDER_ISOLATED;
    if (c.border < 0 || c.border > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "c.border < 0 || c.border > B_REFLECT_101");
    c.ax = ax; c.ay = ay; c.lnx = nx; c.lny = ny; c.big = large;
    SepParams& p = c.sp;
    memset(&p, 0, sizeof p);
    const int rtype = kernelType(kx, ax), ctype = kernelType(ky, ay);
    if (c.wide) {
        SepParams64& q = c.sp64;
        memset(&q, 0, sizeof q);
        q.nx = nx; q.ny = ny; q.ax = ax; q.ay = ay; q.delta = delta;
        for (int i = 0; i < nx; i++) q.kx[i] = kx[i];
        for (int i = 0; i < ny; i++) q.ky[i] = ky[i];
        q.symY = (ctype & K_SYMMETRICAL) ? 1 : (ctype & K_ASYMMETRICAL) ? 2 : 0;
        if (!(ny & 1)) q.symY = 0;
        return MI355CV_OK;
    }
    // which engine the reference builds (createSeparableLinearFilter, filter.dispatch.cpp:305-420): bit-exact integer taps for CV_8U smoothing / derivative
    // kernels, float otherwise
    c.lmode = 0; c.ldeltaI = 0;
    c.lkxi.assign(nx, 0); c.lkyi.assign(ny, 0);
    if (c.sdepth == D8U &&
        ((rtype == K_SMOOTH + K_SYMMETRICAL && ctype == K_SMOOTH + K_SYMMETRICAL && c.ddepth == D8U) ||
         ((rtype & (K_SYMMETRICAL + K_ASYMMETRICAL)) && (ctype & (K_SYMMETRICAL + K_ASYMMETRICAL)) && (rtype & ctype & K_INTEGER) && c.ddepth == D16S))) {
        const int bits = c.ddepth == D8U ? 8 : 0;
        std::vector<int> qx, qy;
        if (bitExactKernel(kx, bits, qx) && bitExactKernel(ky, bits, qy)) {
            c.lmode = bits ? 1 : 2;
            c.lkxi = qx; c.lkyi = qy;
            const double d = delta * (double)(1 << (2 * bits));
            c.ldeltaI = d >= 2147483647.0 ? 2147483647 : d <= -2147483648.0 ? (int)-2147483648LL : (int)nearbyint(d);
        }
    }
    c.lkxf.assign(kx.begin(), kx.end());
    c.lkyf.assign(ky.begin(), ky.end());
    c.ldeltaF = (float)delta;
    c.lsymY = (ctype & K_SYMMETRICAL) ? 1 : (ctype & K_ASYMMETRICAL) ? 2 : 0;
    if (!(ny & 1)) c.lsymY = 0;
    if (!large) {
        p.nx = nx; p.ny = ny; p.ax = ax; p.ay = ay;
        p.mode = c.lmode; p.deltaI = c.ldeltaI; p.deltaF = c.ldeltaF; p.symY = c.lsymY;
        for (int i = 0; i < nx; i++) { p.kxf[i] = c.lkxf[i]; p.kxi[i] = c.lkxi[i]; }
        for (int i = 0; i < ny; i++) { p.kyf[i] = c.lkyf[i]; p.kyi[i] = c.lkyi[i]; }
    }
    return MI355CV_OK;
}

// the LDS-ring kernel for everything the rolling kernels do not take (seplong.hip): any tap count, anchor, border, 1-4 channels
bool sepLong(Stager& stg, const FilterCtx& c, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
             int W, int H, int fullW, int fullH, int offX, int offY)
{
    const SepLongTaps t = {c.lkxf.data(), c.lkyf.data(), c.lkxi.data(), c.lkyi.data(), c.lnx, c.lny, c.ax, c.ay, c.lmode, c.lsymY, c.ldeltaF, c.ldeltaI};
    return seplongRun(stg, src, sstep, sframe, dst, dstep, dframe, nframes, W, H, c.cn, c.sdepth, c.ddepth, fullW, fullH, offX, offY, c.border, t, stream());
}

int sepRun(const char* entry, const FilterCtx& c, const uchar* src, size_t sstep, uchar* dst, size_t dstep,
           int W, int H, int fullW, int fullH, int offX, int offY)
{
    if (disabled() || W <= 0 || H <= 0) return mi355::declined(__func__, __LINE__, "disabled() || W <= 0 || H <= 0");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src, (size_t)W * H, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src, (size_t)W * H, minPixels(HOST_HEAVY))");
    const int se = depthSize(c.sdepth), de = depthSize(c.ddepth);
    // stage the whole parent region so that non-isolated borders can read real neighbours
    const uchar* top = src - (ptrdiff_t)offY * (ptrdiff_t)sstep - (ptrdiff_t)offX * c.cn * se;
    if (overlapOnDevice(top, (size_t)(fullH - 1) * sstep + (size_t)fullW * c.cn * se, dst, (size_t)(H - 1) * dstep + (size_t)W * c.cn * de))
        return setError(MI355CV_NOT_IMPLEMENTED, "%s: dst overlaps the device-resident source rows (in-place)", entry);
    size_t dss, dds;
    const uchar* dtop = stg.in(top, sstep, (size_t)fullW * c.cn * se, fullH, &dss);
    uchar* dd = stg.out(dst, dstep, (size_t)W * c.cn * de, H, &dds);
    if (!dtop || !dd) return mi355::declined(__func__, __LINE__, "!dtop || !dd");
    const uchar* ds = dtop + (size_t)offY * dss + (size_t)offX * c.cn * se;
    if (c.wide) {
        hipLaunchKernelGGL(k_sepfilter_generic64, dim3(divUp(W * c.cn, 64), divUp(H, 4)), dim3(256), 0, stream(), ds, dss, dd, dds, W, H, c.cn, c.sdepth,
                           fullW, fullH, offX, offY, c.border, c.sp64);
        noteKernel("k_sepfilter_generic64 (depth %d -> CV_64F, %d x %d taps)", c.sdepth, c.sp64.nx, c.sp64.ny);
        return stg.finish(entry);
    }
    if (c.big) {
        if (!sepLong(stg, c, ds, dss, 0, dd, dds, 0, 1, W, H, fullW, fullH, offX, offY)) return mi355::declined(__func__, __LINE__, "seplongRun refused a kernel beyond 33 taps");
        return stg.finish(entry);
    }
    const SepParams& p = c.sp;
    // a submatrix with real pixels around it stays on the rolling kernels: they run on the parent's geometry and store the window (roll.h Win)
    const Roi roiv = {fullW, fullH, offX, offY};
    const Roi* roi = (fullW != W || fullH != H) ? &roiv : nullptr;
    const bool centred = p.nx == p.ny && p.ax == p.nx / 2 && p.ay == p.ny / 2;
    if (p.mode == 2 && c.ddepth == D16S && centred && p.deltaI == 0 &&
        seprollDeriv16(ds, dss, 0, dd, dds, 0, 1, W, H, c.cn, p.kxi, p.kyi, p.nx, c.border, stream(), roi))
        return stg.finish(entry);
    if (p.mode == 1 && c.sdepth == D8U && c.ddepth == D8U && centred && p.ny > 1 &&
        seprollFix8U(ds, dss, 0, dd, dds, 0, 1, W, H, c.cn, p.kxi, p.kyi, p.nx, p.deltaF, c.border, stream(), roi))
        return stg.finish(entry);
    if (p.mode == 0 && c.sdepth == D8U && (c.ddepth == D32F || c.ddepth == D8U) && centred &&
        seprollFloat(ds, dss, 0, dd, dds, 0, 1, W, H, c.cn, p.kxf, p.kyf, p.nx, p.symY, p.deltaF, c.ddepth == D32F ? 4 : 1, c.border, stream(), roi))
        return stg.finish(entry);
    if (p.mode == 0 && c.sdepth == D32F && c.ddepth == D32F && c.cn == 1 && centred &&
        seprollF32(ds, dss, 0, dd, dds, 0, 1, W, H, p.kxf, p.kyf, p.nx, p.symY, p.deltaF, c.border, stream(), roi))
        return stg.finish(entry);
    if (p.mode == 0 && (c.sdepth == D16U || c.sdepth == D16S) && (c.ddepth == c.sdepth || c.ddepth == D32F) && c.cn == 1 && centred &&
        seprollF16(ds, dss, 0, dd, dds, 0, 1, W, H, c.sdepth == D16S, c.ddepth == D32F, p.kxf, p.kyf, p.nx, p.symY, p.deltaF, c.border, stream(), roi))
        return stg.finish(entry);
    // everything else with 1-4 channels: the LDS-ring kernel, nx + ny multiply-adds per element
    if (std::getenv("MI355CV_SEP_GENERIC") == nullptr && sepLong(stg, c, ds, dss, 0, dd, dds, 0, 1, W, H, fullW, fullH, offX, offY)) return stg.finish(entry);
    // more than 4 channels (at most 33 taps, sepInit): one thread per output element, nx * ny gathers -- the correctness path of cv::sepFilter2D on exotic Mats
    dim3 grid(divUp(W * c.cn, 64), divUp(H, 4));
    hipLaunchKernelGGL((k_sepfilter_generic<33>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, W, H, c.cn, c.sdepth, c.ddepth,
                       fullW, fullH, offX, offY, c.border, c.sp);
    noteKernel("k_sepfilter_generic<33> (%d x %d taps, %d channels)", c.sp.nx, c.sp.ny, c.cn);
    return stg.finish(entry);
}

// the same over a batch of device-resident whole frames (every frame its own image: isolated borders): one launch of the rolling kernels
// where they apply, the generic kernel frame by frame otherwise (all in stream order, one synchronisation at most)
int sepRunBatch(const char* entry, const FilterCtx& c, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes, int W, int H)
{
    if (disabled() || W <= 0 || H <= 0 || nframes < 1) return mi355::declined(__func__, __LINE__, "disabled() || W <= 0 || H <= 0 || nframes < 1");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (!isDevicePtr(src) || !isDevicePtr(dst)) return setError(MI355CV_NOT_IMPLEMENTED, "%s: batch entry needs device-resident frames", entry);
    const int se = depthSize(c.sdepth), de = depthSize(c.ddepth);
    // one check for every path below, the CV_64F and long-kernel ones included (they used to launch before it)
    if (overlapOnDevice(src, (size_t)(nframes - 1) * sframe + (size_t)(H - 1) * sstep + (size_t)W * c.cn * se,
                        dst, (size_t)(nframes - 1) * dframe + (size_t)(H - 1) * dstep + (size_t)W * c.cn * de))
        return setError(MI355CV_NOT_IMPLEMENTED, "%s: dst overlaps the source frames (in-place)", entry);
    if (c.wide) {                                // CV_64F destinations: the generic double kernel, frame by frame
        for (int f = 0; f < nframes; f++)
            hipLaunchKernelGGL(k_sepfilter_generic64, dim3(divUp(W * c.cn, 64), divUp(H, 4)), dim3(256), 0, stream(), src + (size_t)f * sframe, sstep, dst + (size_t)f * dframe, dstep,
                               W, H, c.cn, c.sdepth, W, H, 0, 0, c.border, c.sp64);
        return stg.finish(entry);
    }
    if (c.big) {                                 // 34 .. lim::SEP_MAX_TAPS taps: the LDS-ring kernel, frames along grid z
        if (!sepLong(stg, c, src, sstep, sframe, dst, dstep, dframe, nframes, W, H, W, H, 0, 0)) return mi355::declined(__func__, __LINE__, "seplongRun refused a kernel beyond 33 taps");
        return stg.finish(entry);
    }
    if (nframes == 1) { sframe = 0; dframe = 0; }
    const SepParams& p = c.sp;
    const bool centred = p.nx == p.ny && p.ax == p.nx / 2 && p.ay == p.ny / 2;
    if (p.mode == 2 && c.ddepth == D16S && centred && p.deltaI == 0 &&
        seprollDeriv16(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, c.cn, p.kxi, p.kyi, p.nx, c.border, stream()))
        return stg.finish(entry);
    if (p.mode == 1 && c.sdepth == D8U && c.ddepth == D8U && centred && p.ny > 1 &&
        seprollFix8U(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, c.cn, p.kxi, p.kyi, p.nx, p.deltaF, c.border, stream()))
        return stg.finish(entry);
    if (p.mode == 0 && c.sdepth == D8U && (c.ddepth == D32F || c.ddepth == D8U) && centred &&
        seprollFloat(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, c.cn, p.kxf, p.kyf, p.nx, p.symY, p.deltaF, c.ddepth == D32F ? 4 : 1, c.border, stream()))
        return stg.finish(entry);
    if (p.mode == 0 && c.sdepth == D32F && c.ddepth == D32F && c.cn == 1 && centred &&
        seprollF32(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, p.kxf, p.kyf, p.nx, p.symY, p.deltaF, c.border, stream()))
        return stg.finish(entry);
    if (p.mode == 0 && (c.sdepth == D16U || c.sdepth == D16S) && (c.ddepth == c.sdepth || c.ddepth == D32F) && c.cn == 1 && centred &&
        seprollF16(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, c.sdepth == D16S, c.ddepth == D32F, p.kxf, p.kyf, p.nx, p.symY, p.deltaF, c.border, stream()))
        return stg.finish(entry);
    if (std::getenv("MI355CV_SEP_GENERIC") == nullptr && sepLong(stg, c, src, sstep, sframe, dst, dstep, dframe, nframes, W, H, W, H, 0, 0)) return stg.finish(entry);
    dim3 grid(divUp(W * c.cn, 64), divUp(H, 4));            // more than 4 channels
    for (int f = 0; f < nframes; f++)
        hipLaunchKernelGGL((k_sepfilter_generic<33>), grid, dim3(256), 0, stream(), src + (size_t)f * sframe, sstep, dst + (size_t)f * dframe, dstep, W, H, c.cn, c.sdepth, c.ddepth,
                           W, H, 0, 0, c.border, c.sp);
    return stg.finish(entry);
}

// getSobelKernels / getScharrKernels (deriv.cpp:55-162) as integer taps
bool derivKernel(int order, int ksize, bool scharr, std::vector<int>& k)
{
    if (scharr) {
        if (order == 0) k = {3, 10, 3}; else if (order == 1) k = {-1, 0, 1}; else return false;
        return true;
    }
    if (ksize == 1 && order > 0) ksize = 3;
    if (ksize % 2 == 0 || ksize > 31 || ksize <= order) return false;
    if (ksize == 1) { k = {1}; return true; }
    if (ksize == 3) {
        if (order == 0) k = {1, 2, 1}; else if (order == 1) k = {-1, 0, 1}; else k = {1, -2, 1};
        return true;
    }
    std::vector<int> kerI(ksize + 1, 0);
    kerI[0] = 1;
    for (int i = 0; i < ksize - order - 1; i++) {
        int oldval = kerI[0];
        for (int j = 1; j <= ksize; j++) { int nv = kerI[j] + kerI[j - 1]; kerI[j - 1] = oldval; oldval = nv; }
    }
    for (int i = 0; i < order; i++) {
        int oldval = -kerI[0];
        for (int j = 1; j <= ksize; j++) { int nv = kerI[j - 1] - kerI[j]; kerI[j - 1] = oldval; oldval = nv; }
    }
    k.assign(kerI.begin(), kerI.begin() + ksize);
    return true;
}

int derivRun(const char* entry, const uchar* src, size_t sstep, uchar* dst, size_t dstep, int W, int H, int sdepth, int ddepth,
             int cn, int mL, int mT, int mR, int mB, int dx, int dy, int ksize, bool scharr, double scale, double delta, int border,
             int nframes = 0, size_t sframe = 0, size_t dframe = 0)
{
    if (dx < 0 || dy < 0 || (scharr ? dx + dy != 1 : dx + dy <= 0) || inPlaceOnDevice(src, dst)) return mi355::declined(__func__, __LINE__, "dx < 0 || dy < 0 || (scharr ? dx + dy != 1 : dx + dy <= 0) || inPlaceOnDevice(src, dst)");
    std::vector<int> ix, iy;
    if (!derivKernel(dx, ksize, scharr, ix) || !derivKernel(dy, ksize, scharr, iy)) return mi355::declined(__func__, __LINE__, "!derivKernel(dx, ksize, scharr, ix) || !derivKernel(dy, ksize, scharr, iy)");
    // ktype = max(CV_32F, ddepth, sdepth): `kx *= scale` is evaluated in double and stored back in the kernel's type (deriv.cpp:421, :432-439) -- float unless
    // the source or the destination is CV_64F
    const bool wideK = MI355CV_MAT_DEPTH(sdepth) == D64F || MI355CV_MAT_DEPTH(ddepth) == D64F;
    std::vector<double> kx(ix.begin(), ix.end()), ky(iy.begin(), iy.end());
    if (scale != 1) {
        std::vector<double>& tgt = dx == 0 ? kx : ky;
        for (double& v : tgt) v = wideK ? v * scale : (double)(float)(v * scale);
    }
    FilterCtx c;
    int rc = sepInit(c, MI355CV_MAKETYPE(sdepth, cn), MI355CV_MAKETYPE(ddepth, cn), kx, ky, -1, -1, delta, border);
    if (rc != MI355CV_OK) return rc;
    if (nframes > 0) return sepRunBatch(entry, c, src, sstep, sframe, dst, dstep, dframe, nframes, W, H);
    return sepRun(entry, c, src, sstep, dst, dstep, W, H, mL + W + mR, mT + H + mB, mL, mT);
}

} // namespace

struct cvhalFilter2D;   // opaque to the caller (hal_replacement.hpp:70-87)

extern "C" {

MI355CV_API int mi355cv_filterInit(cvhalFilter2D** context, uchar* kernel_data, size_t kernel_step, int kernel_type,
        int kernel_width, int kernel_height, int max_width, int max_height, int src_type, int dst_type, int borderType,
        double delta, int anchor_x, int anchor_y, bool allowSubmatrix, bool allowInplace)
{
    mi355::EntryGuard entry_(__func__);
    // allowInplace (src_data == dst_data at the call site, filter.dispatch.cpp:1176): a host image goes through separate device buffers, so it is served;
    // a device image filtered in place is refused by mi355cv_filter itself (overlapOnDevice), which the caller treats as "not replaced" (:1177-1183)
    (void)max_width; (void)max_height; (void)allowSubmatrix; (void)allowInplace;
    if (!context || !kernel_data || disabled()) return mi355::declined(__func__, __LINE__, "!context || !kernel_data || disabled()");
    if (MI355CV_MAT_CN(kernel_type) != 1 || kernel_width < 1 || kernel_height < 1 || kernel_width * kernel_height > 1024)
        return mi355::declined(__func__, __LINE__, "MI355CV_MAT_CN(kernel_type) != 1 || kernel_width < 1 || kernel_height < 1 || kernel_width * kernel_height > 1024");
    FilterCtx* c = new (std::nothrow) FilterCtx();
    if (!c) return mi355::declined(__func__, __LINE__, "!c");
    c->kind = 1;
    c->sdepth = MI355CV_MAT_DEPTH(src_type); c->ddepth = MI355CV_MAT_DEPTH(dst_type); c->cn = MI355CV_MAT_CN(src_type);
    c->border = borderType & ~MI355CV_BORDER_ISOLATED;
    c->kw = kernel_width; c->kh = kernel_height;
    c->ax = anchor_x < 0 ? kernel_width / 2 : anchor_x; c->ay = anchor_y < 0 ? kernel_height / 2 : anchor_y;
    c->delta = (float)delta;                                         // saturate_cast<float>(delta), filter.simd.hpp:3113
    c->wide = widePairOk(c->sdepth, c->ddepth, false); c->delta64 = delta;
    if (c->cn != MI355CV_MAT_CN(dst_type) || !(c->wide || depthPairOk(c->sdepth, c->ddepth)) || c->border < 0 || c->border > B_REFLECT_101 ||
        c->ax >= kernel_width || c->ay >= kernel_height) {
        const int sd = c->sdepth, dd = c->ddepth, bd = c->border; delete c;
        return setError(MI355CV_NOT_IMPLEMENTED, "filter2D: depth pair %d -> %d, border %d, anchor (%d, %d) in %d x %d outside the GPU path", sd, dd, bd, anchor_x, anchor_y, kernel_width, kernel_height);
    }
    for (int i = 0; i < kernel_height; i++)
        for (int j = 0; j < kernel_width; j++) {
            const float v = (float)kernelAt(kernel_data, kernel_step, kernel_type, i, j);   // convertTo(CV_32F), :3201-3205
            if (v == 0) continue;
            c->taps.push_back({v, j, i});
        }
    if (c->taps.empty()) c->taps.push_back({0.f, 0, 0});              // nz == 0 -> one zero tap (:393-395)
    if (c->wide) {                                                   // the kernel converted to CV_64F instead (kdepth = CV_64F, :3201-3205)
        for (int i = 0; i < kernel_height; i++)
            for (int j = 0; j < kernel_width; j++) {
                const double v = kernelAt(kernel_data, kernel_step, kernel_type, i, j);
                if (v != 0) c->taps64.push_back({v, j, i});
            }
        if (c->taps64.empty()) c->taps64.push_back({0.0, 0, 0});
    }
    *context = reinterpret_cast<cvhalFilter2D*>(c);
    return MI355CV_OK;
}

// launches the rolling kernel when the geometry allows it; returns false otherwise
static bool tryFilterRoll(const FilterCtx* c, const uchar* ds, size_t dss, size_t sframe, uchar* dd, size_t dds, size_t dframe, int nframes, int W, int H, hipStream_t st)
{
    const int K = c->kw;
    if (c->sdepth == D32F && c->ddepth == D32F && c->cn == 1 && c->kw == c->kh && (K == 3 || K == 5) && c->ax == K / 2 && c->ay == K / 2 &&
        (((uintptr_t)ds | dss | sframe | (uintptr_t)dd | dds | dframe) & 3) == 0 && roll::eligible(ds, dss, sframe, dd, dds, dframe, W, 4, K / 2, c->border)) {
        DenseTaps t; memset(&t, 0, sizeof t);
        for (const Tap2D& tp : c->taps) t.k[tp.dy * K + tp.dx] = tp.k;
        t.delta = c->delta;
        const roll::Geom g = roll::geometry(W, H, 4, nframes, K == 3 ? 16 : 12, K);
        if (K == 3) hipLaunchKernelGGL((k_filter2d_roll_f32<3>), dim3(g.blocks), dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W, H, g.nchunks, g.nstrips, g.seg, g.nseg, nframes, c->border, roll::wholeImage(), t);
        else        hipLaunchKernelGGL((k_filter2d_roll_f32<5>), dim3(g.blocks), dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W, H, g.nchunks, g.nstrips, g.seg, g.nseg, nframes, c->border, roll::wholeImage(), t);
        noteKernel("k_filter2d_roll_f32<%d> blocks=%u seg=%d rows", K, g.blocks, g.seg);
        return true;
    }
    if (c->sdepth != D8U || c->ddepth != D8U || c->kw != c->kh || (K != 3 && K != 5) || c->ax != K / 2 || c->ay != K / 2) return false;
    if (!((K == 3 && (c->cn == 1 || c->cn == 3 || c->cn == 4)) || (K == 5 && c->cn == 1))) return false;
    if (!roll::eligible(ds, dss, sframe, dd, dds, dframe, W, c->cn, K / 2, c->border)) return false;
    DenseTaps t; memset(&t, 0, sizeof t);
    for (const Tap2D& tp : c->taps) t.k[tp.dy * K + tp.dx] = tp.k;
    t.delta = c->delta;
    // 5x5: 32-row segments (a 4-row halo re-read per 32 rows instead of per 12; 64 x 4K sweep, profiles/r03_roll_seg_sweep.txt: 0.356 -> 0.408 of HBM)
    const roll::Geom g = roll::geometry(W, H, c->cn, nframes, K == 3 ? 16 : 32, K);
#define FROLL(K_, CN_) hipLaunchKernelGGL((k_filter2d_roll<K_, CN_>), dim3(g.blocks), dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W, H, g.nchunks, g.nstrips, g.seg, g.nseg, nframes, c->border, 1, t)
    if (K == 3) { if (c->cn == 1) FROLL(3, 1); else if (c->cn == 3) FROLL(3, 3); else FROLL(3, 4); }
    else FROLL(5, 1);
#undef FROLL
    return true;
}

// the LDS-tile kernel for what the rolling kernels do not take: any anchor / depth pair / channel count / ROI window, kernels of 9 .. 1024 taps up to 32 wide
static bool tryFilterTile(Stager& stg, const FilterCtx* c, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                          int W, int H, int fullW, int fullH, int offX, int offY, hipStream_t st)
{
    static const bool off = [] { const char* v = getenv("MI355CV_FILTER_TILE"); return v && atoi(v) == 0; }();
    const int nc = (c->kw + 3) / 4;
    if (off || c->wide || c->kw * c->kh < 9 || nc > 8 || (long long)nframes * c->cn > 65535 || divUp(H, FT_H) > 65535) return false;
    bool any = false;
    for (const Tap2D& tp : c->taps) any = any || tp.k != 0.f;
    if (!any) return false;                                          // an all-zero kernel is the generic kernel's single zero tap (filter.simd.hpp:393-395)
    std::vector<float> kd((size_t)c->kh * 4 * nc, 0.f);
    std::vector<unsigned> km((size_t)c->kh, 0u);
    for (const Tap2D& tp : c->taps) { kd[(size_t)tp.dy * 4 * nc + tp.dx] = tp.k; km[tp.dy] |= 1u << tp.dx; }
    const float* dk = (const float*)stg.param(kd.data(), kd.size() * sizeof(float));
    const unsigned* dm = (const unsigned*)stg.param(km.data(), km.size() * sizeof(unsigned));
    if (!dk || !dm) return false;
    TileArgs a;
    a.W = W; a.H = H; a.cn = c->cn; a.ddepth = c->ddepth; a.fullW = fullW; a.fullH = fullH; a.offX = offX; a.offY = offY;
    a.kw = c->kw; a.kh = c->kh; a.ax = c->ax; a.ay = c->ay; a.border = c->border; a.ncols = FT_W + 4 * nc; a.pitch = a.ncols; a.delta = c->delta;
    const size_t lds = ((size_t)(FT_HH + c->kh - 1) * a.pitch * 2 + (size_t)c->kh * 4 * nc) * sizeof(float);
    if (lds > 64 * 1024) return false;
    const dim3 grid(divUp(W, FT_W), divUp(H, FT_H), nframes * c->cn);
    switch (c->sdepth) {
    case D8U:  launchFilterTile<uchar>(nc, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, a, dk, dm); break;
    case D16U: launchFilterTile<unsigned short>(nc, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, a, dk, dm); break;
    case D16S: launchFilterTile<short>(nc, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, a, dk, dm); break;
    case D32F: launchFilterTile<float>(nc, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, a, dk, dm); break;
    default: return false;
    }
    noteKernel("k_filter2d_tile<%d> %dx%d taps (%zu non-zero), depth %d -> %d, %d channel(s), lds %zu", nc, c->kw, c->kh, c->taps.size(), c->sdepth, c->ddepth, c->cn, lds);
    return true;
}

// batched filter2D over device-resident frames with a context from mi355cv_filterInit (frames are whole images:
// isolated borders)
MI355CV_API int mi355cv_filterBatch(cvhalFilter2D* context, const uchar* src_data, size_t src_step, size_t src_frame_stride,
        uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes, int width, int height)
{
    mi355::EntryGuard entry_(__func__);
    FilterCtx* c = reinterpret_cast<FilterCtx*>(context);
    if (!c || c->kind != 1 || width <= 0 || height <= 0 || nframes <= 0 || disabled()) return mi355::declined(__func__, __LINE__, "!c || c->kind != 1 || width <= 0 || height <= 0 || nframes <= 0 || disabled()");
    if (c->wide) return setError(MI355CV_NOT_IMPLEMENTED, "filterBatch: CV_64F destinations go frame by frame through mi355cv_filter");
    if (width > 0 && height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {            // frames in host memory: chunks through two sets of device buffers
        const HostBatch hb = {src_data, src_step, src_frame_stride, (size_t)width * c->cn * depthBytes(c->sdepth), height, dst_data, dst_step, dst_frame_stride, (size_t)width * c->cn * depthBytes(c->ddepth), height, nframes};
        return runHostBatch("filterBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_filterBatch(context, s, ss, sf, d, ds, df, nf, width, height); });
    }
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || !isDevicePtr(src_data) || !isDevicePtr(dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || !isDevicePtr(src_data) || !isDevicePtr(dst_data)");
    if (nframes == 1) { src_frame_stride = 0; dst_frame_stride = 0; }
    if (!tryFilterRoll(c, src_data, src_step, src_frame_stride, dst_data, dst_step, dst_frame_stride, nframes, width, height, stream()) &&
        !tryFilterTile(stg, c, src_data, src_step, src_frame_stride, dst_data, dst_step, dst_frame_stride, nframes, width, height, width, height, 0, 0, stream())) {
        Tap2D* dt = (Tap2D*)stg.param(c->taps.data(), c->taps.size() * sizeof(Tap2D));
        if (!dt) return mi355::declined(__func__, __LINE__, "!dt");
        const int se = depthSize(c->sdepth), de = depthSize(c->ddepth); (void)se; (void)de;
        for (int f = 0; f < nframes; f++) {
            dim3 grid(divUp(width * c->cn, 64), divUp(height, 4));
            hipLaunchKernelGGL(k_filter2d_generic, grid, dim3(256), 0, stream(), src_data + (size_t)f * src_frame_stride, src_step,
                               dst_data + (size_t)f * dst_frame_stride, dst_step, width, height, c->cn, c->sdepth, c->ddepth,
                               width, height, 0, 0, dt, (int)c->taps.size(), c->ax, c->ay, c->kw, c->kh, c->delta, c->border);
        }
    }
    return stg.finish("filterBatch");
}

// cv::cvtColor(src, gray, COLOR_BGR2GRAY / RGB2GRAY / BGRA2GRAY / RGBA2GRAY) + cv::filter2D(gray, dst, ...) on device-resident CV_8UC3 / CV_8UC4
// frames in ONE pass (33.2 MB per 4K frame instead of 49.8 MB): `context` comes from mi355cv_filterInit for CV_8UC1 -> CV_8UC1; the
// intermediate gray image is never written.  Served for what the rolling filter kernel serves (3x3 / 5x5, centred anchor, rows of at least 16
// pixels); NOT_IMPLEMENTED otherwise -- the caller then makes the two calls.  Results equal the two-call sequence bit for bit.
MI355CV_API int mi355cv_cvtBGRtoGrayFilterBatch(cvhalFilter2D* context, const uchar* src_data, size_t src_step, size_t src_frame_stride,
        uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes, int width, int height, int scn, bool swapBlue)
{
    mi355::EntryGuard entry_(__func__);
    FilterCtx* c = reinterpret_cast<FilterCtx*>(context);
    if (!c || c->kind != 1 || width <= 0 || height <= 0 || nframes <= 0 || disabled() || (scn != 3 && scn != 4)) return mi355::declined(__func__, __LINE__, "!c || c->kind != 1 || width <= 0 || height <= 0 || nframes <= 0 || disabled() || (scn != 3 && scn != 4)");
    if (width > 0 && height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {            // frames in host memory: chunks through two sets of device buffers
        const HostBatch hb = {src_data, src_step, src_frame_stride, (size_t)width * scn, height, dst_data, dst_step, dst_frame_stride, (size_t)width, height, nframes};
        return runHostBatch("cvtBGRtoGrayFilterBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_cvtBGRtoGrayFilterBatch(context, s, ss, sf, d, ds, df, nf, width, height, scn, swapBlue); });
    }
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || !isDevicePtr(src_data) || !isDevicePtr(dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || !isDevicePtr(src_data) || !isDevicePtr(dst_data)");
    const int K = c->kw;
    if (c->cn != 1 || c->sdepth != D8U || c->ddepth != D8U || c->kw != c->kh || (K != 3 && K != 5) || c->ax != K / 2 || c->ay != K / 2)
        return setError(MI355CV_NOT_IMPLEMENTED, "cvtBGRtoGrayFilterBatch: needs a centred 3x3 / 5x5 CV_8UC1 filter context");
    if (nframes == 1) { src_frame_stride = 0; dst_frame_stride = 0; }
    if (!roll::eligible(dst_data, dst_step, dst_frame_stride, dst_data, dst_step, dst_frame_stride, width, 1, K / 2, c->border) || (width & 15))
        return setError(MI355CV_NOT_IMPLEMENTED, "cvtBGRtoGrayFilterBatch: width must be a multiple of 16 pixels");
    DenseTaps t; memset(&t, 0, sizeof t);
    for (const Tap2D& tp : c->taps) t.k[tp.dy * K + tp.dx] = tp.k;
    t.delta = c->delta;
    const roll::Geom g = roll::geometry(width, height, 1, nframes, K == 3 ? 16 : 12, K);
    const int k0 = swapBlue ? 9798 : 3735, k1 = 19235, k2 = swapBlue ? 3735 : 9798;      // color.simd_helpers.hpp:16-24 ({B2Y, G2Y, R2Y} in channel order)
#define GROLL(K_, S_) hipLaunchKernelGGL((k_gray_filter2d_roll<K_, S_>), dim3(g.blocks), dim3(256), 0, stream(), src_data, src_step, src_frame_stride, dst_data, dst_step, \
                                         dst_frame_stride, width, height, g.nchunks, g.nstrips, g.seg, g.nseg, nframes, c->border, 1, t, k0, k1, k2)
    if (K == 3) { if (scn == 3) GROLL(3, 3); else GROLL(3, 4); }
    else        { if (scn == 3) GROLL(5, 3); else GROLL(5, 4); }
#undef GROLL
    return stg.finish("cvtBGRtoGrayFilterBatch");
}

MI355CV_API int mi355cv_filter(cvhalFilter2D* context, uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step,
        int width, int height, int full_width, int full_height, int offset_x, int offset_y)
{
    mi355::EntryGuard entry_(__func__);
    FilterCtx* c = reinterpret_cast<FilterCtx*>(context);
    if (!c || c->kind != 1 || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "!c || c->kind != 1 || width <= 0 || height <= 0");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
    const int se = depthSize(c->sdepth), de = depthSize(c->ddepth);
    // hal::filter2D tries the hook before its own DFT path (filter.dispatch.cpp:1436-1470): for a whole image and a kernel of >= 130 taps
    // (8U -> 8U / 16S, 32F -> 32F; >= 50 otherwise) the CPU result comes from float FFTs (dftFilter2D :1274-1340), which a direct sum does
    // not reproduce bit for bit -- leave those to the CPU
    {
        // MI355CV_FILTER_LARGE=1 (opt-in, round 5): serve them with the direct sum -- the arithmetic of the reference's OWN non-DFT engine (the one it runs for a submatrix or
        // a smaller kernel: float multiply-add chain over the taps in raster order, one rounding to the destination depth), i.e. the exact correlation rounded once, where the
        // DFT path carries the FFTs' float error (CV_8U results differ from it by at most 1 in isolated pixels, CV_32F by ~1e-6 relative: tests/test_filters_gpu.py reports both)
        // CV_32F destinations are served by default since round 6: the direct sum is within ~1e-6 relative of the DFT result (the float bar of the path is 1e-4; the same test
        // measures it), and the LDS-tile kernel runs 21 x 21 taps on a 4K float frame in ~0.1 ms where the reference spends tens of ms in its FFTs.  Integer destinations stay
        // declined: the bar there is bit for bit, and the DFT result differs from the exact sum by 1 in isolated pixels.
        static const bool serveLarge = [] { const char* v = getenv("MI355CV_FILTER_LARGE"); return v && atoi(v) != 0; }();
        static const bool floatLargeOff = [] { const char* v = getenv("MI355CV_FILTER_LARGE"); return v && atoi(v) == 0 && v[0] == '0'; }();
        const bool fastTypes = (c->sdepth == D8U && (c->ddepth == D8U || c->ddepth == D16S)) || (c->sdepth == D32F && c->ddepth == D32F);
        const bool served = serveLarge || (c->ddepth == D32F && !floatLargeOff);
        if (!served && c->kw * c->kh >= (fastTypes ? lim::FILTER2D_DFT_TAPS : 50) && offset_x == 0 && offset_y == 0 && width == full_width && height == full_height)
            return setError(MI355CV_NOT_IMPLEMENTED, "filter: %dx%d kernel on a whole image is the reference's DFT case", c->kw, c->kh);
    }
    const uchar* top = src_data - (ptrdiff_t)offset_y * (ptrdiff_t)src_step - (ptrdiff_t)offset_x * c->cn * se;
    if (overlapOnDevice(top, (size_t)(full_height - 1) * src_step + (size_t)full_width * c->cn * se, dst_data, (size_t)(height - 1) * dst_step + (size_t)width * c->cn * de))
        return setError(MI355CV_NOT_IMPLEMENTED, "filter: dst overlaps the device-resident source rows (in-place)");
    size_t dss, dds;
    const uchar* dtop = stg.in(top, src_step, (size_t)full_width * c->cn * se, full_height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * c->cn * de, height, &dds);
    Tap2D* dt = (Tap2D*)stg.param(c->taps.data(), c->taps.size() * sizeof(Tap2D));
    if (!dtop || !dd || !dt) return mi355::declined(__func__, __LINE__, "!dtop || !dd || !dt");
    const uchar* ds = dtop + (size_t)offset_y * dss + (size_t)offset_x * c->cn * se;
    if (c->wide) {
        Tap2D64* dt64 = (Tap2D64*)stg.param(c->taps64.data(), c->taps64.size() * sizeof(Tap2D64));
        if (!dt64) return mi355::declined(__func__, __LINE__, "!dt64");
        hipLaunchKernelGGL(k_filter2d_generic64, dim3(divUp(width * c->cn, 64), divUp(height, 4)), dim3(256), 0, stream(), ds, dss, dd, dds, width, height, c->cn, c->sdepth,
                           full_width, full_height, offset_x, offset_y, dt64, (int)c->taps64.size(), c->ax, c->ay, c->delta64, c->border);
        noteKernel("k_filter2d_generic64 (depth %d -> CV_64F, %zu taps)", c->sdepth, c->taps64.size());
        return stg.finish("filter");
    }
    if (full_width == width && full_height == height && tryFilterRoll(c, ds, dss, 0, dd, dds, 0, 1, width, height, stream()))
        return stg.finish("filter");
    if (tryFilterTile(stg, c, ds, dss, 0, dd, dds, 0, 1, width, height, full_width, full_height, offset_x, offset_y, stream()))
        return stg.finish("filter");
    dim3 grid(divUp(width * c->cn, 64), divUp(height, 4));
    hipLaunchKernelGGL(k_filter2d_generic, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, c->cn, c->sdepth, c->ddepth,
                       full_width, full_height, offset_x, offset_y, dt, (int)c->taps.size(), c->ax, c->ay, c->kw, c->kh, c->delta, c->border);
    noteKernel("k_filter2d_generic %dx%d taps, depth %d -> %d, %d channel(s)", c->kw, c->kh, c->sdepth, c->ddepth, c->cn);
    return stg.finish("filter");
}

MI355CV_API int mi355cv_filterFree(cvhalFilter2D* context)
{
    mi355::EntryGuard entry_(__func__);
    delete reinterpret_cast<FilterCtx*>(context);
    return MI355CV_OK;
}

MI355CV_API int mi355cv_sepFilterInit(cvhalFilter2D** context, int src_type, int dst_type, int kernel_type,
        uchar* kernelx_data, int kernelx_length, uchar* kernely_data, int kernely_length,
        int anchor_x, int anchor_y, double delta, int borderType)
{
    mi355::EntryGuard entry_(__func__);
    if (!context || !kernelx_data || !kernely_data || disabled()) return mi355::declined(__func__, __LINE__, "!context || !kernelx_data || !kernely_data || disabled()");
    if (MI355CV_MAT_CN(kernel_type) != 1 || kernelx_length < 1 || kernely_length < 1) return mi355::declined(__func__, __LINE__, "MI355CV_MAT_CN(kernel_type) != 1 || kernelx_length < 1 || kernely_length < 1");
    std::vector<double> kx(kernelx_length), ky(kernely_length);
    for (int i = 0; i < kernelx_length; i++) kx[i] = kernelAt(kernelx_data, 0, kernel_type, 0, i);
    for (int i = 0; i < kernely_length; i++) ky[i] = kernelAt(kernely_data, 0, kernel_type, 0, i);
    FilterCtx* c = new (std::nothrow) FilterCtx();
    if (!c) return mi355::declined(__func__, __LINE__, "!c");
    int rc = sepInit(*c, src_type, dst_type, kx, ky, anchor_x, anchor_y, delta, borderType);
    if (rc != MI355CV_OK) { delete c; return rc; }
    *context = reinterpret_cast<cvhalFilter2D*>(c);
    return MI355CV_OK;
}

MI355CV_API int mi355cv_sepFilter(cvhalFilter2D* context, uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step,
        int width, int height, int full_width, int full_height, int offset_x, int offset_y)
{
    mi355::EntryGuard entry_(__func__);
    FilterCtx* c = reinterpret_cast<FilterCtx*>(context);
    if (!c || c->kind != 2) return mi355::declined(__func__, __LINE__, "!c || c->kind != 2");
    return sepRun("sepFilter", *c, src_data, src_step, dst_data, dst_step, width, height, full_width, full_height, offset_x, offset_y);
}

MI355CV_API int mi355cv_sepFilterFree(cvhalFilter2D* context)
{
    mi355::EntryGuard entry_(__func__);
    delete reinterpret_cast<FilterCtx*>(context);
    return MI355CV_OK;
}

// what mi355cv_sepFilterInit decided, for tests that replay the LDS-ring kernel on the CPU with the product's own host decisions (tests/test_hostemu.py): info = {mode, symY,
// nx, ny, ax, ay, deltaI, wide}; kx / ky (room for lim::SEP_MAX_TAPS each) receive the taps as the kernel reads them -- float bits (mode 0) or int32; needs no device
MI355CV_API int mi355cv_sepFilterDescribe(cvhalFilter2D* context, int* info, float* deltaF, unsigned* kx, unsigned* ky)
{
    const FilterCtx* c = reinterpret_cast<const FilterCtx*>(context);
    if (!c || c->kind != 2 || !info || !deltaF || !kx || !ky) return MI355CV_ERROR_UNKNOWN;
    info[0] = c->lmode; info[1] = c->lsymY; info[2] = c->lnx; info[3] = c->lny; info[4] = c->ax; info[5] = c->ay; info[6] = c->ldeltaI; info[7] = c->wide;
    *deltaF = c->ldeltaF;
    if (c->wide) return MI355CV_OK;
    for (int i = 0; i < c->lnx; i++) { if (c->lmode == 0) memcpy(&kx[i], &c->lkxf[i], 4); else kx[i] = (unsigned)c->lkxi[i]; }
    for (int i = 0; i < c->lny; i++) { if (c->lmode == 0) memcpy(&ky[i], &c->lkyf[i], 4); else ky[i] = (unsigned)c->lkyi[i]; }
    return MI355CV_OK;
}

MI355CV_API int mi355cv_sobel(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
        int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right, int margin_bottom,
        int dx, int dy, int ksize, double scale, double delta, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    const bool scharr = ksize <= 0;                                   // FILTER_SCHARR == -1 (getDerivKernels, deriv.cpp:165-171)
    // (depth arguments arrive as cv::Sobel's caller gave them: a type there carries channel bits, deriv.cpp:425-456; the destination was created from the depth bits)
    return derivRun("sobel", src_data, src_step, dst_data, dst_step, width, height, MI355CV_MAT_DEPTH(src_depth), MI355CV_MAT_DEPTH(dst_depth), cn,
                    margin_left, margin_top, margin_right, margin_bottom, dx, dy, ksize, scharr, scale, delta, border_type);
}

MI355CV_API int mi355cv_scharr(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
        int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right, int margin_bottom,
        int dx, int dy, double scale, double delta, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    return derivRun("scharr", src_data, src_step, dst_data, dst_step, width, height, MI355CV_MAT_DEPTH(src_depth), MI355CV_MAT_DEPTH(dst_depth), cn,
                    margin_left, margin_top, margin_right, margin_bottom, dx, dy, 0, true, scale, delta, border_type);
}

// ---- batches of device-resident whole frames (frame strides in bytes; borders are per frame, i.e. isolated)
MI355CV_API int mi355cv_sobelBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride, uchar* dst_data, size_t dst_step, size_t dst_frame_stride,
        int nframes, int width, int height, int src_depth, int dst_depth, int cn, int dx, int dy, int ksize, double scale, double delta, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (nframes < 1) return mi355::declined(__func__, __LINE__, "nframes < 1");
    if (width > 0 && height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {            // frames in host memory: chunks through two sets of device buffers
        const HostBatch hb = {src_data, src_step, src_frame_stride, (size_t)width * cn * depthBytes(src_depth), height, dst_data, dst_step, dst_frame_stride, (size_t)width * cn * depthBytes(dst_depth), height, nframes};
        return runHostBatch("sobelBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_sobelBatch(s, ss, sf, d, ds, df, nf, width, height, src_depth, dst_depth, cn, dx, dy, ksize, scale, delta, border_type); });
    }
    return derivRun("sobelBatch", src_data, src_step, dst_data, dst_step, width, height, src_depth, dst_depth, cn, 0, 0, 0, 0, dx, dy, ksize, ksize <= 0, scale, delta,
                    border_type & ~MI355CV_BORDER_ISOLATED, nframes, src_frame_stride, dst_frame_stride);
}

MI355CV_API int mi355cv_sepFilterBatch(cvhalFilter2D* context, const uchar* src_data, size_t src_step, size_t src_frame_stride, uchar* dst_data, size_t dst_step,
        size_t dst_frame_stride, int nframes, int width, int height)
{
    mi355::EntryGuard entry_(__func__);
    FilterCtx* c = reinterpret_cast<FilterCtx*>(context);
    if (!c || c->kind != 2 || nframes < 1) return mi355::declined(__func__, __LINE__, "!c || c->kind != 2 || nframes < 1");
    if (width > 0 && height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {            // frames in host memory: chunks through two sets of device buffers
        const HostBatch hb = {src_data, src_step, src_frame_stride, (size_t)width * c->cn * depthBytes(c->sdepth), height, dst_data, dst_step, dst_frame_stride, (size_t)width * c->cn * depthBytes(c->ddepth), height, nframes};
        return runHostBatch("sepFilterBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_sepFilterBatch(context, s, ss, sf, d, ds, df, nf, width, height); });
    }
    return sepRunBatch("sepFilterBatch", *c, src_data, src_step, src_frame_stride, dst_data, dst_step, dst_frame_stride, nframes, width, height);
}

static int boxRun(const char* entry, const uchar* src_data, size_t src_step, size_t sframe, uchar* dst_data, size_t dst_step, size_t dframe, int nframes, int width, int height,
        int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right, int margin_bottom,
        size_t ksize_width, size_t ksize_height, int anchor_x, int anchor_y, bool normalize, int border_type);

MI355CV_API int mi355cv_boxFilterBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride, uchar* dst_data, size_t dst_step, size_t dst_frame_stride,
        int nframes, int width, int height, int src_depth, int dst_depth, int cn, size_t ksize_width, size_t ksize_height, int anchor_x, int anchor_y, bool normalize,
        int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (nframes < 1) return mi355::declined(__func__, __LINE__, "nframes < 1");
    if (width > 0 && height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {            // frames in host memory: chunks through two sets of device buffers
        const HostBatch hb = {src_data, src_step, src_frame_stride, (size_t)width * cn * depthBytes(src_depth), height, dst_data, dst_step, dst_frame_stride, (size_t)width * cn * depthBytes(dst_depth), height, nframes};
        return runHostBatch("boxFilterBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_boxFilterBatch(s, ss, sf, d, ds, df, nf, width, height, src_depth, dst_depth, cn, ksize_width, ksize_height, anchor_x, anchor_y, normalize, border_type); });
    }
    return boxRun("boxFilterBatch", src_data, src_step, src_frame_stride, dst_data, dst_step, dst_frame_stride, nframes, width, height, src_depth, dst_depth, cn,
                  0, 0, 0, 0, ksize_width, ksize_height, anchor_x, anchor_y, normalize, border_type);
}

MI355CV_API int mi355cv_boxFilter(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
        int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right, int margin_bottom,
        size_t ksize_width, size_t ksize_height, int anchor_x, int anchor_y, bool normalize, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    return boxRun("boxFilter", src_data, src_step, 0, dst_data, dst_step, 0, 0, width, height, src_depth, dst_depth, cn, margin_left, margin_top, margin_right, margin_bottom,
                  ksize_width, ksize_height, anchor_x, anchor_y, normalize, border_type);
}

static bool boxTwoPass(Stager& stg, const BoxParams& p, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes, int W, int H, int cn,
                       int sdepth, int ddepth, int fullW, int fullH, int offX, int offY, int border)
{
    static const bool off = [] { const char* v = getenv("MI355CV_BOX_TWOPASS"); return v && atoi(v) == 0; }();
    if (off || p.kw * p.kh < 16 || p.kw > 1024 || cn > 64 || sdepth == D64F || divUp(H, 32) > 65535) return false;
    hipStream_t st = stream();
    int r0, r1;
    boxRowRange(H, p.kh, fullH, offY, border, &r0, &r1);
    if (r1 - r0 < 1 || r1 - r0 > 65535 || (sdepth != D8U && sdepth != D16U && sdepth != D16S && sdepth != D32F)) return false;
    // ONE scratch image of row sums (int or double) for the whole batch: the frames go through it one after the other, in stream order
    void* R = stg.scratch((((size_t)W * cn + 3) & ~(size_t)3) * (size_t)(r1 - r0) * (sdepth == D32F ? sizeof(double) : sizeof(int)));
    if (!R) return false;
    for (int f = 0; f < nframes; f++) {
        const uchar* s = src + (size_t)f * sframe; uchar* d = dst + (size_t)f * dframe;
        switch (sdepth) {
        case D8U:  boxTwoPassT<uchar, int>(R, p, s, sstep, d, dstep, W, H, cn, ddepth, fullW, fullH, offX, offY, border, st); break;
        case D16U: boxTwoPassT<unsigned short, int>(R, p, s, sstep, d, dstep, W, H, cn, ddepth, fullW, fullH, offX, offY, border, st); break;
        case D16S: boxTwoPassT<short, int>(R, p, s, sstep, d, dstep, W, H, cn, ddepth, fullW, fullH, offX, offY, border, st); break;
        default:   boxTwoPassT<float, double>(R, p, s, sstep, d, dstep, W, H, cn, ddepth, fullW, fullH, offX, offY, border, st); break;
        }
    }
    noteKernel("k_box_rows + k_box_cols %dx%d window, depth %d -> %d, %d channel(s), %d frame(s)", p.kw, p.kh, sdepth, ddepth, cn, nframes);
    return true;
}

// nframes == 0: the hook (one image, margins, host or device); nframes >= 1: a batch of device-resident whole frames
// CV_8U -> CV_8U windows the rolling kernels do not take (9 .. 129 per axis, any anchor, 2 channels, ROI windows): the window sum is two products with banded matrices of
// ones -- k_sepmx (sepmx.hip) computes it exactly on the matrix cores, nx + ny "taps" per byte instead of the kw * kh loads of k_box_generic, and finishes with the
// reference's own normalisation.
static bool boxOnMatrixCores(Stager& stg, const BoxParams& p, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes, int W, int H, int cn,
                             int sdepth, int ddepth, int fullW, int fullH, int offX, int offY, int border)
{
    if (sdepth != D8U || ddepth != D8U || cn > 4 || p.mode == 2 || p.kw > lim::BOX_MAX_KSIZE || p.kh > lim::BOX_MAX_KSIZE) return false;      // (the kernel declines rows of taps beyond its K steps: 255 for one channel)
    if (std::getenv("MI355CV_SEPMX") && std::getenv("MI355CV_SEPMX")[0] == '0') return false;
    uint16_t ones[lim::BOX_MAX_KSIZE];
    for (int i = 0; i < lim::BOX_MAX_KSIZE; i++) ones[i] = 1;
    const SepmxBox b = {!p.normalize ? 3 : p.mode == 0 ? 1 : 2, p.divScale, p.divDelta, p.scaleF, p.scaleD};
    return sepmxRun(stg, src, sstep, sframe, dst, dstep, dframe, nframes, W, H, cn, fullW, fullH, offX, offY, border, ones, p.kw, p.ax, ones, p.kh, p.ay, stream(), &b);
}

static int boxRun(const char* entry, const uchar* src_data, size_t src_step, size_t sframe, uchar* dst_data, size_t dst_step, size_t dframe, int nframes, int width, int height,
        int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right, int margin_bottom,
        size_t ksize_width, size_t ksize_height, int anchor_x, int anchor_y, bool normalize, int border_type)
{
    if (disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 512 || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 512 || inPlaceOnDevice(src_data, dst_data)");
    // cv::boxFilter hands its `ddepth` argument through as the caller gave it (box_filter.dispatch.cpp:451-474): a caller that passed a TYPE there (CV_8UC2 = 8 --
    // the reference's own Imgproc_Blur test does) arrives with channel bits set; the destination Mat was created from CV_MAKETYPE(ddepth, cn), i.e. from the depth bits
    src_depth = MI355CV_MAT_DEPTH(src_depth); dst_depth = MI355CV_MAT_DEPTH(dst_depth);
    const int kw = (int)ksize_width, kh = (int)ksize_height;
    if (kw < 1 || kh < 1 || kw > lim::BOX_MAX_KSIZE || kh > lim::BOX_MAX_KSIZE) return mi355::declined(__func__, __LINE__, "kw < 1 || kh < 1 || kw or kh > lim::BOX_MAX_KSIZE");
    const int border = border_type & ~MI355CV_BORDER_ISOLATED;
    if (border < 0 || border > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "border < 0 || border > B_REFLECT_101");
    // integer sources: int sums into any of the destination depths the reference has a ColumnSum<int, T> for (8U -> 16S is what an un-normalised cv::boxFilter of
    // bytes usually asks for); a signed source into an unsigned destination is left to the caller's path
    const bool okDepth = (src_depth == D8U && (dst_depth == D8U || dst_depth == D16U || dst_depth == D16S || dst_depth == D32F)) ||
                         (src_depth == D16U && (dst_depth == D8U || dst_depth == D16U || dst_depth == D16S || dst_depth == D32F)) ||
                         (src_depth == D16S && (dst_depth == D16S || dst_depth == D32F)) ||
                         (src_depth == D32F && (dst_depth == D32F || dst_depth == D64F)) || (src_depth == D64F && dst_depth == D64F) ||
                         ((src_depth == D8U || src_depth == D16U || src_depth == D16S) && dst_depth == D64F);
    if (!okDepth) return setError(MI355CV_NOT_IMPLEMENTED, "boxFilter: depth pair %d -> %d outside the GPU path", src_depth, dst_depth);
    BoxParams p; memset(&p, 0, sizeof p);
    p.kw = kw; p.kh = kh;
    p.ax = anchor_x < 0 ? kw / 2 : anchor_x; p.ay = anchor_y < 0 ? kh / 2 : anchor_y;
    if (p.ax >= kw || p.ay >= kh) return mi355::declined(__func__, __LINE__, "p.ax >= kw || p.ay >= kh");
    p.normalize = normalize ? 1 : 0;
    const int area = kw * kh;
    const double scale = 1.0 / area;
    p.scaleD = scale; p.scaleF = (float)scale;
    if (normalize && area == 1) p.normalize = 0;                       // scale == 1: the reference skips the multiply
    if (src_depth == D32F || src_depth == D64F) p.mode = 2;
    else if (src_depth == D8U && dst_depth == D8U && area <= 256) {
        p.mode = 0;
        // ColumnSum<ushort,uchar> constructor (box_filter.simd.hpp:441-455)
        const int d = (int)nearbyint(1.0 / scale);
        double scalef = ((double)(1 << 23)) / d;
        p.divScale = (int)std::floor(scalef);
        scalef -= p.divScale;
        p.divDelta = d / 2;
        if (scalef < 0.5) p.divDelta++; else p.divScale++;
    } else {
        p.mode = 1;
        const long long lim = src_depth == D8U ? (1LL << 23) : src_depth == D16U ? (1LL << 15) : (1LL << 16);
        if (normalize && area > lim) return mi355::declined(__func__, __LINE__, "normalize && area > lim");    // the reference switches to double sums there
    }
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
    const int se = depthSize(src_depth), de = depthSize(dst_depth);
    const int fullW = margin_left + width + margin_right, fullH = margin_top + height + margin_bottom;
    if (nframes >= 1) {
        if (!isDevicePtr(src_data) || !isDevicePtr(dst_data)) return setError(MI355CV_NOT_IMPLEMENTED, "%s: batch entry needs device-resident frames", entry);
        if (nframes == 1) { sframe = 0; dframe = 0; }
        if (p.mode == 0 && p.normalize && kw == kh && p.ax == kw / 2 && p.ay == kh / 2 &&
            seprollBox(src_data, src_step, sframe, dst_data, dst_step, dframe, nframes, width, height, cn, kw, (unsigned)p.divScale, (unsigned)p.divDelta, border, stream()))
            return stg.finish(entry);
        if (p.mode == 2 && src_depth == D32F && dst_depth == D32F && cn == 1 && kw == kh && p.ax == kw / 2 && p.ay == kh / 2 &&
            seprollBoxF32(src_data, src_step, sframe, dst_data, dst_step, dframe, nframes, width, height, kw, p.normalize != 0, border, stream()))
            return stg.finish(entry);
        if (boxOnMatrixCores(stg, p, src_data, src_step, sframe, dst_data, dst_step, dframe, nframes, width, height, cn, src_depth, dst_depth, width, height, 0, 0, border)) return stg.finish(entry);
        if (boxTwoPass(stg, p, src_data, src_step, sframe, dst_data, dst_step, dframe, nframes, width, height, cn, src_depth, dst_depth, width, height, 0, 0, border)) return stg.finish(entry);
        dim3 grid(divUp(width * cn, 64), divUp(height, 4));
        for (int f = 0; f < nframes; f++)
            hipLaunchKernelGGL(k_box_generic, grid, dim3(256), 0, stream(), src_data + (size_t)f * sframe, src_step, dst_data + (size_t)f * dframe, dst_step, width, height, cn,
                               src_depth, dst_depth, width, height, 0, 0, border, p);
        noteKernel("k_box_generic %dx%d window, depth %d -> %d, %d channel(s)", kw, kh, src_depth, dst_depth, cn);
        return stg.finish(entry);
    }
    size_t dss, dds;
    const uchar* top = src_data - (ptrdiff_t)margin_top * (ptrdiff_t)src_step - (ptrdiff_t)margin_left * cn * se;
    const uchar* dtop = stg.in(top, src_step, (size_t)fullW * cn * se, fullH, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * cn * de, height, &dds);
    if (!dtop || !dd) return mi355::declined(__func__, __LINE__, "!dtop || !dd");
    const uchar* ds = dtop + (size_t)margin_top * dss + (size_t)margin_left * cn * se;
    const Roi roiv = {fullW, fullH, margin_left, margin_top};
    const Roi* roi = (fullW != width || fullH != height) ? &roiv : nullptr;       // a submatrix with real pixels around it: the rolling kernels store the window
    if (p.mode == 0 && p.normalize && kw == kh && p.ax == kw / 2 && p.ay == kh / 2 &&
        seprollBox(ds, dss, 0, dd, dds, 0, 1, width, height, cn, kw, (unsigned)p.divScale, (unsigned)p.divDelta, border, stream(), roi))
        return stg.finish(entry);
    if (p.mode == 2 && src_depth == D32F && dst_depth == D32F && cn == 1 && kw == kh && p.ax == kw / 2 && p.ay == kh / 2 &&
        seprollBoxF32(ds, dss, 0, dd, dds, 0, 1, width, height, kw, p.normalize != 0, border, stream(), roi))
        return stg.finish(entry);
    if (boxOnMatrixCores(stg, p, ds, dss, 0, dd, dds, 0, 1, width, height, cn, src_depth, dst_depth, fullW, fullH, margin_left, margin_top, border)) return stg.finish(entry);
    if (boxTwoPass(stg, p, ds, dss, 0, dd, dds, 0, 1, width, height, cn, src_depth, dst_depth, fullW, fullH, margin_left, margin_top, border)) return stg.finish(entry);
    dim3 grid(divUp(width * cn, 64), divUp(height, 4));
    hipLaunchKernelGGL(k_box_generic, grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, cn, src_depth, dst_depth,
                       fullW, fullH, margin_left, margin_top, border, p);
    noteKernel("k_box_generic %dx%d window, depth %d -> %d, %d channel(s)", kw, kh, src_depth, dst_depth, cn);
    return stg.finish(entry);
}

} // extern "C"
