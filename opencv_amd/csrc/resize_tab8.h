// resize_tab8.h -- INTER_CUBIC / INTER_LANCZOS4 cv::resize of CV_8U images (resize.cpp:974-1003 coefficients, :1877-2158 passes) on tiles of 256 x 16
// output elements, four elements per lane:
//   stage   the source bytes a tile reads (rows rmin .. rmin + R - 1, the byte columns between its first and last tap) go to LDS once
//   hpass   the horizontal sums S_k of every staged row for the tile's 256 elements (integer taps * 2048, exact) -> LDS ints
//   vpass   each lane combines NT of them for its four neighbouring elements of a row (the reference's vector body in float for cubic, the integer
//           (sum + 2^21) >> 22 form in its scalar tail and for Lanczos) and stores one dword
// against the 64 x 16 tile kernel (warp.hip k_resize_tiled), whose lanes gather their taps byte by byte from global memory (a byte gather is served lane by
// lane: profiles/r02_warp_pmc.txt) and store single bytes.  The phases are functions of the thread index so that tests/hostemu/resize_tab8_emu.cpp can run
// a workgroup thread by thread on the CPU against the pinned restatement; the tap tables (also used by the other cubic / Lanczos kernels) are built here.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>

#ifndef MI355_HD
#  if defined(__HIPCC__)
#    define MI355_HD __host__ __device__ __forceinline__
#  else
#    define MI355_HD inline
#  endif
#endif

namespace rt8 {

template <int NT> struct Tap { int s; float f[NT]; short i[NT]; };          // first source index (before the -OFF shift), float taps, taps * 2048

// INTER_CUBIC coefficients (resize.cpp interpolateCubic, A = -0.75) for destination index d at scale = 1 / inv_scale
inline void buildCubicTab(int dsize, double scale, std::vector<Tap<4>>& tab)
{
    tab.resize((size_t)dsize);
    const float A = -0.75f;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int sI = (int)f; sI -= sI > f;                                      // cvFloor
        f -= sI;
        Tap<4>& t = tab[(size_t)d];
        t.s = sI;
        t.f[0] = ((A * (f + 1) - 5 * A) * (f + 1) + 8 * A) * (f + 1) - 4 * A;
        t.f[1] = ((A + 2) * f - (A + 3)) * f * f + 1;
        t.f[2] = ((A + 2) * (1 - f) - (A + 3)) * (1 - f) * (1 - f) + 1;
        t.f[3] = 1.f - t.f[0] - t.f[1] - t.f[2];
        for (int k = 0; k < 4; k++) { const long q = lrintf(t.f[k] * 2048); t.i[k] = (short)(q < -32768 ? -32768 : q > 32767 ? 32767 : q); }
    }
}

// INTER_LANCZOS4 (resize.cpp:974-1003): 8 taps at s-3 .. s+4
inline void buildLanczosTab(int dsize, double scale, std::vector<Tap<8>>& tab)
{
    static const double s45 = 0.70710678118654752440084436210485, pi = 3.1415926535897932384626433832795;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    tab.resize((size_t)dsize);
    for (int d = 0; d < dsize; d++) {
        float x = (float)((d + 0.5) * scale - 0.5);
        int sI = (int)x; sI -= sI > x;                                      // cvFloor
        x -= sI;
        Tap<8>& t = tab[(size_t)d];
        t.s = sI;
        float sum = 0;
        const double y0 = -(x + 3) * pi * 0.25, s0 = sin(y0), c0 = cos(y0);
        for (int i = 0; i < 8; i++) {
            const float y0_ = (x + 3 - i);
            if (fabsf(y0_) >= 1e-6f) { const double y = -y0_ * pi * 0.25; t.f[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y)); }
            else t.f[i] = 1e30f;
            sum += t.f[i];
        }
        sum = 1.f / sum;
        for (int i = 0; i < 8; i++) {
            t.f[i] *= sum;
            const long q = lrintf(t.f[i] * 2048); t.i[i] = (short)(q < -32768 ? -32768 : q > 32767 ? 32767 : q);
        }
    }
}

constexpr int TW = 256, TH = 16;                                            // a tile: 256 elements (pixels x channels) by 16 rows, 256 threads
struct Geom { int sw, sh, dw, dh, cn, sp; };                                // sp: LDS pitch of a staged row in bytes (host bound, multiple of 4)

MI355_HD int clipI(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }
// 24-bit multiply (v_mul_i32_i24 / v_mad_i32_i24: full rate; the 32-bit v_mul_lo_u32 is a quarter-rate instruction): both operands within +-2^23 --
// a byte times a tap * 2048, a horizontal sum (< 2^21) times a tap, an index times a channel count or an LDS pitch
MI355_HD int mul24(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}
MI355_HD int roundHalfEven(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rn(v);
#else
    return (int)lrintf(v);
#endif
}

// host: LDS pitch that holds any tile's staged row (xs range of 256 elements: ((TW - 1) / cn + 1) * scale + 1 + NT pixels at most)
inline int stagePitch(int cn, double scale, int nt) { const int px = (int)ceil(((TW - 1) / cn + 1) * scale) + nt + 2; return (px * cn + 7) & ~3; }

// what every thread of the workgroup derives from the block index
template <int NT> struct Tile {
    int e0, width, dy0, rmin, R, cb0, nb;                                  // first element, elements per row, first row, first source row, staged rows, first staged byte column, staged bytes per row
};
template <int NT>
MI355_HD Tile<NT> tileOf(const Geom& g, int bx, int by, const Tap<NT>* xt, const Tap<NT>* yt)
{
    constexpr int OFF = NT / 2 - 1;
    Tile<NT> t;
    t.e0 = bx * TW; t.width = g.dw * g.cn; t.dy0 = by * TH;
    const int dyLast = (t.dy0 + TH < g.dh ? t.dy0 + TH : g.dh) - 1;
    t.rmin = yt[t.dy0].s - OFF; t.R = yt[dyLast].s - OFF + NT - 1 - t.rmin + 1;
    const int eLast = (t.e0 + TW < t.width ? t.e0 + TW : t.width) - 1;
    const int x0 = clipI(xt[t.e0 / g.cn].s - OFF, 0, g.sw), x1 = clipI(xt[eLast / g.cn].s - OFF + NT - 1, 0, g.sw);
    t.cb0 = x0 * g.cn; t.nb = (x1 - x0 + 1) * g.cn;
    return t;
}

// stage: thread tid copies bytes tid, tid + 256, ... of every staged row; eight rows' loads are issued before the first LDS store, so that their
// latencies overlap (a load-store-load-store chain through byte pointers cannot be reordered by the compiler: one memory latency per row, and the
// first version of this kernel was bound by exactly that -- profiles/r03_resize_tab8_ab.txt)
template <int NT>
MI355_HD void stage(int tid, const Geom& g, const Tile<NT>& t, const unsigned char* src, size_t sstep, unsigned char* ldsSrc)
{
    for (int b = tid; b < t.nb; b += 256) {
        for (int r0 = 0; r0 < t.R; r0 += 8) {
            unsigned char v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int r = r0 + k < t.R ? r0 + k : t.R - 1;             // the tail repeats the last row: always a legal address
                v[k] = src[(size_t)clipI(t.rmin + r, 0, g.sh) * sstep + t.cb0 + b];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) if (r0 + k < t.R) ldsSrc[mul24(r0 + k, g.sp) + b] = v[k];
        }
    }
}

// hpass: lane lx = tid & 63 owns elements e0 + lx, + 64, + 128, + 192 (neighbouring lanes read neighbouring source bytes and write neighbouring ints:
// no LDS bank conflicts), wave w = tid >> 6 the staged rows w, w + 4, ...
// the taps a thread needs in the two passes, fetched at the head of the workgroup's work so that these loads, the staging loads and each other overlap
// (fetched where they are used, each costs a memory latency in front of a short loop: the first version of this kernel spent most of its time there)
template <int NT> struct HTaps { Tap<NT> tx[4]; int cc[4]; };
template <int NT> struct VTaps { Tap<NT> ty[TH / 4]; };
template <int NT>
MI355_HD void loadTaps(int tid, const Geom& g, const Tile<NT>& t, const Tap<NT>* xt, const Tap<NT>* yt, HTaps<NT>& h, VTaps<NT>& v)
{
    const int lx = tid & 63, w = tid >> 6;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int e = t.e0 + lx + 64 * q, ev = e < t.width ? e : t.width - 1;
        const int dx = ev / g.cn;
        h.cc[q] = ev - mul24(dx, g.cn);
        h.tx[q] = xt[dx];
    }
#pragma unroll
    for (int k4 = 0; k4 < TH / 4; k4++) { const int dy = t.dy0 + w + 4 * k4; v.ty[k4] = yt[dy < g.dh ? dy : g.dh - 1]; }
}

template <int NT>
MI355_HD void hpass(int tid, const Geom& g, const Tile<NT>& t, const HTaps<NT>& ht, const unsigned char* ldsSrc, int* H)
{
    constexpr int OFF = NT / 2 - 1;
    const int lx = tid & 63, w = tid >> 6;
    const Tap<NT>* tx = ht.tx; const int* cc = ht.cc;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int el = lx + 64 * q;
        if (t.e0 + el >= t.width) break;
        int xs[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) xs[j] = mul24(clipI(tx[q].s - OFF + j, 0, g.sw), g.cn) + cc[q] - t.cb0;
        for (int r = w; r < t.R; r += 4) {
            const unsigned char* row = ldsSrc + mul24(r, g.sp);
            int v = 0;
#pragma unroll
            for (int j = 0; j < NT; j++) v += mul24((int)row[xs[j]], (int)tx[q].i[j]);
            H[r * TW + el] = v;
        }
    }
}

// vpass: lane lx owns the four NEIGHBOURING elements e0 + 4 lx .. + 3 (one dword of the destination row), rows dy0 + w + 4 k
template <int NT>
MI355_HD void vpass(int tid, const Geom& g, const Tile<NT>& t, const VTaps<NT>& vt, const int* H, unsigned char* dst, size_t dstep)
{
    constexpr int OFF = NT / 2 - 1;
    const int lx = tid & 63, w = tid >> 6, el0 = 4 * lx, e = t.e0 + el0;
    if (e >= t.width) return;
    const int body = (t.width / 8) * 8;                                     // VResizeCubicVec_32s8u covers the first width / 8 * 8 elements of a row
#pragma unroll
    for (int k4 = 0; k4 < TH / 4; k4++) {
        const int dy = t.dy0 + w + 4 * k4;
        if (dy >= g.dh) continue;
        const Tap<NT>& ty = vt.ty[k4];
        const int* S = H + (ty.s - OFF - t.rmin) * TW + el0;               // S[k * TW + q]: horizontal sum of window row k for element q
        uint32_t out = 0; int n = 0;
        for (int q = 0; q < 4 && e + q < t.width; q++, n++) {
            int r;
            if (NT == 4 && e + q < body) {                                  // float, taps * 2^-22, nested from the last row
                const float sc = 1.f / (2048.f * 2048.f);
                float v = (float)S[3 * TW + q] * ((float)ty.i[3] * sc);
                v = (float)S[2 * TW + q] * ((float)ty.i[2] * sc) + v;
                v = (float)S[1 * TW + q] * ((float)ty.i[1] * sc) + v;
                v = (float)S[q] * ((float)ty.i[0] * sc) + v;
                r = roundHalfEven(v);
            } else {
                int acc = 0;
#pragma unroll
                for (int k = 0; k < NT; k++) acc += mul24(S[k * TW + q], (int)ty.i[k]);
                r = (acc + (1 << 21)) >> 22;
            }
            out |= (uint32_t)(r < 0 ? 0 : r > 255 ? 255 : r) << (8 * q);
        }
        unsigned char* o = dst + (size_t)dy * dstep + e;
        if (n == 4 && (((uintptr_t)o) & 3) == 0) *reinterpret_cast<uint32_t*>(o) = out;
        else for (int q = 0; q < n; q++) o[q] = (unsigned char)(out >> (8 * q));
    }
}

} // namespace rt8
