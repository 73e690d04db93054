// smooth.hip -- row a1/a2 of SURVEY.md §8: cv::GaussianBlur on CV_8U.
//
// Reference semantics (modules/imgproc/src/smooth.simd.hpp:1926 fixedSmoothInvoker,
// :546 hlineSmooth5N14641, :1561 vlineSmooth5N14641; fixedpoint.inl.hpp:325 ufixedpoint16):
//   H[y][b]  = sum_i kx[i] * p[y][bI(x+i-rx)][c]            (u16, Q8.8, exact: sum kx == 256)
//   dst[y][b] = ( sum_j ky[j] * H[bI(y+j-ry)][b] + 2^15 ) >> 16
// For the sigma==0 binomial tables (smooth.dispatch.cpp:89-145) this collapses to
//   5x5: (sum w_i w_j p + 128) >> 8,  w = 1 4 6 4 1         3x3: (sum w_i w_j p + 8) >> 4,  w = 1 2 1
//
// Two kernels:
//  * k_binomial_roll<KS,CN>: the hot kernel.  HBM-bound (2 B/pixel algorithmic).  One lane owns
//    16 consecutive bytes of a row (one global_load_dwordx4, one global_store_dwordx4) and walks
//    DOWN the image keeping the last KS horizontally-filtered rows in registers (packed 2xu16 per
//    VGPR), so every source byte is fetched once per vertical segment and there is no LDS round
//    trip.  The +-R*cn neighbour bytes come from the adjacent lanes (ds_bpermute crossbar), from
//    a 4-byte side load at the wave edges, or from borderInterpolate at the image edges.
//  * k_sepfixed_generic: any taps / alignment / margins / tiny images.  One thread per output
//    byte, straight from the formula above.  Correctness path, not a fast path.
#include "rt.h"
#include "gausskernel.h"
#include "seproll.h"
#include "seplong.h"
#include "sepmx.h"
#include <cstring>
#include <cstdlib>

using namespace mi355;

namespace {

// ---------------------------------------------------------------------------------- generic
struct FixedTaps { uint16_t kx[33]; uint16_t ky[33]; int nx, ny; };

__global__ __launch_bounds__(256) void k_sepfixed_generic(
    const uchar* __restrict__ src, size_t sstep, size_t sframe,
    uchar* __restrict__ dst, size_t dstep, size_t dframe,
    int W, int H, int cn, int mL, int mT, int mR, int mB, int border, FixedTaps t)
{
    const int b = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (b >= W * cn || y >= H) return;
    src += (size_t)blockIdx.z * sframe;
    dst += (size_t)blockIdx.z * dframe;
    const int x = b / cn, c = b - x * cn;
    const int fullW = mL + W + mR, fullH = mT + H + mB;
    const int rx = t.nx / 2, ry = t.ny / 2;
    uint32_t acc = 0;
    for (int j = 0; j < t.ny; j++) {
        int yy = mi355_borderInterpolate(y + mT + j - ry, fullH, border);
        if (yy < 0) continue;                               // BORDER_CONSTANT: zero row (smooth.simd.hpp:2092)
        const uchar* row = src + (ptrdiff_t)(yy - mT) * (ptrdiff_t)sstep;
        uint32_t h = 0;
        for (int i = 0; i < t.nx; i++) {
            int xx = mi355_borderInterpolate(x + mL + i - rx, fullW, border);
            if (xx < 0) continue;
            h += (uint32_t)t.kx[i] * row[(ptrdiff_t)(xx - mL) * cn + c];
        }
        h = h > 0xFFFFu ? 0xFFFFu : h;                       // ufixedpoint16 saturating add (fixedpoint.inl.hpp:345)
        uint64_t a = (uint64_t)acc + (uint64_t)t.ky[j] * h;
        acc = a > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a; // ufixedpoint32 saturating add (:247)
    }
    uint32_t r = (uint32_t)(((uint64_t)acc + 0x8000u) >> 16);
    dst[(size_t)y * dstep + b] = (uchar)(r > 255 ? 255 : r);
}

// ---------------------------------------------------------------------------------- rolling kernel
constexpr int floordiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
constexpr int mod2(int a) { return ((a % 2) + 2) % 2; }

// The lane's window of one source row as two planes of packed u16 pairs:
//   E[d] = (byte 4d, byte 4d+2), O[d] = (byte 4d+1, byte 4d+3), d = -HD .. 3+HD (array index d+HD)
// pair<Q,S>(k): the two bytes (4k+Q+S, 4k+Q+S+2) as a packed u16 pair -- i.e. the neighbours at
// byte distance S of output pair (plane Q, dword k).
template <int Q, int S, int HD>
__device__ __forceinline__ uint32_t pairAt(const uint32_t* E, const uint32_t* O, int k)
{
    constexpr int q2 = mod2(Q + S);
    constexpr int f = floordiv2(Q + S);
    const uint32_t* P = q2 ? O : E;
    if constexpr (mod2(f) == 0) {
        return P[k + f / 2 + HD];
    } else {
        constexpr int lo = floordiv2(f - 1) + 0;   // (f-1)/2, f odd
        return __builtin_amdgcn_alignbit(P[k + lo + 1 + HD], P[k + lo + HD], 16);
    }
}

template <int HD> struct RawRow {
    uint4 m;              // the lane's 16 bytes
    uint32_t hl[HD];      // side-loaded left halo (meaningful on lane 0 only)
    uint32_t hr[HD];      // side-loaded right halo (lane 63 / last chunk only)
};

template <int KS, int CN> struct RollCfg {
    static constexpr int R = KS / 2;
    static constexpr int HB = R * CN;            // halo bytes per side
    static constexpr int HD = (HB + 3) / 4;      // halo dwords per side
};

// issue the global loads of source row yy for this lane (no dependent ALU: keeps loads in flight)
template <int KS, int CN>
__device__ __forceinline__ void loadRow(RawRow<RollCfg<KS, CN>::HD>& r, const uchar* __restrict__ src, size_t sstep,
                                        int yy, int W, int H, int c, int nchunks, int lane, int border)
{
    constexpr int HD = RollCfg<KS, CN>::HD, HB = RollCfg<KS, CN>::HB;
    r.m = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < HD; d++) { r.hl[d] = 0; r.hr[d] = 0; }
    const int ry = mi355_borderInterpolate(yy, H, border);
    if (ry < 0 || c >= nchunks) return;          // constant border row / idle lane: zeros
    const uchar* row = src + (size_t)ry * sstep;
    r.m = *reinterpret_cast<const uint4*>(row + 16 * (size_t)c);
    if (lane == 0) {
        if (c > 0) {
#pragma unroll
            for (int d = 0; d < HD; d++) r.hl[d] = *reinterpret_cast<const uint32_t*>(row + 16 * (size_t)c - 4 * (HD - d));
        } else {
            // bytes -HB..-1 of the row = pixels -R..-1 by borderInterpolate
#pragma unroll
            for (int t = -HB; t < 0; t++) {
                const int px = (t - (CN - 1)) / CN;            // floor(t/CN) for t<0
                const int ch = t - px * CN;
                const int sp = mi355_borderInterpolate(px, W, border);
                const uint32_t v = sp < 0 ? 0u : row[sp * CN + ch];
                r.hl[(4 * HD + t) >> 2] |= v << (8 * ((4 * HD + t) & 3));
            }
        }
    }
    if (c == nchunks - 1) {
#pragma unroll
        for (int t = 0; t < HB; t++) {
            const int px = W + t / CN, ch = t % CN;
            const int sp = mi355_borderInterpolate(px, W, border);
            const uint32_t v = sp < 0 ? 0u : row[sp * CN + ch];
            r.hr[t >> 2] |= v << (8 * (t & 3));
        }
    } else if (lane == 63) {
#pragma unroll
        for (int d = 0; d < HD; d++) r.hr[d] = *reinterpret_cast<const uint32_t*>(row + 16 * (size_t)c + 16 + 4 * d);
    }
}

// horizontal pass of one row: raw bytes -> 8 packed-u16 dwords (Hrow[0..3] = even bytes, [4..7] = odd bytes)
template <int KS, int CN>
__device__ __forceinline__ void hfilter(uint32_t (&Hrow)[8], const RawRow<RollCfg<KS, CN>::HD>& r, int c, int nchunks, int lane)
{
    constexpr int HD = RollCfg<KS, CN>::HD;
    constexpr int NW = 4 + 2 * HD;
    uint32_t X[NW];
    const uint32_t mv[4] = {r.m.x, r.m.y, r.m.z, r.m.w};
#pragma unroll
    for (int d = 0; d < HD; d++) {
        uint32_t l = __shfl_up(mv[4 - HD + d], 1);
        uint32_t rr = __shfl_down(mv[d], 1);
        X[d] = (lane == 0) ? r.hl[d] : l;
        X[HD + 4 + d] = (lane == 63 || c == nchunks - 1) ? r.hr[d] : rr;
    }
#pragma unroll
    for (int d = 0; d < 4; d++) X[HD + d] = mv[d];
    uint32_t E[NW], O[NW];
#pragma unroll
    for (int d = 0; d < NW; d++) {
        E[d] = __builtin_amdgcn_perm(0u, X[d], 0x0c020c00u);   // (b0, b2) zero-extended to u16 pairs
        O[d] = __builtin_amdgcn_perm(0u, X[d], 0x0c030c01u);   // (b1, b3)
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if constexpr (KS == 5) {
            {   uint32_t a = pairAt<0, -2 * CN, HD>(E, O, k) + pairAt<0, 2 * CN, HD>(E, O, k);
                uint32_t b = pairAt<0, -CN, HD>(E, O, k) + pairAt<0, CN, HD>(E, O, k);
                uint32_t m = pairAt<0, 0, HD>(E, O, k);
                Hrow[k] = a + (b << 2) + (m << 1) + (m << 2); }
            {   uint32_t a = pairAt<1, -2 * CN, HD>(E, O, k) + pairAt<1, 2 * CN, HD>(E, O, k);
                uint32_t b = pairAt<1, -CN, HD>(E, O, k) + pairAt<1, CN, HD>(E, O, k);
                uint32_t m = pairAt<1, 0, HD>(E, O, k);
                Hrow[4 + k] = a + (b << 2) + (m << 1) + (m << 2); }
        } else {
            Hrow[k]     = pairAt<0, -CN, HD>(E, O, k) + pairAt<0, CN, HD>(E, O, k) + (pairAt<0, 0, HD>(E, O, k) << 1);
            Hrow[4 + k] = pairAt<1, -CN, HD>(E, O, k) + pairAt<1, CN, HD>(E, O, k) + (pairAt<1, 0, HD>(E, O, k) << 1);
        }
    }
}

template <int KS, int CN>
__global__ __launch_bounds__(256) void k_binomial_roll(
    const uchar* __restrict__ src, size_t sstep, size_t sframe,
    uchar* __restrict__ dst, size_t dstep, size_t dframe,
    int W, int H, int nchunks, int segRows, int border)
{
    constexpr int R = KS / 2, HD = RollCfg<KS, CN>::HD;
    const int lane = threadIdx.x & 63;
    const int c = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
    const int y0 = blockIdx.y * segRows;
    const int y1 = min(H, y0 + segRows);
    src += (size_t)blockIdx.z * sframe;
    dst += (size_t)blockIdx.z * dframe;
    const bool active = c < nchunks;

    uint32_t Hw[KS][8];
    {   // prologue: rows y0-R .. y0+R-1 into slots 0..KS-2
        RawRow<HD> pre[KS - 1];
#pragma unroll
        for (int i = 0; i < KS - 1; i++) loadRow<KS, CN>(pre[i], src, sstep, y0 - R + i, W, H, c, nchunks, lane, border);
#pragma unroll
        for (int i = 0; i < KS - 1; i++) hfilter<KS, CN>(Hw[i], pre[i], c, nchunks, lane);
    }
    for (int y = y0; y < y1; y += KS) {
        RawRow<HD> raw[KS];
#pragma unroll
        for (int u = 0; u < KS; u++)
            if (y + u < y1) loadRow<KS, CN>(raw[u], src, sstep, y + u + R, W, H, c, nchunks, lane, border);
#pragma unroll
        for (int u = 0; u < KS; u++) {
            if (y + u < y1) {
                hfilter<KS, CN>(Hw[(KS - 1 + u) % KS], raw[u], c, nchunks, lane);
                uint32_t o[4];
                if constexpr (KS == 5) {
                    const uint32_t* h0 = Hw[(u + 0) % 5]; const uint32_t* h1 = Hw[(u + 1) % 5];
                    const uint32_t* h2 = Hw[(u + 2) % 5]; const uint32_t* h3 = Hw[(u + 3) % 5];
                    const uint32_t* h4 = Hw[(u + 4) % 5];
                    // 6 h2 as 4 h2 + 2 h2 inside two shift-adds: four full-rate VALU operations per dword (add3, lshl_add, lshl_add, add3); written
                    // as (h2 << 1) + (h2 << 2) the compiler emits v_mul_lo_u32 by 6, which issues at a quarter of that rate
                    uint32_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const uint32_t b4 = lshlAdd(h1[i] + h3[i] + h2[i], 2, 0x00800080u);
                        v[i] = h0[i] + h4[i] + lshlAdd(h2[i], 1, b4);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) o[k] = __builtin_amdgcn_perm(v[4 + k], v[k], 0x07030501u);  // (Ve.b1,Vo.b1,Ve.b3,Vo.b3) = >>8
                } else {
                    const uint32_t* h0 = Hw[(u + 0) % 3]; const uint32_t* h1 = Hw[(u + 1) % 3]; const uint32_t* h2 = Hw[(u + 2) % 3];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t ve = ((h0[k] + h2[k] + (h1[k] << 1) + 0x00080008u) >> 4) & 0x00FF00FFu;
                        uint32_t vo = ((h0[4 + k] + h2[4 + k] + (h1[4 + k] << 1) + 0x00080008u) >> 4) & 0x00FF00FFu;
                        o[k] = ve | (vo << 8);
                    }
                }
                if (active)
                    *reinterpret_cast<uint4*>(dst + (size_t)(y + u) * dstep + 16 * (size_t)c) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------- rolling kernel v2
// Same arithmetic as k_binomial_roll, restructured after the first rocprof passes (profiles/r01_*):
//  * one WAVE = one work item (strip of 64 chunks x segment of rows x frame); work items are ordered so
//    that the resident waves sweep memory linearly, like a streaming copy;
//  * the row loop is free of exec-mask branches and calls: v1's divergent edge loads made the compiler
//    drain vmcnt to 0 every row.  Every lane issues the same two loads per row (its 16 B + one side
//    dword whose address only differs on lanes 0 / 63), image-border halos are rebuilt from the lane's
//    own registers, and border rows are resolved to scalars before the loop;
//  * a KS-deep ring of in-flight row loads per wave (a slot is refilled right after it is consumed);
//  * neighbour bytes through DPP wave_shr/wave_shl (one v_mov_dpp) instead of ds_bpermute;
//  * optional non-temporal stores (results are never re-read by this kernel).
// Handles BORDER_CONSTANT/REPLICATE/REFLECT/REFLECT_101 with W > KS/2 (halo source bytes then lie
// inside the first/last 16-byte chunk); BORDER_WRAP and narrower images use k_binomial_roll.
template <int HD> struct RawRow2 {
    uint4 m;
    uint32_t side[HD];   // lane 0: bytes just left of its chunk; lane 63: bytes just right of it
};

// Image-border halos are rebuilt from the edge lane's own 16 bytes with three v_perm per halo dword:
//   t1 = perm(m.y, m.x, selA) gathers candidates from bytes 0..7, t2 = perm(m.w, m.z, selB) from bytes 8..15,
//   halo = perm(t2, t1, selC) picks per byte (0x0c = constant zero).  Selectors are wave-uniform.
template <int HD> struct EdgeSel {
    uint32_t la[HD], lb[HD], lc[HD];   // left halo dwords
    uint32_t ra[HD], rb[HD], rc[HD];   // right halo dwords
};

__device__ __forceinline__ void selSetByte(uint32_t& a, uint32_t& b, uint32_t& c, int j, int idx /* 0..15 or <0 */)
{
    const uint32_t sh = 8u * (uint32_t)j, clr = ~(0xffu << sh);
    uint32_t va = 0x0cu, vb = 0x0cu, vc = 0x0cu;
    if (idx >= 8)      { vb = (uint32_t)(idx - 8); vc = 4u + (uint32_t)j; }
    else if (idx >= 0) { va = (uint32_t)idx;       vc = (uint32_t)j; }
    a = (a & clr) | (va << sh); b = (b & clr) | (vb << sh); c = (c & clr) | (vc << sh);
}

__device__ __forceinline__ uint32_t gather16(const uint4& m, uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t t1 = __builtin_amdgcn_perm(m.y, m.x, a);
    const uint32_t t2 = __builtin_amdgcn_perm(m.w, m.z, b);
    return __builtin_amdgcn_perm(t2, t1, c);
}

template <int KS, int CN, bool NTL = false>
__device__ __forceinline__ void issueRow(RawRow2<RollCfg<KS, CN>::HD>& r, const uchar* __restrict__ row, int mainOff, int sideOff)
{
    constexpr int HD = RollCfg<KS, CN>::HD;
    if constexpr (NTL) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + mainOff));
        r.m = make_uint4(v.x, v.y, v.z, v.w);
    } else
    r.m = *reinterpret_cast<const uint4*>(row + mainOff);
#pragma unroll
    for (int d = 0; d < HD; d++) r.side[d] = *reinterpret_cast<const uint32_t*>(row + sideOff + 4 * d);
}

template <int KS, int CN, bool EB = true>
__device__ __forceinline__ void hfilter2(uint32_t (&Hrow)[8], const RawRow2<RollCfg<KS, CN>::HD>& r,
                                         bool hasFirst, bool hasLast, bool isLastChunk, const EdgeSel<RollCfg<KS, CN>::HD>& es)
{
    constexpr int HD = RollCfg<KS, CN>::HD;
    constexpr int NW = 4 + 2 * HD;
    uint32_t X[NW];
    const uint32_t mv[4] = {r.m.x, r.m.y, r.m.z, r.m.w};
    uint32_t hl[HD], hr[HD];
#pragma unroll
    for (int d = 0; d < HD; d++) { hl[d] = r.side[d]; hr[d] = r.side[d]; }
    // the two edge cases are wave-uniform; the empty volatile asm keeps them real (scalar) branches -- if-converted into selects, their six
    // v_perm would be paid by every strip of every row, and three strips in four have no image border on one or both sides
    if (hasFirst) {          // lane 0 is chunk 0, its left halo is the image border
#pragma unroll
        for (int d = 0; d < HD; d++) { hl[d] = gather16(r.m, es.la[d], es.lb[d], es.lc[d]); if constexpr (EB) asm volatile("" : "+v"(hl[d])); }
    }
    uint32_t hb[HD];
#pragma unroll
    for (int d = 0; d < HD; d++) hb[d] = 0;
    if (hasLast) {           // some lane is the last chunk, its right halo is the image border
#pragma unroll
        for (int d = 0; d < HD; d++) { hb[d] = gather16(r.m, es.ra[d], es.rb[d], es.rc[d]); if constexpr (EB) asm volatile("" : "+v"(hb[d])); }
    }
#pragma unroll
    for (int d = 0; d < HD; d++) {
        // wave_shr:1 -> lane i takes lane i-1, lane 0 keeps `old` (its left halo);  wave_shl:1 mirrors it
        X[d] = __builtin_amdgcn_update_dpp(hl[d], mv[4 - HD + d], 0x138, 0xf, 0xf, false);
        uint32_t rr = __builtin_amdgcn_update_dpp(hr[d], mv[d], 0x130, 0xf, 0xf, false);
        if (hasLast) { rr = isLastChunk ? hb[d] : rr; if constexpr (EB) asm volatile("" : "+v"(rr)); }
        X[HD + 4 + d] = rr;
    }
#pragma unroll
    for (int d = 0; d < 4; d++) X[HD + d] = mv[d];
    uint32_t E[NW], O[NW];
#pragma unroll
    for (int d = 0; d < NW; d++) {
        E[d] = __builtin_amdgcn_perm(0u, X[d], 0x0c020c00u);
        O[d] = __builtin_amdgcn_perm(0u, X[d], 0x0c030c01u);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if constexpr (KS == 5) {
            {   uint32_t a = pairAt<0, -2 * CN, HD>(E, O, k) + pairAt<0, 2 * CN, HD>(E, O, k);
                uint32_t b = pairAt<0, -CN, HD>(E, O, k) + pairAt<0, CN, HD>(E, O, k);
                uint32_t m = pairAt<0, 0, HD>(E, O, k);
                Hrow[k] = a + (b << 2) + (m << 1) + (m << 2); }
            {   uint32_t a = pairAt<1, -2 * CN, HD>(E, O, k) + pairAt<1, 2 * CN, HD>(E, O, k);
                uint32_t b = pairAt<1, -CN, HD>(E, O, k) + pairAt<1, CN, HD>(E, O, k);
                uint32_t m = pairAt<1, 0, HD>(E, O, k);
                Hrow[4 + k] = a + (b << 2) + (m << 1) + (m << 2); }
        } else {
            Hrow[k]     = pairAt<0, -CN, HD>(E, O, k) + pairAt<0, CN, HD>(E, O, k) + (pairAt<0, 0, HD>(E, O, k) << 1);
            Hrow[4 + k] = pairAt<1, -CN, HD>(E, O, k) + pairAt<1, CN, HD>(E, O, k) + (pairAt<1, 0, HD>(E, O, k) << 1);
        }
    }
}

template <int KS, int CN, bool NT, bool NTL, int WPS, bool EB = true>
__global__ __launch_bounds__(256, WPS) void k_binomial_roll2(
    const uchar* __restrict__ src, size_t sstep, size_t sframe,
    uchar* __restrict__ dst, size_t dstep, size_t dframe,
    int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes, int border, int alt)
{
    constexpr int R = KS / 2, HD = RollCfg<KS, CN>::HD, HB = RollCfg<KS, CN>::HB;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int strip = wid % nstrips;
    const int t0 = wid / nstrips;
    int seg = t0 % nseg;
    const int frame = t0 / nseg;
    if (frame >= nframes) return;
    // alt: vertically adjacent segments are walked in opposite directions (even: down, odd: up) so that the R rows two
    // neighbours share are touched by both at the same moment (both start there, or both end there), and the two
    // neighbours are placed 8 work-item groups apart = on the same XCD (same L2) under the round-robin block placement.
    // alt = log2 of the number of vertically adjacent segments that share an XCD (1: pairs, 2: quads, ...): inside such a group every boundary
    // is walked towards by both neighbours at once (odd segments go up, even ones down), so its R halo rows come from L2 for one of them; only
    // the group's outer boundaries are fetched twice -- 2R / (G * segRows) of the rows instead of 2R / (2 * segRows).
    int up = 0;
    if (alt) {
        const int G = 1 << alt, span = 8 * G;
        const int g = seg / span, j = seg - g * span;
        if ((g + 1) * span <= nseg) seg = g * span + G * (j & 7) + (j >> 3);
        else if (alt > 1) {                                    // tail of the frame: pairs
            const int base = g * span, t = seg - base, g2 = t >> 4, j2 = t & 15;
            if (base + (g2 << 4) + 16 <= nseg) seg = base + (g2 << 4) + 2 * (j2 & 7) + (j2 >> 3);
        }
        up = seg & 1;
    }
    const int c = strip * 64 + lane;
    const int y0 = seg * segRows;
    const int y1 = min(H, y0 + segRows);
    src += (size_t)frame * sframe;
    dst += (size_t)frame * dframe;
    const bool active = c < nchunks;
    const bool hasFirst = strip == 0, hasLast = strip == nstrips - 1;
    const bool isLastChunk = c == nchunks - 1;
    const int ce = active ? c : nchunks - 1;                  // idle lanes re-read the last chunk (harmless)
    const int mainOff = 16 * ce;
    // One side-dword load per row for the whole wave, touching only two 64-byte sectors: lanes 0..31 read
    // the bytes just left of the wave's first chunk (lane 0 consumes them), lanes 32..63 the bytes just right
    // of its last chunk (lane 63 consumes them).  At the image edges the address degenerates to in-row bytes
    // whose value is ignored (the halo is then rebuilt from the border rule).
    const int c0 = strip * 64;
    const int leftOff = c0 > 0 ? 16 * c0 - 4 * HD : 0;
    const int rightOff = c0 + 64 < nchunks ? 16 * (c0 + 64) : 16 * (nchunks - 1);
    const int sideOff = lane < 32 ? leftOff : rightOff;

    // image-border halo bytes expressed as byte indices inside the first / last chunk (wave-uniform)
    EdgeSel<HD> es;
#pragma unroll
    for (int d = 0; d < HD; d++) { es.la[d] = es.lb[d] = es.lc[d] = es.ra[d] = es.rb[d] = es.rc[d] = 0x0c0c0c0cu; }
    if (hasFirst) {
#pragma unroll
        for (int t = 0; t < HB; t++) {
            const int bt = t - HB;                             // byte position relative to the row start (<0)
            const int px = (bt - (CN - 1)) / CN;               // floor(bt / CN)
            const int sp = mi355_borderInterpolate(px, W, border);
            const int pos = 4 * HD - HB + t;                   // byte position inside the HD halo dwords
            selSetByte(es.la[pos >> 2], es.lb[pos >> 2], es.lc[pos >> 2], pos & 3, sp < 0 ? -1 : sp * CN + (bt - px * CN));
        }
    }
    if (hasLast) {
#pragma unroll
        for (int t = 0; t < HB; t++) {
            const int sp = mi355_borderInterpolate(W + t / CN, W, border);
            selSetByte(es.ra[t >> 2], es.rb[t >> 2], es.rc[t >> 2], t & 3, sp < 0 ? -1 : sp * CN + (t % CN) - 16 * (nchunks - 1));
        }
    }
    // rows: everything outside [0,H) is resolved here, the loop only selects.  Logical row j of the segment (in walking
    // order) is image row gy(j); the kernel is symmetric, so walking up needs no other change.
    int rowBelow[R], rowAbove[R];
#pragma unroll
    for (int i = 0; i < R; i++) { rowBelow[i] = mi355_borderInterpolate(H + i, H, border); rowAbove[i] = mi355_borderInterpolate(-1 - i, H, border); }
    const int nrows = y1 - y0;
    auto gy = [&](int j) -> int { return up ? y1 - 1 - j : y0 + j; };
    auto rowIdx = [&](int g) -> int {                          // g in [-R, H+R)
        int ry = g;
#pragma unroll
        for (int i = 0; i < R; i++) { ry = (g == H + i) ? rowBelow[i] : ry; ry = (g == -1 - i) ? rowAbove[i] : ry; }
        return ry;
    };

    uint32_t Hw[KS][8];
#pragma unroll
    for (int i = 0; i < KS - 1; i++) {      // prologue: logical rows -R .. R-1 -> slots 0..KS-2
        const int ry = rowIdx(gy(min(i - R, nrows - 1 + R)));
        if (ry < 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) Hw[i][j] = 0;
        } else {
            RawRow2<HD> pre;
            issueRow<KS, CN>(pre, src + (size_t)ry * sstep, mainOff, sideOff);
            hfilter2<KS, CN, EB>(Hw[i], pre, hasFirst, hasLast, isLastChunk, es);
        }
    }
    RawRow2<HD> raw[KS];
    int rvalid[KS];
#pragma unroll
    for (int u = 0; u < KS; u++) {          // prime the ring: logical rows R+u
        const int ry = rowIdx(gy(min(u + R, nrows - 1 + R)));
        rvalid[u] = ry >= 0;
        issueRow<KS, CN, NTL>(raw[u], src + (size_t)max(ry, 0) * sstep, mainOff, sideOff);
    }
    for (int y = 0; y < nrows; y += KS) {
#pragma unroll
        for (int u = 0; u < KS; u++) {
            if (y + u < nrows) {
                uint32_t (&Hn)[8] = Hw[(KS - 1 + u) % KS];
                if (rvalid[u]) hfilter2<KS, CN, EB>(Hn, raw[u], hasFirst, hasLast, isLastChunk, es);
                else {
#pragma unroll
                    for (int j = 0; j < 8; j++) Hn[j] = 0;
                }
                {   // refill this ring slot with the row needed KS iterations from now (clamped: always a legal address)
                    const int ry = rowIdx(gy(min(y + u + KS + R, nrows - 1 + R)));
                    rvalid[u] = ry >= 0;
                    issueRow<KS, CN, NTL>(raw[u], src + (size_t)max(ry, 0) * sstep, mainOff, sideOff);
                }
                uint32_t o[4];
                if constexpr (KS == 5) {
                    const uint32_t* h0 = Hw[(u + 0) % 5]; const uint32_t* h1 = Hw[(u + 1) % 5];
                    const uint32_t* h2 = Hw[(u + 2) % 5]; const uint32_t* h3 = Hw[(u + 3) % 5];
                    const uint32_t* h4 = Hw[(u + 4) % 5];
                    uint32_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {                // four full-rate operations per dword, no multiply by 6 (see k_binomial_roll)
                        if constexpr (EB) {
                            const uint32_t b4 = lshlAdd(h1[i] + h3[i] + h2[i], 2, 0x00800080u);
                            v[i] = h0[i] + h4[i] + lshlAdd(h2[i], 1, b4);
                        } else
                            v[i] = (h0[i] + h4[i]) + ((h1[i] + h3[i]) << 2) + (h2[i] << 1) + (h2[i] << 2) + 0x00800080u;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) o[k] = __builtin_amdgcn_perm(v[4 + k], v[k], 0x07030501u);
                } else {
                    const uint32_t* h0 = Hw[(u + 0) % 3]; const uint32_t* h1 = Hw[(u + 1) % 3]; const uint32_t* h2 = Hw[(u + 2) % 3];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t ve = ((h0[k] + h2[k] + (h1[k] << 1) + 0x00080008u) >> 4) & 0x00FF00FFu;
                        uint32_t vo = ((h0[4 + k] + h2[4 + k] + (h1[4 + k] << 1) + 0x00080008u) >> 4) & 0x00FF00FFu;
                        o[k] = ve | (vo << 8);
                    }
                }
                if (active) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 ov = {o[0], o[1], o[2], o[3]};
                    u32x4* dp = reinterpret_cast<u32x4*>(dst + (size_t)gy(y + u) * dstep + 16 * (size_t)c);
                    if constexpr (NT) __builtin_nontemporal_store(ov, dp);
                    else *dp = ov;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------- copy probe
// 16 B/lane streaming copy with the same launch geometry as the rolling kernel's traffic (1 read :
// 1 write): the measured-copy denominator BASELINE.md asks for next to the 8 TB/s spec figure.
template <bool NT>
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n16, int perThread)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    size_t base = ((size_t)blockIdx.x * perThread) * 256 + threadIdx.x;
#pragma unroll 4
    for (int i = 0; i < perThread; i++) {
        size_t j = base + (size_t)i * 256;
        if (j < n16) {
            uint4 v = s[j];
            u32x4 ov = {v.x, v.y, v.z, v.w};
            if constexpr (NT) __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(d + j));
            else d[j] = v;
        }
    }
}

// column-walk copy probe: the rolling kernel's work decomposition and access order (wave = 1 KB wide strip
// walking down `segRows` rows) with the arithmetic removed -- separates access-pattern effects from ALU/wait effects
template <int UNROLL>
__global__ __launch_bounds__(256) void k_copy_colwalk(const uchar* __restrict__ src, uchar* __restrict__ dst, size_t step, size_t frame,
                                                      int H, int nchunks, int nstrips, int segRows, int nseg, int nframes)
{
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int strip = wid % nstrips, t0 = wid / nstrips, seg = t0 % nseg, fr = t0 / nseg;
    if (fr >= nframes) return;
    const int c = strip * 64 + lane;
    if (c >= nchunks) return;
    const int y0 = seg * segRows, y1 = min(H, y0 + segRows);
    const uchar* s = src + (size_t)fr * frame + 16 * (size_t)c;
    uchar* d = dst + (size_t)fr * frame + 16 * (size_t)c;
    for (int y = y0; y < y1; y += UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = *reinterpret_cast<const uint4*>(s + (size_t)min(y + u, y1 - 1) * step);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) if (y + u < y1) *reinterpret_cast<uint4*>(d + (size_t)(y + u) * step) = v[u];
    }
}

// ---------------------------------------------------------------------------------- host side

bool aligned16(const void* p, size_t step) { return (((uintptr_t)p | step) & 15) == 0; }

int envInt(const char* n, int d) { const char* v = getenv(n); return v ? atoi(v) : d; }

int& tuneSeg() { static int v = envInt("MI355CV_GAUSS_SEG", 0); return v; }
int& tuneVariant() { static int v = envInt("MI355CV_GAUSS_VARIANT", 3); return v; }   // 1: k_binomial_roll, 2: roll2, 3: roll2 + nt stores, 4: 3 with the edge halos as selects instead of branches, 5: 3 at 6 waves / SIMD

int& tuneAlt() { static int v = envInt("MI355CV_GAUSS_ALT", 1); return v; }   // 1: alternate walking direction of vertical neighbours
int& tuneLaunchWaves() { static int v = envInt("MI355CV_GAUSS_LAUNCH_WAVES", 393216); return v; }   // work items per launch of a batch (0: one launch)

template <int KS, int CN>
void launchRoll2(const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nframes, int W, int H, int border, hipStream_t st, int nt)
{
    const int nchunks = W * CN / 16;
    const int nstrips = divUp(nchunks, 64);
    int seg = tuneSeg();
    if (seg <= 0) {
        // Short segments keep the set of rows being streamed at any instant compact, although a 5x5 then re-reads 4 / seg of its rows (from
        // L2, thanks to the pairing).  Interleaved A/Bs (tools/ab_gauss.py, tools/footprint_sweep.py, profiles/r02_ab_gauss.txt): 12 and 16
        // rows are within 2 % of each other and the order depends on the box; 16 won on three boxes out of four with this kernel (72.9-75.4 %
        // against 70.8-73.3 %), 20 and 24 rows lose 2-4 %.  Single frames go shorter still so that >= ~2 waves per SIMD exist.
        long long per = (long long)nstrips * nframes;
        long long wantSeg = (2048 + per - 1) / per;
        seg = (int)((H + wantSeg - 1) / wantSeg);
        const int best = 16;
        if (seg > best) seg = best;
        if (seg < KS) seg = KS;
    }
    if (seg > H) seg = H;                       // any length works: the row loop guards its tail rows
    const int nseg = divUp(H, seg);
    // A pass over a large batch is issued as consecutive launches of at most ~tuneLaunchWaves() work items: the same 9216 x 4K frames ran at
    // 70.0-71.0 % of 8 TB/s as ONE launch and at 73.3-75.4 % as 18 launches of 512 frames (tools/split_probe.py, same buffers, launch gaps
    // included) -- over a very long launch the resident work items drift apart, which costs the shared halo rows their L2 hits and spreads the
    // rows being streamed.  Stream order keeps the launches back to back.
    const long long perFrame = (long long)nstrips * nseg;
    long long chunk = tuneLaunchWaves() > 0 ? (tuneLaunchWaves() + perFrame - 1) / perFrame : nframes;
    if (chunk < 1) chunk = 1;
    if (chunk > nframes) chunk = nframes;
    const int nlaunch = (int)((nframes + chunk - 1) / chunk);
    noteKernel("k_binomial_roll2<%d,%d,%s,false,%d,%s> grid=%u x256 seg=%d rows alt=%d, %d launch(es) of <= %lld frames", KS, CN, nt >= 1 ? "true" : "false", nt == 3 ? 6 : 4,
               nt == 2 ? "false" : "true", (unsigned)((perFrame * chunk + 3) / 4), seg, tuneAlt(), nlaunch, chunk);
    for (long long f0 = 0; f0 < nframes; f0 += chunk) {
        const int nf = (int)(nframes - f0 < chunk ? nframes - f0 : chunk);
        const uchar* sp = s + (size_t)f0 * sf;
        uchar* dp = d + (size_t)f0 * df;
        dim3 grid((unsigned)((perFrame * nf + 3) / 4));
        if (nt == 3)      hipLaunchKernelGGL((k_binomial_roll2<KS, CN, true, false, 6>), grid, dim3(256), 0, st, sp, ss, sf, dp, ds, df, W, H, nchunks, nstrips, seg, nseg, nf, border, tuneAlt());
        else if (nt == 2) hipLaunchKernelGGL((k_binomial_roll2<KS, CN, true, false, 4, false>), grid, dim3(256), 0, st, sp, ss, sf, dp, ds, df, W, H, nchunks, nstrips, seg, nseg, nf, border, tuneAlt());
        else if (nt == 1) hipLaunchKernelGGL((k_binomial_roll2<KS, CN, true, false, 4>), grid, dim3(256), 0, st, sp, ss, sf, dp, ds, df, W, H, nchunks, nstrips, seg, nseg, nf, border, tuneAlt());
        else              hipLaunchKernelGGL((k_binomial_roll2<KS, CN, false, false, 4>), grid, dim3(256), 0, st, sp, ss, sf, dp, ds, df, W, H, nchunks, nstrips, seg, nseg, nf, border, tuneAlt());
    }
}

template <int KS, int CN>
void launchRoll(const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nframes, int W, int H, int border, hipStream_t st)
{
    const int nchunks = W * CN / 16;
    const int gx = divUp(nchunks, 256);
    const int segEnv = tuneSeg();
    int seg;
    if (segEnv > 0) seg = segEnv;
    else {
        // enough (strip x segment x frame) work items to fill 256 CUs several times over, while
        // keeping the 2R re-read halo rows per segment small
        long long stripsWaves = (long long)divUp(nchunks, 64) * nframes;
        long long wantSeg = (16384 + stripsWaves - 1) / stripsWaves;
        if (wantSeg < 1) wantSeg = 1;
        seg = (int)((H + wantSeg - 1) / wantSeg);
        if (seg < 4 * KS) seg = 4 * KS;
    }
    seg = divUp(seg, KS) * KS;
    if (seg > H) seg = divUp(H, KS) * KS;
    dim3 grid(gx, divUp(H, seg), nframes);
    hipLaunchKernelGGL((k_binomial_roll<KS, CN>), grid, dim3(256), 0, st, s, ss, sf, d, ds, df, W, H, nchunks, seg, border);
}

bool rollEligible(const uchar* s, size_t ss, size_t sf, const uchar* d, size_t ds, size_t df, int W, int H, int cn, int ks)
{
    if (ks != 3 && ks != 5) return false;
    if (cn < 1 || cn > 4) return false;
    if (!aligned16(s, ss) || !aligned16(d, ds) || (sf & 15) || (df & 15)) return false;
    if ((W * cn) % 16 != 0) return false;
    if (H < 1) return false;
    return true;
}

// from this many taps on (either axis) the matrix-core kernel goes first: it runs at the same speed for 3 .. 33 taps (3.8 us per 4K CV_8UC1 frame), the register-rolling
// kernel slows down with every tap (5 / 7 / 9 taps: 3.5 / 4.3 / 5.5 us, three channels 9 taps 24 us against 11; profiles/r06_sepmx.txt)
static int sepmxMinTaps()
{
    static const int v = std::getenv("MI355CV_SEPMX_MIN_TAPS") ? atoi(std::getenv("MI355CV_SEPMX_MIN_TAPS")) : 7;
    return v;
}

int runSmooth(const char* entry, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe,
              int nframes, int W, int H, int cn, int mL, int mT, int mR, int mB,
              const uint16_t* kx, int nx, const uint16_t* ky, int ny, int border, bool binomial)
{
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    if (W <= 0 || H <= 0 || nframes <= 0 || cn < 1 || cn > 4) return mi355::declined(__func__, __LINE__, "W <= 0 || H <= 0 || nframes <= 0 || cn < 1 || cn > 4");
    if (nx < 1 || ny < 1 || nx > lim::GAUSS8U_MAX_KSIZE || ny > lim::GAUSS8U_MAX_KSIZE || !(nx & 1) || !(ny & 1)) return mi355::declined(__func__, __LINE__, "nx < 1 || ny < 1 || nx or ny > lim::GAUSS8U_MAX_KSIZE || !(nx & 1) || !(ny & 1)");
    if (border < 0 || border > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "border < 0 || border > B_REFLECT_101");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    const bool hostSrc = !isDevicePtr(src);
    if (hostSrc && (size_t)W * H < minPixels()) return mi355::declined(__func__, __LINE__, "hostSrc && (size_t)W * H < minPixels()");
    if (!hostSrc && src == dst) return mi355::declined(__func__, __LINE__, "!hostSrc && src == dst");            // in place on the device (cv::GaussianBlur itself clones, smooth.dispatch.cpp:685)

    size_t dss = 0, dds = 0;
    const size_t rowB = (size_t)W * cn;
    const uchar* dsrc; uchar* ddst;
    if (nframes == 1) {
        // stage the ROI together with its real margins (non-isolated borders read them)
        const uchar* top = src - (ptrdiff_t)mT * (ptrdiff_t)sstep - (ptrdiff_t)mL * cn;
        const uchar* dtop = stg.in(top, sstep, (size_t)(mL + W + mR) * cn, mT + H + mB, &dss);
        if (!dtop) return mi355::declined(__func__, __LINE__, "!dtop");
        dsrc = dtop + (size_t)mT * dss + (size_t)mL * cn;
        ddst = stg.out(dst, dstep, rowB, H, &dds);
        if (!ddst) return mi355::declined(__func__, __LINE__, "!ddst");
    } else {
        // batches are an HBM-resident construct: no per-frame staging
        if (hostSrc || !isDevicePtr(dst)) return setError(MI355CV_NOT_IMPLEMENTED, "%s: batch entry needs device-resident frames", entry);
        dsrc = src; ddst = dst; dss = sstep; dds = dstep;
    }
    hipStream_t st = stream();
    const bool noMargins = !(mL | mT | mR | mB);
    if (binomial && noMargins && rollEligible(dsrc, dss, sframe, ddst, dds, dframe, W, H, cn, nx) && nx == ny) {
#define ROLL(KS_, CN_) do { if (tuneVariant() <= 1 || border == B_WRAP || W <= KS_ / 2) launchRoll<KS_, CN_>(dsrc, dss, sframe, ddst, dds, dframe, nframes, W, H, border, st); \
                            else launchRoll2<KS_, CN_>(dsrc, dss, sframe, ddst, dds, dframe, nframes, W, H, border, st, tuneVariant() - 2); } while (0)
        if (nx == 5) { switch (cn) { case 1: ROLL(5, 1); break; case 2: ROLL(5, 2); break; case 3: ROLL(5, 3); break; default: ROLL(5, 4); } }
        else         { switch (cn) { case 1: ROLL(3, 1); break; case 2: ROLL(3, 2); break; case 3: ROLL(3, 3); break; default: ROLL(3, 4); } }
#undef ROLL
    } else if (std::getenv("MI355CV_SMOOTH_GENERIC") == nullptr && std::max(nx, ny) >= sepmxMinTaps() && !(std::getenv("MI355CV_SEPMX") && std::getenv("MI355CV_SEPMX")[0] == '0') &&
               sepmxRun(stg, dsrc, dss, sframe, ddst, dds, dframe, nframes, W, H, cn, mL + W + mR, mT + H + mB, mL, mT, border, kx, nx, nx / 2, ky, ny, ny / 2, st)) {
        // 7 taps or more on either axis, any geometry, margins included: both passes on the matrix cores (sepmx.hip) -- taps that fit int8 and span at most five 32-byte K steps
    } else if (noMargins && std::getenv("MI355CV_SMOOTH_GENERIC") == nullptr &&
               seprollFixedSmooth(dsrc, dss, sframe, ddst, dds, dframe, nframes, W, H, cn, kx, nx, ky, ny, border, st)) {
        // any-sigma Q8.8 taps (<= 9) on the rolling skeleton
    } else if (!noMargins && nframes == 1 && std::getenv("MI355CV_SMOOTH_GENERIC") == nullptr && [&] {
                   // a submatrix with real pixels around it (cv_hal_gaussianBlurBinomial's margins): the rolling kernel on the parent's geometry, storing the window
                   const Roi roi = {mL + W + mR, mT + H + mB, mL, mT};
                   return seprollFixedSmooth(dsrc, dss, 0, ddst, dds, 0, 1, W, H, cn, kx, nx, ky, ny, border, st, &roi); }()) {
    } else if (std::getenv("MI355CV_SMOOTH_GENERIC") == nullptr && std::max(nx, ny) < sepmxMinTaps() && !(std::getenv("MI355CV_SEPMX") && std::getenv("MI355CV_SEPMX")[0] == '0') &&
               sepmxRun(stg, dsrc, dss, sframe, ddst, dds, dframe, nframes, W, H, cn, mL + W + mR, mT + H + mB, mL, mT, border, kx, nx, nx / 2, ky, ny, ny / 2, st)) {
        // short kernels the rolling skeleton did not take (two channels, ...)
    } else if (std::getenv("MI355CV_SMOOTH_GENERIC") == nullptr && [&] {
                   // any length, any geometry, margins included: the LDS-ring kernel in its Q8.8 mode -- when no ufixedpoint16 / ufixedpoint32 sum can saturate
                   unsigned sx = 0, sy = 0;
                   for (int i = 0; i < nx; i++) sx += kx[i];
                   for (int i = 0; i < ny; i++) sy += ky[i];
                   if (sx > 256 || sy > 256) return false;
                   std::vector<int> ix(kx, kx + nx), iy(ky, ky + ny);
                   const SepLongTaps t = {nullptr, nullptr, ix.data(), iy.data(), nx, ny, nx / 2, ny / 2, 3, 0, 0.f, 0};
                   return seplongRun(stg, dsrc, dss, sframe, ddst, dds, dframe, nframes, W, H, cn, MI355CV_8U, MI355CV_8U, mL + W + mR, mT + H + mB, mL, mT, border, t, st); }()) {
    } else {
        // taps that can saturate the reference's fixed-point types (never cv::GaussianBlur's own: they sum to 256), or MI355CV_SMOOTH_GENERIC=1: one thread per byte, <= 33 taps
        if (nx > 33 || ny > 33) return mi355::declined(__func__, __LINE__, "Q8.8 taps that sum beyond 256 with more than 33 of them");
        FixedTaps t;
        t.nx = nx; t.ny = ny;
        for (int i = 0; i < 33; i++) { t.kx[i] = i < nx ? kx[i] : 0; t.ky[i] = i < ny ? ky[i] : 0; }
        dim3 grid(divUp(W * cn, 64), divUp(H, 4), nframes);
        hipLaunchKernelGGL(k_sepfixed_generic, grid, dim3(256), 0, st, dsrc, dss, sframe, ddst, dds, dframe,
                           W, H, cn, mL, mT, mR, mB, border, t);
    }
    return stg.finish(entry);
}

// ---- CV_16U, sigma = 0 (cv_hal_gaussianBlurBinomial on the reference's Q16.16 path, smooth.dispatch.cpp:726-768): 3 / 5 / 7 / 9 taps, 1-4 channels, every border rule.
// The reference's fixed-point passes (fixedSmoothInvoker<uint16_t, ufixedpoint32>: exact Q16.16 products, one rounding in the column pass) evaluate to the plain integer
// sums with ONE rounding, (S + 2^(2s-1)) >> 2s for taps that sum to 2^s -- the CPU suite pins that restatement to the reference bit for bit (tests/, the 16-bit smoothing file).  One channel
// with 3 or 5 taps on a geometry the rolling skeleton takes: k_sep_roll<Binom16>; everything else -- more channels, 7 / 9 taps, BORDER_WRAP, images smaller than the
// kernel (round 5: 80 of the 88 hook calls of GaussianBlur_Bitexact.Linear16U and overflow_20121 were declined) -- one thread per element, the K x K sum directly.
template <int K>
__global__ __launch_bounds__(256) void k_binom16_direct(const uchar* __restrict__ parent, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int cn,
                                                        int mL, int mT, int fullW, int fullH, int border)
{
    constexpr int R = K / 2, SH = K == 3 ? 2 : K == 5 ? 4 : K == 7 ? 6 : 8;
    constexpr uint32_t TAP[9] = {K == 3 ? 1u : K == 5 ? 1u : K == 7 ? 2u : 4u, K == 3 ? 2u : K == 5 ? 4u : K == 7 ? 7u : 13u, K == 3 ? 1u : K == 5 ? 6u : K == 7 ? 14u : 30u,
                                 K == 5 ? 4u : K == 7 ? 18u : 51u, K == 5 ? 1u : K == 7 ? 14u : 60u, K == 7 ? 7u : 51u, K == 7 ? 2u : 30u, 13u, 4u};
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, c = e - x * cn;
    int xs[K];
#pragma unroll
    for (int i = 0; i < K; i++) { const int q = mi355_borderInterpolate(x + mL + i - R, fullW, border); xs[i] = q < 0 ? -1 : q * cn + c; }
    uint32_t acc = 0;                                      // <= 65535 * 2^16: fits, and so does the rounding term on top
#pragma unroll
    for (int j = 0; j < K; j++) {
        const int q = mi355_borderInterpolate(y + mT + j - R, fullH, border);
        if (q < 0) continue;
        const unsigned short* row = reinterpret_cast<const unsigned short*>(parent + (size_t)q * sstep);
        uint32_t h = 0;
#pragma unroll
        for (int i = 0; i < K; i++) h += xs[i] < 0 ? 0u : TAP[i] * (uint32_t)row[xs[i]];
        acc += TAP[j] * h;
    }
    reinterpret_cast<unsigned short*>(dst + (size_t)y * dstep)[e] = (unsigned short)((acc + (1u << (2 * SH - 1))) >> (2 * SH));
}

int runBinom16(const char* entry, const uchar* src, size_t sstep, uchar* dst, size_t dstep, int W, int H, int cn, int mL, int mT, int mR, int mB, int ksize, int border)
{
    if (disabled() || W <= 0 || H <= 0 || cn < 1 || cn > 4 || (ksize != 3 && ksize != 5 && ksize != 7 && ksize != 9) || border < 0 || border > B_REFLECT_101 ||
        mL < 0 || mT < 0 || mR < 0 || mB < 0 || ((sstep | dstep | (uintptr_t)src | (uintptr_t)dst) & 1))
        return mi355::declined(__func__, __LINE__, "disabled() || W <= 0 || H <= 0 || cn < 1 || cn > 4 || ksize not in {3, 5, 7, 9} || border outside CONSTANT .. REFLECT_101 || negative margins || odd pointers / steps");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    const bool hostSrc = !isDevicePtr(src);
    if (hostSrc && (size_t)W * H < minPixels()) return mi355::declined(__func__, __LINE__, "hostSrc && (size_t)W * H < minPixels()");
    if (!hostSrc && src == dst) return mi355::declined(__func__, __LINE__, "!hostSrc && src == dst");
    size_t dss = 0, dds = 0;
    const size_t esz = (size_t)cn * 2;
    const uchar* top = src - (ptrdiff_t)mT * (ptrdiff_t)sstep - (ptrdiff_t)mL * (ptrdiff_t)esz;
    const uchar* dtop = stg.in(top, sstep, (size_t)(mL + W + mR) * esz, mT + H + mB, &dss);
    if (!dtop) return mi355::declined(__func__, __LINE__, "!dtop");
    const uchar* dsrc = dtop + (size_t)mT * dss + (size_t)mL * esz;
    uchar* ddst = stg.out(dst, dstep, (size_t)W * esz, H, &dds);
    if (!ddst) return mi355::declined(__func__, __LINE__, "!ddst");
    const Roi roi = {mL + W + mR, mT + H + mB, mL, mT};
    if (cn == 1 && (ksize == 3 || ksize == 5) && border != B_WRAP &&
        seprollBinom16(dsrc, dss, 0, ddst, dds, 0, 1, W, H, ksize, border, stream(), (mL | mT | mR | mB) ? &roi : nullptr))
        return stg.finish(entry);
    dim3 grid(divUp(W * cn, 64), divUp(H, 4));
#define B16(K_) hipLaunchKernelGGL(k_binom16_direct<K_>, grid, dim3(256), 0, stream(), dtop, dss, ddst, dds, W, H, cn, mL, mT, mL + W + mR, mT + H + mB, border)
    if (ksize == 3) B16(3); else if (ksize == 5) B16(5); else if (ksize == 7) B16(7); else B16(9);
#undef B16
    noteKernel("k_binom16_direct<%d> grid=%ux%u x256 cn=%d border=%d", ksize, grid.x, grid.y, cn, border);
    return stg.finish(entry);
}

// sigma==0 Q8.8 tables (smooth.dispatch.cpp:89-145 scaled by 256; cf. test_smooth_bitexact.cpp:14-20)
const uint16_t kBinom1[1] = {256};
const uint16_t kBinom3[3] = {64, 128, 64};
const uint16_t kBinom5[5] = {16, 64, 96, 64, 16};
const uint16_t kBinom7[7] = {8, 28, 56, 72, 56, 28, 8};
const uint16_t kBinom9[9] = {4, 13, 30, 51, 60, 51, 30, 13, 4};

const uint16_t* binomTaps(size_t k)
{
    switch (k) { case 1: return kBinom1; case 3: return kBinom3; case 5: return kBinom5; case 7: return kBinom7; case 9: return kBinom9; }
    return nullptr;
}

} // namespace

extern "C" {

MI355CV_API int mi355cv_gaussianBlurBinomial(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int cn, size_t margin_left, size_t margin_top, size_t margin_right,
        size_t margin_bottom, size_t ksize, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (depth == MI355CV_16U)
        return runBinom16("gaussianBlurBinomial", src_data, src_step, dst_data, dst_step, width, height, cn, (int)margin_left, (int)margin_top, (int)margin_right,
                                 (int)margin_bottom, (int)ksize, border_type & ~MI355CV_BORDER_ISOLATED);
    if (depth != MI355CV_8U) return mi355::declined(__func__, __LINE__, "depth != MI355CV_8U");
    const uint16_t* k = binomTaps(ksize);
    if (!k) return mi355::declined(__func__, __LINE__, "!k");
    return runSmooth("gaussianBlurBinomial", src_data, src_step, 0, dst_data, dst_step, 0, 1, width, height, cn,
                     (int)margin_left, (int)margin_top, (int)margin_right, (int)margin_bottom,
                     k, (int)ksize, k, (int)ksize, border_type & ~MI355CV_BORDER_ISOLATED, true);
}

// A batch that lives in HOST memory (frames that originate on the host): the frames cross PCIe in chunks through two sets of device buffers --
// the upload of chunk i+1 runs on the auxiliary stream while chunk i is filtered and downloaded on the main one, so the two DMA directions overlap and the
// kernel hides under them.  Page-locked host memory (mi355cv_hostAlloc, FrameAllocator::Pinned) is what makes the copies asynchronous; pageable memory
// works, without the overlap.  One 4K CV_8UC1 frame is 8.3 MB each way: the link, not the 3 us kernel, sets the rate.
static int gaussBatchFromHost(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes, int W, int H, int cn,
                              const uint16_t* k, int ks, int border)
{
    const HostBatch hb = {src, sstep, sframe, (size_t)W * cn, H, dst, dstep, dframe, (size_t)W * cn, H, nframes};
    return runHostBatch("gaussianBlurBinomialBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
        return runSmooth("gaussianBlurBinomialBatch", s, ss, nf > 1 ? sf : 0, d, ds, nf > 1 ? df : 0, nf, W, H, cn, 0, 0, 0, 0, k, ks, k, ks, border, true);
    });
}

MI355CV_API int mi355cv_gaussianBlurBinomialBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride,
        uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int nframes,
        int width, int height, int depth, int cn, size_t ksize, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (depth != MI355CV_8U) return mi355::declined(__func__, __LINE__, "depth != MI355CV_8U");
    const uint16_t* k = binomTaps(ksize);
    if (!k) return mi355::declined(__func__, __LINE__, "!k");
    if (nframes > 1 && width > 0 && height > 0 && cn >= 1 && cn <= 4 && hostBatchEligible(src_data, dst_data, nframes))
        return gaussBatchFromHost(src_data, src_step, src_frame_stride, dst_data, dst_step, dst_frame_stride, nframes, width, height, cn, k, (int)ksize,
                                  border_type & ~MI355CV_BORDER_ISOLATED);
    if (nframes == 1)
        return runSmooth("gaussianBlurBinomialBatch", src_data, src_step, 0, dst_data, dst_step, 0, 1, width, height, cn,
                         0, 0, 0, 0, k, (int)ksize, k, (int)ksize, border_type & ~MI355CV_BORDER_ISOLATED, true);
    return runSmooth("gaussianBlurBinomialBatch", src_data, src_step, src_frame_stride, dst_data, dst_step, dst_frame_stride,
                     nframes, width, height, cn, 0, 0, 0, 0, k, (int)ksize, k, (int)ksize,
                     border_type & ~MI355CV_BORDER_ISOLATED, true);
}

MI355CV_API int mi355cv_gaussianBlur(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step,
        int width, int height, int depth, int cn, size_t margin_left, size_t margin_top, size_t margin_right,
        size_t margin_bottom, size_t ksize_width, size_t ksize_height, double sigmaX, double sigmaY, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    // CV_8U: the Q8.8 path cv::GaussianBlur takes for CV_8U (smooth.dispatch.cpp:658-724), below.
    // Other depths: the reference asks this hook (:813) and then calls sepFilter2D(src, dst, sdepth, kx, ky, Point(-1,-1), 0, borderType) with the taps of
    // createGaussianKernels (:279-304: getGaussianKernel(ksize, sigma, max(depth, CV_32F)), ky = kx for a square kernel with equal sigmas) -- :825.  Until round 5 the
    // hook declined and the separable hook served that second call (257 declined calls in the reference's Imgproc_GaussianBlur tests); now the same call is made here.
    if (depth != MI355CV_8U) {
        if (depth != MI355CV_16U && depth != MI355CV_16S && depth != MI355CV_32F) return mi355::declined(__func__, __LINE__, "depth is none of 8U / 16U / 16S / 32F");
        const int n = (int)ksize_width, m = (int)ksize_height;
        if (n < 1 || m < 1 || n > lim::GAUSS_FLOAT_MAX_KSIZE || m > lim::GAUSS_FLOAT_MAX_KSIZE || !(n & 1) || !(m & 1)) return mi355::declined(__func__, __LINE__, "kernel size outside 1 .. lim::GAUSS_FLOAT_MAX_KSIZE or even");
        if (sigmaY <= 0) sigmaY = sigmaX;
        const double s1 = sigmaX > 0 ? sigmaX : 0, s2 = sigmaY > 0 ? sigmaY : 0;
        std::vector<double> dx, dy;
        if (!gaussianKernelBitExact(n, s1, dx)) return mi355::declined(__func__, __LINE__, "!gaussianKernelBitExact(n, s1, dx)");
        if (m == n && std::fabs(s1 - s2) < 2.220446049250313e-16) dy = dx;
        else if (!gaussianKernelBitExact(m, s2, dy)) return mi355::declined(__func__, __LINE__, "!gaussianKernelBitExact(m, s2, dy)");
        std::vector<float> fx(dx.begin(), dx.end()), fy(dy.begin(), dy.end());          // getGaussianKernel(.., CV_32F): the bit-exact doubles rounded to float
        const int type = MI355CV_MAKETYPE(depth, cn);
        cvhalFilter2D* ctx = nullptr;
        int rc = mi355cv_sepFilterInit(&ctx, type, type, MI355CV_32F, reinterpret_cast<uchar*>(fx.data()), n, reinterpret_cast<uchar*>(fy.data()), m, -1, -1, 0.0, border_type);
        if (rc != MI355CV_OK) return rc;
        rc = mi355cv_sepFilter(ctx, const_cast<uchar*>(src_data), src_step, dst_data, dst_step, width, height, (int)(margin_left + (size_t)width + margin_right),
                               (int)(margin_top + (size_t)height + margin_bottom), (int)margin_left, (int)margin_top);
        (void)mi355cv_sepFilterFree(ctx);
        return rc;
    }
    // Real pixels around the ROI mean the caller is the submatrix / non-isolated site (:813): the fixed-point branch (:658) is skipped there and
    // the CPU result is sepFilter2D with float Gaussian taps, which can differ from Q8.8 by 1 LSB.  Decline: the reference then calls
    // sepFilter2D itself, whose hook (mi355cv_sepFilter, ROI offsets included) reproduces that arithmetic.
    if (margin_left | margin_top | margin_right | margin_bottom)
        return mi355::setError(MI355CV_NOT_IMPLEMENTED, "gaussianBlur: submatrix with real margins is the reference's sepFilter2D case");
    if (ksize_width > (size_t)lim::GAUSS8U_MAX_KSIZE || ksize_height > (size_t)lim::GAUSS8U_MAX_KSIZE) return mi355::declined(__func__, __LINE__, "ksize_width or ksize_height > lim::GAUSS8U_MAX_KSIZE");
    if (sigmaY <= 0) sigmaY = sigmaX;
    std::vector<int64_t> qx, qy;
    if (!gaussianKernelFixedQ((int)ksize_width, sigmaX > 0 ? sigmaX : 0, 8, qx)) return mi355::declined(__func__, __LINE__, "!gaussianKernelFixedQ((int)ksize_width, sigmaX > 0 ? sigmaX : 0, 8, qx)");
    if (!gaussianKernelFixedQ((int)ksize_height, sigmaY > 0 ? sigmaY : 0, 8, qy)) return mi355::declined(__func__, __LINE__, "!gaussianKernelFixedQ((int)ksize_height, sigmaY > 0 ? sigmaY : 0, 8, qy)");
    uint16_t kx[lim::GAUSS8U_MAX_KSIZE], ky[lim::GAUSS8U_MAX_KSIZE];
    for (size_t i = 0; i < ksize_width; i++) { if (qx[i] < 0 || qx[i] > 65535) return mi355::declined(__func__, __LINE__, "qx[i] < 0 || qx[i] > 65535"); kx[i] = (uint16_t)qx[i]; }
    for (size_t i = 0; i < ksize_height; i++) { if (qy[i] < 0 || qy[i] > 65535) return mi355::declined(__func__, __LINE__, "qy[i] < 0 || qy[i] > 65535"); ky[i] = (uint16_t)qy[i]; }
    bool binom = ksize_width == ksize_height && (ksize_width == 3 || ksize_width == 5);
    if (binom) {
        const uint16_t* b = binomTaps(ksize_width);
        for (size_t i = 0; i < ksize_width; i++) if (kx[i] != b[i] || ky[i] != b[i]) binom = false;
    }
    return runSmooth("gaussianBlur", src_data, src_step, 0, dst_data, dst_step, 0, 1, width, height, cn,
                     (int)margin_left, (int)margin_top, (int)margin_right, (int)margin_bottom,
                     kx, (int)ksize_width, ky, (int)ksize_height, border_type & ~MI355CV_BORDER_ISOLATED, binom);
}

// cv::GaussianBlur on CV_8U with any sigma over a batch of device-resident whole frames (no reference counterpart: SURVEY section 8e, frames are independent units): the
// Q8.8 taps of getGaussianKernelFixedPoint_ED once, then ONE launch with the frames along the grid -- the register-rolling kernels up to 9 taps, the LDS-ring kernel beyond
MI355CV_API int mi355cv_gaussianBlurBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride, uchar* dst_data, size_t dst_step, size_t dst_frame_stride,
        int nframes, int width, int height, int depth, int cn, size_t ksize_width, size_t ksize_height, double sigmaX, double sigmaY, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (depth != MI355CV_8U) return mi355::declined(__func__, __LINE__, "depth != MI355CV_8U");
    if (nframes < 1 || !(ksize_width & 1) || !(ksize_height & 1) || ksize_width > (size_t)lim::GAUSS8U_MAX_KSIZE || ksize_height > (size_t)lim::GAUSS8U_MAX_KSIZE)
        return mi355::declined(__func__, __LINE__, "nframes < 1 || even kernel size || ksize > lim::GAUSS8U_MAX_KSIZE");
    if (sigmaY <= 0) sigmaY = sigmaX;
    std::vector<int64_t> qx, qy;
    if (!gaussianKernelFixedQ((int)ksize_width, sigmaX > 0 ? sigmaX : 0, 8, qx) || !gaussianKernelFixedQ((int)ksize_height, sigmaY > 0 ? sigmaY : 0, 8, qy))
        return mi355::declined(__func__, __LINE__, "!gaussianKernelFixedQ");
    uint16_t kx[lim::GAUSS8U_MAX_KSIZE], ky[lim::GAUSS8U_MAX_KSIZE];
    for (size_t i = 0; i < ksize_width; i++) { if (qx[i] < 0 || qx[i] > 65535) return mi355::declined(__func__, __LINE__, "qx[i] < 0 || qx[i] > 65535"); kx[i] = (uint16_t)qx[i]; }
    for (size_t i = 0; i < ksize_height; i++) { if (qy[i] < 0 || qy[i] > 65535) return mi355::declined(__func__, __LINE__, "qy[i] < 0 || qy[i] > 65535"); ky[i] = (uint16_t)qy[i]; }
    bool binom = ksize_width == ksize_height && (ksize_width == 3 || ksize_width == 5);
    if (binom) {
        const uint16_t* b = binomTaps(ksize_width);
        for (size_t i = 0; i < ksize_width; i++) if (kx[i] != b[i] || ky[i] != b[i]) binom = false;
    }
    return runSmooth("gaussianBlurBatch", src_data, src_step, nframes > 1 ? src_frame_stride : 0, dst_data, dst_step, nframes > 1 ? dst_frame_stride : 0, nframes, width, height, cn,
                     0, 0, 0, 0, kx, (int)ksize_width, ky, (int)ksize_height, border_type & ~MI355CV_BORDER_ISOLATED, binom);
}

// tuning knobs for experiments (tools/tune_gauss.py): "gauss_seg" rows per work item (0 = heuristic),
// "gauss_variant" 1|2|3
MI355CV_API int mi355cv_setParam(const char* key, int value)
{
    mi355::EntryGuard entry_(__func__);
    if (!key) return -1;
    if (!strcmp(key, "gauss_seg")) { tuneSeg() = value; return 0; }
    if (!strcmp(key, "gauss_variant")) { tuneVariant() = value; return 0; }
    if (!strcmp(key, "gauss_alt")) { tuneAlt() = value; return 0; }
    if (!strcmp(key, "gauss_launch_waves")) { tuneLaunchWaves() = value; return 0; }
    return -1;
}

// streaming-copy probe: copies `bytes` (multiple of 16) device->device with 16 B/lane accesses
MI355CV_API int mi355cv_copyProbe(const void* src, void* dst, size_t bytes, int perThread, int nt)
{
    mi355::EntryGuard entry_(__func__);
    if (!ensureDevice() || (bytes & 15) || perThread < 1) return mi355::declined(__func__, __LINE__, "!ensureDevice() || (bytes & 15) || perThread < 1");
    size_t n16 = bytes / 16;
    size_t blocks = (n16 + (size_t)256 * perThread - 1) / ((size_t)256 * perThread);
    if (nt) hipLaunchKernelGGL((k_copy16<true>), dim3((unsigned)blocks), dim3(256), 0, stream(), (const uint4*)src, (uint4*)dst, n16, perThread);
    else    hipLaunchKernelGGL((k_copy16<false>), dim3((unsigned)blocks), dim3(256), 0, stream(), (const uint4*)src, (uint4*)dst, n16, perThread);
    if (hipGetLastError() != hipSuccess) return MI355CV_ERROR_UNKNOWN;
    if (!asyncMode()) (void)hipStreamSynchronize(stream());
    return MI355CV_OK;
}

MI355CV_API int mi355cv_copyProbeColwalk(const void* src, void* dst, int W, int H, int nframes, int segRows, int unroll)
{
    mi355::EntryGuard entry_(__func__);
    if (!ensureDevice() || (W & 15)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || (W & 15)");
    const int nchunks = W / 16, nstrips = divUp(nchunks, 64), nseg = divUp(H, segRows);
    const long long items = (long long)nstrips * nseg * nframes;
    dim3 grid((unsigned)((items + 3) / 4));
    const uchar* s = (const uchar*)src; uchar* d = (uchar*)dst;
    if (unroll == 1) hipLaunchKernelGGL((k_copy_colwalk<1>), grid, dim3(256), 0, stream(), s, d, (size_t)W, (size_t)W * H, H, nchunks, nstrips, segRows, nseg, nframes);
    else if (unroll == 2) hipLaunchKernelGGL((k_copy_colwalk<2>), grid, dim3(256), 0, stream(), s, d, (size_t)W, (size_t)W * H, H, nchunks, nstrips, segRows, nseg, nframes);
    else if (unroll == 5) hipLaunchKernelGGL((k_copy_colwalk<5>), grid, dim3(256), 0, stream(), s, d, (size_t)W, (size_t)W * H, H, nchunks, nstrips, segRows, nseg, nframes);
    else hipLaunchKernelGGL((k_copy_colwalk<10>), grid, dim3(256), 0, stream(), s, d, (size_t)W, (size_t)W * H, H, nchunks, nstrips, segRows, nseg, nframes);
    if (hipGetLastError() != hipSuccess) return MI355CV_ERROR_UNKNOWN;
    if (!asyncMode()) (void)hipStreamSynchronize(stream());
    return MI355CV_OK;
}

// host-side tap generator, exported so bindings can show / test the exact Q8.8 kernel in use
MI355CV_API int mi355cv_getGaussianKernelQ(int n, double sigma, int fractionBits, int64_t* taps)
{
    mi355::EntryGuard entry_(__func__);
    std::vector<int64_t> q;
    if (!gaussianKernelFixedQ(n, sigma, fractionBits, q)) return mi355::declined(__func__, __LINE__, "!gaussianKernelFixedQ(n, sigma, fractionBits, q)");
    for (int i = 0; i < n; i++) taps[i] = q[i];
    return MI355CV_OK;
}

MI355CV_API int mi355cv_getGaussianKernel(int n, double sigma, double* taps)
{
    mi355::EntryGuard entry_(__func__);
    std::vector<double> k;
    if (!gaussianKernelBitExact(n, sigma, k)) return mi355::declined(__func__, __LINE__, "!gaussianKernelBitExact(n, sigma, k)");
    for (int i = 0; i < n; i++) taps[i] = k[i];
    return MI355CV_OK;
}

MI355CV_API int mi355cv_sepSmoothFixedU8(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step,
        int width, int height, int cn, size_t margin_left, size_t margin_top, size_t margin_right, size_t margin_bottom,
        const uint16_t* kx, int kxlen, const uint16_t* ky, int kylen, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (!kx || !ky) return mi355::declined(__func__, __LINE__, "!kx || !ky");
    bool binom = kxlen == kylen && (kxlen == 3 || kxlen == 5);
    if (binom) {
        const uint16_t* b = binomTaps(kxlen);
        for (int i = 0; i < kxlen; i++) if (kx[i] != b[i] || ky[i] != b[i]) binom = false;
    }
    return runSmooth("sepSmoothFixedU8", src_data, src_step, 0, dst_data, dst_step, 0, 1, width, height, cn,
                     (int)margin_left, (int)margin_top, (int)margin_right, (int)margin_bottom,
                     kx, kxlen, ky, kylen, border_type & ~MI355CV_BORDER_ISOLATED, binom);
}

} // extern "C"
