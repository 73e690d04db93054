// seproll.hip -- register-rolling kernels for the separable 8-bit filters that are not the sigma-0 binomial Gaussian
// (smooth.hip has its own tuned copy of the skeleton): Q8.8 Gaussian with arbitrary taps, normalised box filter, integer
// derivative filters.  All are instances of one kernel: a wave walks a segment of rows (roll.h); per new source row a
// horizontal pass produces that row's intermediates (kept for KY rows in registers, packed two 16-bit values per VGPR on
// even/odd byte planes), a vertical pass over the ring produces the output row.  HBM-bound by design: every source byte is
// fetched once per segment, outputs are written once with non-temporal stores.
#include "seproll.h"
#include "roll.h"

using namespace mi355;

namespace {

// Policy interface:
//   KX, KY, CN, CB        taps, channels, source bytes per lane (16, or 8 when the outputs are 32-bit: a lane then writes
//                         32 contiguous bytes);  OUTB = bytes per output element
//   Args                  kernel parameters (by value)
//   pre(m, side, args)                        applied to the raw dwords of a row right after the load (no-op for most)
//   hpass(Inter&, E, O, args)                 one row's intermediates from the byte planes of its window
//   vpass<UP>(ring, u, args, out[MD*OUTB])    output dwords of the lane from the KY ring rows; ring[(u + j) % KY] is the
//                                             j-th row in WALKING order (image order reversed when UP)
template <class P, bool UP>
__device__ __forceinline__ void sepRows(roll::Ctx<P::KX / 2, P::KY / 2, P::CN, P::CB>& cx, uchar* __restrict__ dst, size_t dstep, const typename P::Args& a)
{
    typedef roll::Ctx<P::KX / 2, P::KY / 2, P::CN, P::CB> Cx;
    typedef typename Cx::RawT RawT;
    constexpr int KY = P::KY, RY = KY / 2, NW = Cx::NW, MD = Cx::MD, OD = MD * P::OUTB;
    typename P::Inter ring[KY];
    auto hrow = [&](typename P::Inter& o, RawT raw, int valid) {
        P::pre(raw.m, raw.side, a);                    // (morphology: erode = complement o dilate o complement)
        if (!valid) {                                  // BORDER_CONSTANT row: zeros (window() only moves bytes)
#pragma unroll
            for (int d = 0; d < MD; d++) raw.m[d] = 0;
#pragma unroll
            for (int d = 0; d < Cx::HD; d++) raw.side[d] = 0;
        }
        uint32_t X[NW];
        cx.window(X, raw);
        if constexpr (P::RAWX) P::hpassX(o, X, a);      // 32-bit elements: the window's dwords are the elements
        else {
            uint32_t E[NW], O[NW];
            roll::planes<NW>(E, O, X);
            P::hpass(o, E, O, a);
        }
    };
#pragma unroll
    for (int i = 0; i < KY - 1; i++) { RawT pre; int v; cx.issue(pre, i - RY, v); hrow(ring[i], pre, v); }
    RawT raw[KY]; int rv[KY];
#pragma unroll
    for (int u = 0; u < KY; u++) cx.issue(raw[u], u + RY, rv[u]);
    for (int y = 0; y < cx.nrows; y += KY) {
#pragma unroll
        for (int u = 0; u < KY; u++) {
            if (y + u < cx.nrows) {
                hrow(ring[(KY - 1 + u) % KY], raw[u], rv[u]);
                cx.issue(raw[u], y + u + KY + RY, rv[u]);
                uint32_t o[OD];
                P::template vpass<UP>(ring, u, a, o);
                cx.template store<P::OUTB>(dst, dstep, cx.gy(y + u), o);
            }
        }
    }
}

template <class P>
__global__ __launch_bounds__(256) void k_sep_roll(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                  int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes, int border, int alt,
                                                  roll::Win win, typename P::Args a)
{
    typedef roll::Ctx<P::KX / 2, P::KY / 2, P::CN, P::CB> Cx;
    Cx cx;
    if (!cx.init(src, sstep, sframe, W, H, nchunks, nstrips, segRows, nseg, nframes, border, alt, win)) return;
    if constexpr ((P::CB / 4) * P::OUTB > 4) {           // a lane owns more than 16 output bytes of a row: roll.h transposes the wave's row piece through LDS into 1 KiB stores
        __shared__ __attribute__((aligned(16))) uchar tscratch[4 * Cx::template tldsBytesPerWave<P::OUTB>()];
        cx.useLds(tscratch, Cx::template tldsBytesPerWave<P::OUTB>());
    }
    dst += (size_t)cx.frame * dframe;
    if (cx.up) sepRows<P, true>(cx, dst, dstep, a);
    else       sepRows<P, false>(cx, dst, dstep, a);
}

// `roi`: the image (src, W, H) is a window of a larger one whose pixels around it are real (the HAL's offset / full-size contract): the kernel then
// runs on the PARENT's geometry and stores the window only (roll.h Win); `bpp` = bytes per pixel of the source as the policy counts channels
template <class P>
void launchSep(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes, int W, int H, int border,
               int bestSeg, const typename P::Args& a, hipStream_t st, const Roi* roi = nullptr)
{
    roll::Win win = roll::wholeImage();
    if (roi) {
        win.x0b = roi->offX * P::CN; win.x1b = (roi->offX + W) * P::CN; win.y0 = roi->offY; win.y1 = roi->offY + H;
        src -= (size_t)roi->offY * sstep + (size_t)roi->offX * P::CN;
        W = roi->fullW; H = roi->fullH;
    }
    const roll::Geom g = roll::geometry(W, H, P::CN, nframes, bestSeg, P::KY, P::CB, 2048, win);
    noteKernel("k_sep_roll<%s K=%d bpp=%d> blocks=%u seg=%d rows%s", P::name(), P::KY, P::CN, g.blocks, g.seg, roi ? " window of a larger image" : "");
    hipLaunchKernelGGL((k_sep_roll<P>), dim3(g.blocks), dim3(256), 0, st, src, sstep, sframe, dst, dstep, dframe, W, H, g.nchunks, g.nstrips, g.seg, g.nseg,
                       nframes, border, 1, win, a);
}

// is (W x H at the ROI's offsets inside fullW x fullH) a call the rolling kernels take?  roll::eligible on the parent's geometry; a window needs nframes == 1
inline bool roiEligible(const Roi* roi, int nframes, int W, int H)
{
    return !roi || (nframes == 1 && roi->offX >= 0 && roi->offY >= 0 && roi->offX + W <= roi->fullW && roi->offY + H <= roi->fullH);
}

// take byte 2 of four 32-bit accumulators -> one dword (accumulators hold value << 16 with value <= 255)
__device__ __forceinline__ uint32_t packB2(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
{
    const uint32_t lo = __builtin_amdgcn_perm(b1, b0, 0x0c0c0602u);      // (b0.byte2, b1.byte2, 0, 0)
    const uint32_t hi = __builtin_amdgcn_perm(b3, b2, 0x06020c0cu);      // (0, 0, b2.byte2, b3.byte2)
    return lo | hi;
}

// ---------------------------------------------------------------------------------- Q8.8 smoothing (fixedSmoothInvoker)
// h = sum kx[i] * p (u16, exact because sum kx <= 256), acc = sum ky[j] * h_j (u32), dst = (acc + 2^15) >> 16
// (smooth.simd.hpp:1926-2170, fixedpoint.inl.hpp:247-345).  Horizontal: one v_mad_u32_u24 per tap and PAIR of pixels (the two
// 16-bit lanes cannot carry into each other); vertical: one v_dot2_u32_u16 per tap and pixel with (ky, 0) / (0, ky) operands.
template <int K, int CN_>
struct FixedSmooth {
    static const char* name() { return "FixedSmooth"; }
    static constexpr int KX = K, KY = K, CN = CN_, CB = 16, OUTB = 1, R = K / 2;
    static constexpr bool RAWX = false;
    static constexpr int HD = roll::Cfg<R, CN>::HD;
    struct Args { uint32_t kx[K]; uint32_t kyLo[K], kyHi[K]; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { uint32_t e[4], o[4]; };
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    template <int Q, int I>
    static __device__ __forceinline__ uint32_t hsum(const uint32_t* E, const uint32_t* O, int k, const Args& a)
    {
        const uint32_t v = roll::pairAt<Q, (I - R) * CN, HD>(E, O, k);
        if constexpr (I == 0) return __umul24(v, a.kx[0]);
        else {
            // the compiler splits a mul24 chain into v_mul_u32_u24 + v_add3_u32; keep it one v_mad_u32_u24 per tap
            const uint32_t acc = hsum<Q, I - 1>(E, O, k, a);
            uint32_t d;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(v), "s"(a.kx[I]), "v"(acc));
            return d;
        }
    }
    static __device__ __forceinline__ void hpass(Inter& o, const uint32_t* E, const uint32_t* O, const Args& a)
    {
#pragma unroll
        for (int k = 0; k < 4; k++) { o.e[k] = hsum<0, K - 1>(E, O, k, a); o.o[k] = hsum<1, K - 1>(E, O, k, a); }
    }
    static __device__ __forceinline__ uint32_t dot(uint32_t h, uint32_t k, uint32_t acc)
    {
        return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, h), __builtin_bit_cast(u16x2, k), acc, false);
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int u, const Args& a, uint32_t (&out)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t el = 0x8000u, eh = 0x8000u, ol = 0x8000u, oh = 0x8000u;
#pragma unroll
            for (int j = 0; j < K; j++) {
                const Inter& r = ring[(u + j) % K];
                const int t = UP ? K - 1 - j : j;
                el = dot(r.e[k], a.kyLo[t], el); eh = dot(r.e[k], a.kyHi[t], eh);
                ol = dot(r.o[k], a.kyLo[t], ol); oh = dot(r.o[k], a.kyHi[t], oh);
            }
            out[k] = packB2(el, ol, eh, oh);            // bytes 4k, 4k+1, 4k+2, 4k+3
        }
    }
};


// ---------------------------------------------------------------------------------- sepFilter2D 8U -> 8U, "bit-exact" integer taps
// cv::sepFilter2D with smooth symmetric kernels whose taps are multiples of 1/256 (createSeparableLinearFilter's fixed-point branch,
// filter.dispatch.cpp:305-420): integer row sums (taps * 2^8), and a column pass that the reference's AVX2 build evaluates in FLOAT for
// every element its 16-lane loop reaches (SymmColumnVec_32s8u filter.simd.hpp:1011-1085): s = fma(S_c, k_c 2^-16, delta),
// s = fma(S_{c+k} + S_{c-k}, k_{c+k} 2^-16, s), rounded half-even and saturated.  Rows whose length is a multiple of 16 elements have no
// scalar tail, which is the only case this policy is launched for.  Row sums are the packed u16 pairs of FixedSmooth (taps >= 0, sum <= 256).
template <int K, int CN_>
struct SepFix8U {
    static const char* name() { return "SepFix8U"; }
    static constexpr int KX = K, KY = K, CN = CN_, CB = 16, OUTB = 1, R = K / 2;
    static constexpr bool RAWX = false;
    static constexpr int HD = roll::Cfg<R, CN>::HD;
    struct Args { uint32_t kx[K]; float ky[K]; float delta; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { uint32_t e[4], o[4]; };
    template <int Q, int I>
    static __device__ __forceinline__ uint32_t hsum(const uint32_t* E, const uint32_t* O, int k, const Args& a)
    {
        const uint32_t v = roll::pairAt<Q, (I - R) * CN, HD>(E, O, k);
        if constexpr (I == 0) return __umul24(v, a.kx[0]);
        else {
            const uint32_t acc = hsum<Q, I - 1>(E, O, k, a);
            uint32_t d;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(v), "s"(a.kx[I]), "v"(acc));
            return d;
        }
    }
    static __device__ __forceinline__ void hpass(Inter& o, const uint32_t* E, const uint32_t* O, const Args& a)
    {
#pragma unroll
        for (int k = 0; k < 4; k++) { o.e[k] = hsum<0, K - 1>(E, O, k, a); o.o[k] = hsum<1, K - 1>(E, O, k, a); }
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int u, const Args& a, uint32_t (&out)[4])
    {
        auto row = [&](int t) -> const Inter& { return ring[(u + (UP ? K - 1 - t : t)) % K]; };     // image row t of the window
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t o4 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {                    // output bytes 4k+q = (E.lo, O.lo, E.hi, O.hi)
                auto at = [&](int t) -> uint32_t { const uint32_t w = (q & 1) ? row(t).o[k] : row(t).e[k]; return (q & 2) ? (w >> 16) : (w & 0xffffu); };
                float sF = __builtin_fmaf((float)at(R), a.ky[R], a.delta);
#pragma unroll
                for (int d = 1; d <= R; d++) sF = __builtin_fmaf((float)(at(R + d) + at(R - d)), a.ky[R + d], sF);
                o4 = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(sF), q, o4);
            }
            out[k] = o4;
        }
    }
};

// take byte 3 of four 32-bit values -> one dword
__device__ __forceinline__ uint32_t packB3(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
{
    const uint32_t lo = __builtin_amdgcn_perm(b1, b0, 0x0c0c0703u);
    const uint32_t hi = __builtin_amdgcn_perm(b3, b2, 0x07030c0cu);
    return lo | hi;
}

// ---------------------------------------------------------------------------------- normalised box filter, u16 sums
// RowSum<uchar,ushort> + ColumnSum<ushort,uchar> (box_filter.simd.hpp:429-560): s = sum of the K*K bytes (<= 65280),
// dst = ((s + dd) * ds) >> 23.  Sums run on the packed planes (two pixels per add); the division is one v_mad_u32_u24 per
// pixel with 2*ds, so that the quotient lands in byte 3: ((s + dd) * ds) >> 23 == (s * 2ds + 2*dd*ds) >> 24, all < 2^32.
template <int K, int CN_>
struct BoxU8 {
    static const char* name() { return "BoxU8"; }
    static constexpr int KX = K, KY = K, CN = CN_, CB = 16, OUTB = 1, R = K / 2;
    static constexpr bool RAWX = false;
    static constexpr int HD = roll::Cfg<R, CN>::HD;
    struct Args { uint32_t ds2, c2; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { uint32_t e[4], o[4]; };
    template <int Q, int I>
    static __device__ __forceinline__ uint32_t hsum(const uint32_t* E, const uint32_t* O, int k)
    {
        const uint32_t v = roll::pairAt<Q, (I - R) * CN, HD>(E, O, k);
        if constexpr (I == 0) return v; else return v + hsum<Q, I - 1>(E, O, k);
    }
    static __device__ __forceinline__ void hpass(Inter& o, const uint32_t* E, const uint32_t* O, const Args&)
    {
#pragma unroll
        for (int k = 0; k < 4; k++) { o.e[k] = hsum<0, K - 1>(E, O, k); o.o[k] = hsum<1, K - 1>(E, O, k); }
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int, const Args& a, uint32_t (&out)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t se = ring[0].e[k], so = ring[0].o[k];
#pragma unroll
            for (int j = 1; j < K; j++) { se += ring[j].e[k]; so += ring[j].o[k]; }
            const uint32_t el = __umul24(se & 0xffffu, a.ds2) + a.c2, eh = __umul24(se >> 16, a.ds2) + a.c2;
            const uint32_t ol = __umul24(so & 0xffffu, a.ds2) + a.c2, oh = __umul24(so >> 16, a.ds2) + a.c2;
            out[k] = packB3(el, ol, eh, oh);
        }
    }
};

// ---------------------------------------------------------------------------------- integer derivative filters, u8 -> s16
// cv::Sobel / cv::Scharr with CV_16S output, scale 1, delta 0: exact integer arithmetic in the reference (SymmRowSmallVec_8u32s
// filter.simd.hpp:530-860, SymmColumnSmallVec_32s16s :1420-1600), result within int16 by the host's range check, so the two
// passes run as packed 16-bit multiply-adds (v_pk_mad_i16 / v_pk_mul_lo_u16) on the byte planes.
template <int K, int CN_>
struct Deriv16 {
    static const char* name() { return "Deriv16"; }
    static constexpr int KX = K, KY = K, CN = CN_, CB = 16, OUTB = 2, R = K / 2;
    static constexpr bool RAWX = false;
    static constexpr int HD = roll::Cfg<R, CN>::HD;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    struct Args { uint32_t kx[K], ky[K]; };              // taps splatted into both 16-bit halves
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { s16x2 e[4], o[4]; };
    template <int Q, int I>
    static __device__ __forceinline__ s16x2 hsum(const uint32_t* E, const uint32_t* O, int k, const Args& a)
    {
        const s16x2 v = __builtin_bit_cast(s16x2, roll::pairAt<Q, (I - R) * CN, HD>(E, O, k));
        const s16x2 t = __builtin_bit_cast(s16x2, a.kx[I]);
        if constexpr (I == 0) return v * t; else return v * t + hsum<Q, I - 1>(E, O, k, a);
    }
    static __device__ __forceinline__ void hpass(Inter& o, const uint32_t* E, const uint32_t* O, const Args& a)
    {
#pragma unroll
        for (int k = 0; k < 4; k++) { o.e[k] = hsum<0, K - 1>(E, O, k, a); o.o[k] = hsum<1, K - 1>(E, O, k, a); }
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int u, const Args& a, uint32_t (&out)[8])
    {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s16x2 se = {0, 0}, so = {0, 0};
#pragma unroll
            for (int j = 0; j < K; j++) {
                const Inter& r = ring[(u + j) % K];
                const s16x2 t = __builtin_bit_cast(s16x2, a.ky[UP ? K - 1 - j : j]);
                se = r.e[k] * t + se; so = r.o[k] * t + so;
            }
            const uint32_t ue = __builtin_bit_cast(uint32_t, se), uo = __builtin_bit_cast(uint32_t, so);
            out[2 * k] = __builtin_amdgcn_perm(uo, ue, 0x05040100u);          // pixels 4k, 4k+1
            out[2 * k + 1] = __builtin_amdgcn_perm(uo, ue, 0x07060302u);      // pixels 4k+2, 4k+3
        }
    }
};

// ---------------------------------------------------------------------------------- float separable filter, u8 -> f32 / u8
// The float path of cv::sepFilter2D / cv::Sobel (RowFilter filter.simd.hpp:2386: r = k0*v0, r = fma(ki, vi, r);
// SymmColumnFilter :2679-2751 in its symmetric / antisymmetric pair form, ColumnFilter :2640 as a plain chain), in that
// association order.  Values are held as pairs (pixel i, pixel i+8) so that every multiply-add is a v_pk_fma_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int K, int SYM, int OUTB_, int CN_>
struct SepF32 {
    static const char* name() { return "SepF32"; }
    static constexpr int KX = K, KY = K, CN = CN_, OUTB = OUTB_, R = K / 2;
    static constexpr bool RAWX = false;
    static constexpr int CB = OUTB_ == 4 ? 8 : 16;           // 32-bit outputs: 8 elements per lane = 32 contiguous output bytes
    static constexpr int NP = CB / 2;                        // element pairs (i, i + NP)
    static constexpr int HD = roll::Cfg<R, CN, CB>::HD;
    struct Args { float kx[K], ky[K], delta; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { f32x2 h[NP]; };
    static __device__ __forceinline__ void hpass(Inter& o, const uint32_t* E, const uint32_t* O, const Args& a)
    {
        // window byte b (0 = first halo byte of dword 0): even bytes of dword d in E[d] (low half = byte 0, high = byte 2)
        auto byteAt = [&](int b) -> float {
            const uint32_t w = (b & 1) ? O[b >> 2] : E[b >> 2];
            return (float)((b & 2) ? (w >> 16) : (w & 0xffffu));
        };
        constexpr int HB = R * CN;
        f32x2 Q[NP + 2 * HB];
#pragma unroll
        for (int m = 0; m < NP + 2 * HB; m++) { Q[m].x = byteAt(4 * HD - HB + m); Q[m].y = byteAt(4 * HD - HB + m + NP); }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            f32x2 r = Q[i] * f32x2{a.kx[0], a.kx[0]};
#pragma unroll
            for (int t = 1; t < K; t++) r = __builtin_elementwise_fma(f32x2{a.kx[t], a.kx[t]}, Q[i + t * CN], r);
            o.h[i] = r;
        }
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int u, const Args& a, uint32_t (&out)[(CB / 4) * OUTB_])
    {
        auto row = [&](int t) -> const Inter& { return ring[(u + (UP ? K - 1 - t : t)) % K]; };     // image row t of the window
        f32x2 o[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const f32x2 d2 = {a.delta, a.delta};
            f32x2 s;
            if (SYM == 1 || SYM == 2) {
                s = SYM == 1 ? __builtin_elementwise_fma(f32x2{a.ky[R], a.ky[R]}, row(R).h[i], d2) : d2;
#pragma unroll
                for (int k = 1; k <= R; k++) {
                    const f32x2 p = SYM == 1 ? row(R + k).h[i] + row(R - k).h[i] : row(R + k).h[i] - row(R - k).h[i];
                    s = __builtin_elementwise_fma(f32x2{a.ky[R + k], a.ky[R + k]}, p, s);
                }
            } else {
                s = __builtin_elementwise_fma(f32x2{a.ky[0], a.ky[0]}, row(0).h[i], d2);
#pragma unroll
                for (int j = 1; j < K; j++) s = __builtin_elementwise_fma(f32x2{a.ky[j], a.ky[j]}, row(j).h[i], s);
            }
            o[i] = s;
        }
        if (OUTB_ == 4) {
#pragma unroll
            for (int i = 0; i < NP; i++) { out[i] = __float_as_uint(o[i].x); out[NP + i] = __float_as_uint(o[i].y); }
        } else {
#pragma unroll
            for (int q = 0; q < CB / 4; q++) out[q] = 0;
#pragma unroll
            for (int i = 0; i < NP; i++) {            // cvRound + saturate_cast<uchar>
                out[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(o[i].x), i & 3, out[i >> 2]);
                out[NP / 4 + (i >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(o[i].y), i & 3, out[NP / 4 + (i >> 2)]);
            }
        }
    }
};

// ---------------------------------------------------------------------------------- CV_32F -> CV_32F separable filter
// cv::sepFilter2D / cv::Sobel / cv::GaussianBlur on CV_32FC1 (north_star's second parity class).  The skeleton moves bytes, so a float is handled as a
// pixel of CN = 4 "channels": halos are RX whole floats, border rules act on whole floats, a lane owns four floats (one dwordx4 in, one out), and the
// window's dwords ARE the elements (hpassX).  Arithmetic: the float path of the reference in the association order of its scalar / FMA forms, as
// SepF32 above (RowFilter filter.simd.hpp:2386: r = k0*v0, r = fma(k_i, v_i, r); SymmColumnFilter :2679-2751 pair forms; ColumnFilter :2640 chain).
template <int K, int SYM>
struct SepF32F {
    static const char* name() { return "SepF32F"; }
    static constexpr int KX = K, KY = K, CN = 4, CB = 16, OUTB = 1, R = K / 2;
    static constexpr bool RAWX = true;
    struct Args { float kx[K], ky[K], delta; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { float h[4]; };
    template <int NWn>
    static __device__ __forceinline__ void hpassX(Inter& o, const uint32_t (&X)[NWn], const Args& a)
    {
        static_assert(NWn == 4 + 2 * R, "a float is one halo dword");
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float r = a.kx[0] * __uint_as_float(X[k]);
#pragma unroll
            for (int t = 1; t < K; t++) r = __builtin_fmaf(a.kx[t], __uint_as_float(X[k + t]), r);
            o.h[k] = r;
        }
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int u, const Args& a, uint32_t (&out)[4])
    {
        auto row = [&](int t) -> const Inter& { return ring[(u + (UP ? K - 1 - t : t)) % K]; };     // image row t of the window
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float s;
            if (SYM == 1 || SYM == 2) {
                s = SYM == 1 ? __builtin_fmaf(a.ky[R], row(R).h[k], a.delta) : a.delta;
#pragma unroll
                for (int d = 1; d <= R; d++)
                    s = __builtin_fmaf(a.ky[R + d], SYM == 1 ? row(R + d).h[k] + row(R - d).h[k] : row(R + d).h[k] - row(R - d).h[k], s);
            } else {
                s = __builtin_fmaf(a.ky[0], row(0).h[k], a.delta);
#pragma unroll
                for (int j = 1; j < K; j++) s = __builtin_fmaf(a.ky[j], row(j).h[k], s);
            }
            out[k] = __float_as_uint(s);
        }
    }
};

// ---------------------------------------------------------------------------------- box filter on CV_32FC1: sums in double
// RowSum<float,double> + ColumnSum<double,float> (box_filter.simd.hpp:64-140, :176-260, selected at :1256-1265): the window's K*K floats are summed
// in double -- here directly (K adds per row, K per column) instead of the reference's running sums, which differ from it by double rounding only,
// eleven decimal digits below the float result -- and the result is float(s * scale) with the double scale 1 / (K*K), or float(s) un-normalised.
template <int K>
struct BoxF32 {
    static const char* name() { return "BoxF32"; }
    static constexpr int KX = K, KY = K, CN = 4, CB = 16, OUTB = 1, R = K / 2;
    static constexpr bool RAWX = true;
    struct Args { double scale; int normalize; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { double h[4]; };
    template <int NWn>
    static __device__ __forceinline__ void hpassX(Inter& o, const uint32_t (&X)[NWn], const Args&)
    {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double r = (double)__uint_as_float(X[k]);
#pragma unroll
            for (int t = 1; t < K; t++) r = __dadd_rn(r, (double)__uint_as_float(X[k + t]));
            o.h[k] = r;
        }
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int u, const Args& a, uint32_t (&out)[4])
    {
        auto row = [&](int t) -> const Inter& { return ring[(u + (UP ? K - 1 - t : t)) % K]; };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double s = row(0).h[k];
#pragma unroll
            for (int j = 1; j < K; j++) s = __dadd_rn(s, row(j).h[k]);
            out[k] = __float_as_uint(a.normalize ? (float)__dmul_rn(s, a.scale) : (float)s);
        }
    }
};

// ---------------------------------------------------------------------------------- erode / dilate, rectangular element, u8
// MorphRowFilter / MorphColumnFilter (morph.simd.hpp:590-700): running max over the K x K window.  dilate with the default
// constant border pads with 0, which is what a BORDER_CONSTANT halo is here; erode is computed as ~dilate(~src), so its default
// border (255) is the same zero halo.  Max of packed 16-bit pairs: v_pk_max_u16 on the byte planes.
template <int K, int CN_>
struct MorphMax {
    static const char* name() { return "MorphMax"; }
    static constexpr int KX = K, KY = K, CN = CN_, CB = 16, OUTB = 1, R = K / 2;
    static constexpr bool RAWX = false;
    static constexpr int HD = roll::Cfg<R, CN>::HD;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    struct Args { uint32_t flip; };                      // 0 dilate, 0xffffffff erode
    struct Inter { uint32_t e[4], o[4]; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&m)[MDn], uint32_t (&side)[HDn], const Args& a)
    {
#pragma unroll
        for (int d = 0; d < MDn; d++) m[d] ^= a.flip;
#pragma unroll
        for (int d = 0; d < HDn; d++) side[d] ^= a.flip;
    }
    static __device__ __forceinline__ uint32_t mx(uint32_t a, uint32_t b)
    {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
    }
    template <int Q, int I>
    static __device__ __forceinline__ uint32_t hmax(const uint32_t* E, const uint32_t* O, int k)
    {
        const uint32_t v = roll::pairAt<Q, (I - R) * CN, HD>(E, O, k);
        if constexpr (I == 0) return v; else return mx(v, hmax<Q, I - 1>(E, O, k));
    }
    static __device__ __forceinline__ void hpass(Inter& o, const uint32_t* E, const uint32_t* O, const Args&)
    {
#pragma unroll
        for (int k = 0; k < 4; k++) { o.e[k] = hmax<0, K - 1>(E, O, k); o.o[k] = hmax<1, K - 1>(E, O, k); }
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int, const Args& a, uint32_t (&out)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t se = ring[0].e[k], so = ring[0].o[k];
#pragma unroll
            for (int j = 1; j < K; j++) { se = mx(se, ring[j].e[k]); so = mx(so, ring[j].o[k]); }
            out[k] = (se | (so << 8)) ^ a.flip;            // bytes 4k..4k+3 = (E.lo, O.lo, E.hi, O.hi)
        }
    }
};

} // namespace

namespace mi355 {

bool seprollFixedSmooth(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                        int W, int H, int cn, const uint16_t* kx, int nx, const uint16_t* ky, int ny, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if (nx != ny || (nx != 3 && nx != 5 && nx != 7 && nx != 9) || !(cn == 1 || cn == 3 || cn == 4)) return false;
    unsigned sx = 0, sy = 0;
    for (int i = 0; i < nx; i++) { sx += kx[i]; sy += ky[i]; }
    if (sx > 256 || sy > 256) return false;
    if (!roll::eligible(src, sstep, sframe, dst, dstep, dframe, roi ? roi->fullW : W, cn, nx / 2, border)) return false;
#define FS(K_, CN_) do { typedef FixedSmooth<K_, CN_> P; P::Args a; \
        for (int i = 0; i < K_; i++) { a.kx[i] = kx[i]; a.kyLo[i] = ky[i]; a.kyHi[i] = (uint32_t)ky[i] << 16; } \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, 16, a, st, roi); } while (0)
#define FSK(K_) do { if (cn == 1) FS(K_, 1); else if (cn == 3) FS(K_, 3); else FS(K_, 4); } while (0)
    switch (nx) { case 3: FSK(3); break; case 5: FSK(5); break; case 7: FSK(7); break; default: FSK(9); }
#undef FSK
#undef FS
    return true;
}

bool seprollFix8U(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                  int W, int H, int cn, const int* kx, const int* ky, int n, float delta, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if ((n != 3 && n != 5) || !(cn == 1 || cn == 3 || cn == 4) || ((size_t)W * cn) % 16 != 0) return false;
    int sx = 0;
    for (int i = 0; i < n; i++) { if (kx[i] < 0) return false; sx += kx[i]; }
    if (sx > 256) return false;
    for (int i = 0; i < n / 2; i++) if (ky[i] != ky[n - 1 - i]) return false;              // the pair form needs a symmetric column kernel
    if (!roll::eligible(src, sstep, sframe, dst, dstep, dframe, roi ? roi->fullW : W, cn, n / 2, border)) return false;
#define FX(K_, CN_) do { typedef SepFix8U<K_, CN_> P; P::Args a; \
        for (int i = 0; i < K_; i++) { a.kx[i] = (uint32_t)kx[i]; a.ky[i] = (float)ky[i] * (1.0f / 65536.0f); } a.delta = delta; \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, 16, a, st, roi); } while (0)
#define FXK(K_) do { if (cn == 1) FX(K_, 1); else if (cn == 3) FX(K_, 3); else FX(K_, 4); } while (0)
    if (n == 3) FXK(3); else FXK(5);
#undef FXK
#undef FX
    return true;
}

bool seprollBox(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                int W, int H, int cn, int ksize, unsigned divScale, unsigned divDelta, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if ((ksize != 3 && ksize != 5 && ksize != 7) || !(cn == 1 || cn == 3 || cn == 4)) return false;
    if (divScale >= (1u << 22) || (unsigned long long)divDelta * divScale >= (1ull << 30)) return false;
    if (!roll::eligible(src, sstep, sframe, dst, dstep, dframe, roi ? roi->fullW : W, cn, ksize / 2, border)) return false;
#define BX(K_, CN_) do { typedef BoxU8<K_, CN_> P; P::Args a = {2u * divScale, 2u * divDelta * divScale}; \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, 16, a, st, roi); } while (0)
#define BXK(K_) do { if (cn == 1) BX(K_, 1); else if (cn == 3) BX(K_, 3); else BX(K_, 4); } while (0)
    switch (ksize) { case 3: BXK(3); break; case 5: BXK(5); break; default: BXK(7); }
#undef BXK
#undef BX
    return true;
}

bool seprollDeriv16(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                    int W, int H, int cn, const int* kx, const int* ky, int n, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if ((n != 3 && n != 5) || !(cn == 1 || cn == 3 || cn == 4)) return false;
    long long ax = 0, ay = 0;
    for (int i = 0; i < n; i++) { ax += kx[i] < 0 ? -kx[i] : kx[i]; ay += ky[i] < 0 ? -ky[i] : ky[i]; }
    if (255 * ax > 32767 || 255 * ax * ay > 32767) return false;
    if ((((uintptr_t)dst | dstep | dframe) & 1) != 0 || !roll::eligible(src, sstep, sframe, src, sstep, sframe, roi ? roi->fullW : W, cn, n / 2, border)) return false;
#define DV(K_, CN_) do { typedef Deriv16<K_, CN_> P; P::Args a; \
        for (int i = 0; i < K_; i++) { a.kx[i] = ((uint32_t)kx[i] & 0xffffu) * 0x10001u; a.ky[i] = ((uint32_t)ky[i] & 0xffffu) * 0x10001u; } \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, 16, a, st, roi); } while (0)
#define DVC(K_) do { if (cn == 1) DV(K_, 1); else if (cn == 3) DV(K_, 3); else DV(K_, 4); } while (0)
    if (n == 3) DVC(3); else DVC(5);
#undef DVC
#undef DV
    return true;
}

bool seprollFloat(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                  int W, int H, int cn, const float* kx, const float* ky, int n, int symY, float delta, int outBytes, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if ((n != 3 && n != 5 && n != 7) || (outBytes != 1 && outBytes != 4) || symY < 0 || symY > 2 || !(cn == 1 || cn == 3)) return false;
    if (cn == 3 && outBytes == 4 && n == 5) return false;         // 2 pixels x 3 channels of halo do not fit an 8-byte chunk
    if (n == 7 && (cn != 1 || outBytes != 1 || symY != 1)) return false;   // 7 taps: the symmetric 8-bit form only (cv::GaussianBlur 7 x 7 with a sigma that has no Q8 taps: ORB's blur)
    if ((((uintptr_t)dst | dstep | dframe) & (outBytes - 1)) != 0 || !roll::eligible(src, sstep, sframe, src, sstep, sframe, roi ? roi->fullW : W, cn, n / 2, border, outBytes == 4 ? 8 : 16)) return false;
#define SF1(K_, S_, O_, CN_) do { typedef SepF32<K_, S_, O_, CN_> P; P::Args a; for (int i = 0; i < K_; i++) { a.kx[i] = kx[i]; a.ky[i] = ky[i]; } a.delta = delta; \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, K_ == 3 ? 16 : 12, a, st, roi); } while (0)
#define SF(K_, S_, O_) do { if (cn == 1) SF1(K_, S_, O_, 1); else SF1(K_, S_, O_, 3); } while (0)
#define SFS(K_, O_) do { if (symY == 1) SF(K_, 1, O_); else if (symY == 2) SF(K_, 2, O_); else SF(K_, 0, O_); } while (0)
    if (n == 3) { if (outBytes == 4) SFS(3, 4); else SFS(3, 1); }
    else if (n == 7) SF1(7, 1, 1, 1);
    else if (outBytes == 4) { if (symY == 1) SF1(5, 1, 4, 1); else if (symY == 2) SF1(5, 2, 4, 1); else SF1(5, 0, 4, 1); }
    else SFS(5, 1);
#undef SFS
#undef SF
#undef SF1
    return true;
}

bool seprollF32(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                int W, int H, const float* kx, const float* ky, int n, int symY, float delta, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if ((n != 3 && n != 5 && n != 7) || symY < 0 || symY > 2) return false;
    if ((((uintptr_t)src | sstep | sframe | (uintptr_t)dst | dstep | dframe) & 3) != 0) return false;
    if (!roll::eligible(src, sstep, sframe, dst, dstep, dframe, roi ? roi->fullW : W, 4, n / 2, border)) return false;
#define FF(K_, S_) do { typedef SepF32F<K_, S_> P; P::Args a; for (int i = 0; i < K_; i++) { a.kx[i] = kx[i]; a.ky[i] = ky[i]; } a.delta = delta; \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, K_ == 3 ? 16 : 12, a, st, roi); } while (0)
#define FFS(K_) do { if (symY == 1) FF(K_, 1); else if (symY == 2) FF(K_, 2); else FF(K_, 0); } while (0)
    switch (n) { case 3: FFS(3); break; case 5: FFS(5); break; default: FFS(7); }
#undef FFS
#undef FF
    return true;
}

// ---------------------------------------------------------------------------------- CV_16U / CV_16S -> same depth or CV_32F, float taps
// cv::sepFilter2D / cv::Sobel / cv::Scharr on 16-bit single-channel images (ktype CV_32F: RowFilter<ushort|short, float>, SymmColumnFilter / ColumnFilter with
// Cast<float, ushort|short> = cvRound + saturate, filter.simd.hpp:2386, :2640-2751): a 16-bit element is a pixel of CN = 2 bytes for the skeleton, a lane owns
// eight of them (one dwordx4 in), the window's dwords hold two elements each.  Arithmetic as SepF32F above.
template <int K, int SYM, bool SGN, bool OUTF>
struct SepF16 {
    static const char* name() { return SGN ? (OUTF ? "SepF16<16S->32F>" : "SepF16<16S>") : (OUTF ? "SepF16<16U->32F>" : "SepF16<16U>"); }
    static constexpr int KX = K, KY = K, CN = 2, CB = 16, OUTB = OUTF ? 2 : 1, R = K / 2;
    static constexpr bool RAWX = true;
    static constexpr int HD = roll::Cfg<R, CN, CB>::HD;
    struct Args { float kx[K], ky[K], delta; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { float h[8]; };
    template <int NWn>
    static __device__ __forceinline__ void hpassX(Inter& o, const uint32_t (&X)[NWn], const Args& a)
    {
        static_assert(NWn == 4 + 2 * HD, "window = own four dwords + HD halo dwords per side");
        float e[8 + 2 * R];                                           // elements first own - R .. last own + R
#pragma unroll
        for (int j = 0; j < 8 + 2 * R; j++) {
            const int idx = 2 * HD - R + j;                           // element index inside the window (two per dword)
            const uint32_t w = (idx & 1) ? X[idx >> 1] >> 16 : X[idx >> 1] & 0xffffu;
            e[j] = SGN ? (float)(short)w : (float)w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float r = a.kx[0] * e[k];
#pragma unroll
            for (int t = 1; t < K; t++) r = __builtin_fmaf(a.kx[t], e[k + t], r);
            o.h[k] = r;
        }
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int u, const Args& a, uint32_t (&out)[4 * OUTB])
    {
        auto row = [&](int t) -> const Inter& { return ring[(u + (UP ? K - 1 - t : t)) % K]; };     // image row t of the window
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float s;
            if (SYM == 1 || SYM == 2) {
                s = SYM == 1 ? __builtin_fmaf(a.ky[R], row(R).h[k], a.delta) : a.delta;
#pragma unroll
                for (int d = 1; d <= R; d++)
                    s = __builtin_fmaf(a.ky[R + d], SYM == 1 ? row(R + d).h[k] + row(R - d).h[k] : row(R + d).h[k] - row(R - d).h[k], s);
            } else {
                s = __builtin_fmaf(a.ky[0], row(0).h[k], a.delta);
#pragma unroll
                for (int j = 1; j < K; j++) s = __builtin_fmaf(a.ky[j], row(j).h[k], s);
            }
            v[k] = s;
        }
        if (OUTF) {
#pragma unroll
            for (int k = 0; k < 8; k++) out[k] = __float_as_uint(v[k]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {                             // saturate_cast<ushort|short>(float): cvRound, then clamp (core/saturate.hpp)
                float r0 = __builtin_rintf(v[2 * q]), r1 = __builtin_rintf(v[2 * q + 1]);
                r0 = SGN ? fminf(fmaxf(r0, -32768.f), 32767.f) : fminf(fmaxf(r0, 0.f), 65535.f);
                r1 = SGN ? fminf(fmaxf(r1, -32768.f), 32767.f) : fminf(fmaxf(r1, 0.f), 65535.f);
                out[q] = ((uint32_t)(int)r0 & 0xffffu) | ((uint32_t)(int)r1 << 16);
            }
        }
    }
};

bool seprollF16(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                int W, int H, bool sgn, bool outFloat, const float* kx, const float* ky, int n, int symY, float delta, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if ((n != 3 && n != 5) || symY < 0 || symY > 2) return false;
    if ((((uintptr_t)src | sstep | sframe) & 1) != 0 || (((uintptr_t)dst | dstep | dframe) & 3) != 0) return false;
    if (!roll::eligible(src, sstep, sframe, src, sstep, sframe, roi ? roi->fullW : W, 2, n / 2, border)) return false;
#define FH(K_, S_, G_, O_) do { typedef SepF16<K_, S_, G_, O_> P; P::Args a; for (int i = 0; i < K_; i++) { a.kx[i] = kx[i]; a.ky[i] = ky[i]; } a.delta = delta; \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, K_ == 3 ? 16 : 12, a, st, roi); } while (0)
#define FHO(K_, S_) do { if (sgn) { if (outFloat) FH(K_, S_, true, true); else FH(K_, S_, true, false); } else { if (outFloat) FH(K_, S_, false, true); else FH(K_, S_, false, false); } } while (0)
#define FHS(K_) do { if (symY == 1) FHO(K_, 1); else if (symY == 2) FHO(K_, 2); else FHO(K_, 0); } while (0)
    if (n == 3) FHS(3); else FHS(5);
#undef FHS
#undef FHO
#undef FH
    return true;
}

// ---------------------------------------------------------------------------------- CV_16UC1 sigma = 0 Gaussian 3x3 / 5x5
// cv::GaussianBlur on CV_16U (smooth.dispatch.cpp:726-760 -> cv_hal_gaussianBlurBinomial; GaussianBlurFixedPointImpl<uint32_t, uint16_t, ufixedpoint32>):
// hlineSmooth3N121 / 5N14641 shift every term into Q16.16 exactly ((s << 14) ..., no rounding, no saturation: the taps sum to 1), vlineSmooth3N121 / 5N14641
// round once: (sum + 2^17) >> 18 resp. (sum + 2^19) >> 20 (smooth.simd.hpp:1425-1452, :1596-1632) -- i.e. (S + 8) >> 4 and (S + 128) >> 8 of the plain integer
// binomial sums S, as for CV_8U.  A 16-bit element is a 2-byte pixel of the skeleton, eight per lane.
template <int K>
struct Binom16 {
    static const char* name() { return "Binom16"; }
    static constexpr int KX = K, KY = K, CN = 2, CB = 16, OUTB = 1, R = K / 2;
    static constexpr bool RAWX = true;
    static constexpr int HD = roll::Cfg<R, CN, CB>::HD;
    struct Args { int unused; };
    template <int MDn, int HDn> static __device__ __forceinline__ void pre(uint32_t (&)[MDn], uint32_t (&)[HDn], const Args&) {}
    struct Inter { uint32_t h[8]; };
    template <int NWn>
    static __device__ __forceinline__ void hpassX(Inter& o, const uint32_t (&X)[NWn], const Args&)
    {
        static_assert(NWn == 4 + 2 * HD, "window = own four dwords + HD halo dwords per side");
        uint32_t e[8 + 2 * R];
#pragma unroll
        for (int j = 0; j < 8 + 2 * R; j++) {
            const int idx = 2 * HD - R + j;
            e[j] = (idx & 1) ? X[idx >> 1] >> 16 : X[idx >> 1] & 0xffffu;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            o.h[k] = K == 3 ? e[k] + 2 * e[k + 1] + e[k + 2] : e[k] + e[k + 4] + 4 * (e[k + 1] + e[k + 3]) + 6 * e[k + 2];
    }
    template <bool UP>
    static __device__ __forceinline__ void vpass(const Inter (&ring)[K], int u, const Args&, uint32_t (&out)[4])
    {
        auto row = [&](int t) -> const Inter& { return ring[(u + t) % K]; };               // symmetric taps: the direction of the walk does not matter
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = K == 3 ? (row(0).h[k] + 2 * row(1).h[k] + row(2).h[k] + 8u) >> 4
                          : (row(0).h[k] + row(4).h[k] + 4 * (row(1).h[k] + row(3).h[k]) + 6 * row(2).h[k] + 128u) >> 8;
#pragma unroll
        for (int q = 0; q < 4; q++) out[q] = v[2 * q] | (v[2 * q + 1] << 16);
    }
};

bool seprollBinom16(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                    int W, int H, int ksize, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if (ksize != 3 && ksize != 5) return false;
    if ((((uintptr_t)src | sstep | sframe) & 1) != 0 || (((uintptr_t)dst | dstep | dframe) & 3) != 0) return false;
    if (!roll::eligible(src, sstep, sframe, src, sstep, sframe, roi ? roi->fullW : W, 2, ksize / 2, border)) return false;
    if (ksize == 3) { typedef Binom16<3> P; P::Args a = {0}; launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, 16, a, st, roi); }
    else            { typedef Binom16<5> P; P::Args a = {0}; launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, 12, a, st, roi); }
    return true;
}

bool seprollBoxF32(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                   int W, int H, int ksize, bool normalize, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if (ksize != 3 && ksize != 5 && ksize != 7) return false;
    if ((((uintptr_t)src | sstep | sframe | (uintptr_t)dst | dstep | dframe) & 3) != 0) return false;
    if (!roll::eligible(src, sstep, sframe, dst, dstep, dframe, roi ? roi->fullW : W, 4, ksize / 2, border)) return false;
#define BF(K_) do { typedef BoxF32<K_> P; P::Args a = {1.0 / (double)(K_ * K_), normalize ? 1 : 0}; \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, K_ == 3 ? 16 : 12, a, st, roi); } while (0)
    switch (ksize) { case 3: BF(3); break; case 5: BF(5); break; default: BF(7); }
#undef BF
    return true;
}

bool seprollMorph(int erode, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                  int W, int H, int cn, int ksize, int border, hipStream_t st, const Roi* roi)
{
    if (!roiEligible(roi, nframes, W, H)) return false;
    if ((ksize != 3 && ksize != 5 && ksize != 7) || !(cn == 1 || cn == 3 || cn == 4)) return false;
    if (!roll::eligible(src, sstep, sframe, dst, dstep, dframe, roi ? roi->fullW : W, cn, ksize / 2, border)) return false;
#define MM(K_, CN_) do { typedef MorphMax<K_, CN_> P; P::Args a = {erode ? 0xffffffffu : 0u}; \
        launchSep<P>(src, sstep, sframe, dst, dstep, dframe, nframes, W, H, border, 16, a, st, roi); } while (0)
#define MMK(K_) do { if (cn == 1) MM(K_, 1); else if (cn == 3) MM(K_, 3); else MM(K_, 4); } while (0)
    switch (ksize) { case 3: MMK(3); break; case 5: MMK(5); break; default: MMK(7); }
#undef MMK
#undef MM
    return true;
}

} // namespace mi355
