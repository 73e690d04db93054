// warp8.h -- bilinear warpAffine / warpPerspective on CV_8U sources through an LDS tile (kernel k_warp8_tile, warp.hip).
//
// Why: the thread-per-column kernel (k_warp_lin<uchar>) gathers bytes -- a 64-lane 2-byte gather along a sloped line was served as ~61 vector-L1
// accesses per wave load (profiles/r02_warp_pmc.txt), it stores bytes, and ran at 7-17 % of the HBM roofline.  Here a workgroup owns a 128 x TH
// destination tile: it computes the EXACT bounding box of the tile's source footprint (the reference's fixed-point coordinate sums are monotone in
// x and y, so the box of the four corners bounds every pixel), copies that box from HBM into LDS with whole aligned dwords (coalesced along rows),
// and every lane then produces four horizontally adjacent destination pixels from LDS -- the aligned dwords around a tap pair + a byte funnel shift, the
// bilinear weights evaluated in their exact separable form -- and stores them as ONE dword (12 / 16 bytes for 3 / 4 channels).  Pixels whose 2x2 footprint is not
// strictly inside the source take the generic sampler (borders, BORDER_TRANSPARENT blending).
//
// The arithmetic is the reference's (imgwarp.cpp:2233-2298 WarpAffineInvoker, :3160-3240 WarpPerspectiveInvoker, :675-904 remapBilinear<FixedPtCast>,
// the 1024-entry Q15 table of initInterTab2D :213-275 with its fix-up quirk): coordinates in 1/1024 px rounded to 1/32, weights Q15, (sum + 2^14) >> 15.
// The Q15 weights are EXACT integers, w = 32 (32 - ax | ax) (32 - ay | ay), for every table entry but (0, 0) -- (32767, 0, 0, 1) instead of (32768, 0, 0, 0),
// which cannot change an 8-bit result ((32767 a + d + 16384) >> 15 == a for all bytes a, d) -- so the sum is 32 Q with
//   Q = (p00 (32 - ax) + p01 ax) (32 - ay) + (p10 (32 - ax) + p11 ax) ay      and      (32 Q + 2^14) >> 15 == (Q + 512) >> 10
// evaluated as two packed 16-bit multiply-adds and one v_dot2_u32_u16: no table, no table reads (they were two thirds of the kernel's LDS cycles: 1024
// random 8-byte entries conflict in the banks), 8 KB of LDS less per workgroup.  tests/test_hostemu.py checks the identity on all 1024 entries.
//
// Everything here is __host__ __device__ and free of wave intrinsics so that tests/hostemu can run the very same staging and per-pixel code on the
// CPU, thread by thread, against the pinned restatement (the GPU adds only the launch geometry).
#pragma once
#include <cstdlib>
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define W8_HD __host__ __device__ __forceinline__
#else
#define W8_HD inline
#endif

namespace warp8 {

enum { LX = 32, PX = 4, TW = LX * PX /* 128 destination pixels per tile row */, ROWS_PER_STEP = 8 /* 4 waves x 2 rows */, MAX_TH = 32 };

// what the host passes (one copy per launch)
struct Args {
    double M[9];                 // inverse map, as the hook receives it
    int sw, sh, dw, dh;
    uint32_t sstep, dstep;       // bytes, both multiples of 4; both images 4-byte aligned and below 4 GB
    int th;                      // tile height: 32 (1 channel) or 16 (3 / 4 channels) = tileRows<CN>()
    int ldsPitch;                // bytes per LDS row (multiple of 4, covers the widest tile box + 8 bytes of slack)
    int ldsRows;                 // rows the LDS tile can hold
    uint32_t pitchMagic;         // floor(2^32 / (ldsPitch / 4)) + 1: dword index -> row by one mul_hi
    int bw0;                     // WarpPerspectiveInvoker's block width (the x the projective terms restart from)
    int gx, gy;
    unsigned long long sframe, dframe;
    const int* colT;             // affine: colX[dw], colY[dw] -- the column terms of every destination column, evaluated once per call (k_warp8_terms) instead of
    const int* rowT;             //         by every tile in double arithmetic; rowX[dh], rowY[dh] likewise.  nullptr: evaluate in place
    int constBorder;             // BORDER_CONSTANT: a pixel whose whole 2x2 footprint is outside the source is the border value, no sampling
    uint32_t cval;               // the border value's channels as bytes (saturate_cast<uchar> of the cv::Scalar)
    int skipLean;                // the general tile kernel leaves the tiles leanTile() accepts alone (k_warp8_lean1 served them)
    int leanLW, leanNR;          // k_warp8_lean's staging geometry: threads per box row (16 .. 256; 0 = no lean path) and rounds per tile
    uint32_t leanBuf;            // bytes per LDS tile buffer of the lean kernel (two of them)
    double rzScaleX, rzInvX, rzScaleY, rzInvY; int rzArea;      // bilinear resize on the same machinery (k_resize8_lean): cv::resize's scale factors; INTER_AREA's coefficient form
};

W8_HD int satIntD(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return v >= 2147483647.0 ? 2147483647 : v <= -2147483648.0 ? (int)-2147483648LL : (int)__double2int_rn(v);
#else
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int)-2147483648LL;
    return (int)__builtin_rint(v);                       // round half to even, like cvRound / v_cvt_i32_f64
#endif
}
W8_HD double dmul(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __dmul_rn(a, b);
#else
    return a * b;                                        // the host emulation is compiled with -ffp-contract=off
#endif
}
W8_HD double dadd(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __dadd_rn(a, b);
#else
    return a + b;
#endif
}
W8_HD double ddiv(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ddiv_rn(a, b);
#else
    return a / b;
#endif
}
W8_HD uint32_t dot4(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    return c + (a & 255) * (b & 255) + ((a >> 8) & 255) * ((b >> 8) & 255) + ((a >> 16) & 255) * ((b >> 16) & 255) + (a >> 24) * (b >> 24);
#endif
}
// packed 16-bit lanes: a * b (low 16 bits per lane), a * b + c, and the dot product of two pairs + c
W8_HD uint32_t pkMul(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b)));
#else
    return (((a & 0xffffu) * (b & 0xffffu)) & 0xffffu) | ((((a >> 16) * (b >> 16)) & 0xffffu) << 16);
#endif
}
W8_HD uint32_t pkMad(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c)));
#else
    return (((a & 0xffffu) * (b & 0xffffu) + (c & 0xffffu)) & 0xffffu) | ((((a >> 16) * (b >> 16) + (c >> 16)) & 0xffffu) << 16);
#endif
}
W8_HD uint32_t dot2(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
#else
    return c + (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16);
#endif
}
W8_HD uint32_t ld16(const unsigned char* p)
{
    typedef unsigned short u16u __attribute__((aligned(1)));
    return *reinterpret_cast<const u16u*>(p);
}
W8_HD uint32_t ld32(const unsigned char* p)
{
    typedef uint32_t u32u __attribute__((aligned(1)));
    return *reinterpret_cast<const u32u*>(p);
}
// ({hi, lo} >> 8 * (sh & 3)) & 0xffffffff
W8_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8 * (sh & 3u)));
#endif
}

// (the four functions below are the definitions; tcolX .. trowY read the per-call tables when the launch provided them)
// affine coordinate terms (WarpAffineInvoker imgwarp.cpp:2252-2262 with hal::warpAffineBlocklineNN's adelta / bdelta :2699-2713): 1/1024 px, round delta 16
W8_HD int affRowX(const Args& a, int y) { return satIntD(dmul(dadd(dmul(a.M[1], (double)y), a.M[2]), 1024.0)) + 16; }
W8_HD int affRowY(const Args& a, int y) { return satIntD(dmul(dadd(dmul(a.M[4], (double)y), a.M[5]), 1024.0)) + 16; }
W8_HD int affColX(const Args& a, int x) { return satIntD(dmul(dmul(a.M[0], (double)x), 1024.0)); }
W8_HD int affColY(const Args& a, int x) { return satIntD(dmul(dmul(a.M[3], (double)x), 1024.0)); }
W8_HD int tcolX(const Args& a, int x) { return a.colT ? a.colT[x < a.dw ? x : a.dw - 1] : affColX(a, x); }
W8_HD int tcolY(const Args& a, int x) { return a.colT ? a.colT[a.dw + (x < a.dw ? x : a.dw - 1)] : affColY(a, x); }
W8_HD int trowX(const Args& a, int y) { return a.rowT ? a.rowT[y < a.dh ? y : a.dh - 1] : affRowX(a, y); }
W8_HD int trowY(const Args& a, int y) { return a.rowT ? a.rowT[a.dh + (y < a.dh ? y : a.dh - 1)] : affRowY(a, y); }

// projective coordinates in 1/32 px (WarpPerspectiveInvoker :3195-3235 / hal::warpPerspectiveBlockline: the row terms start at the block's first column)
W8_HD void perspXY(const Args& a, int x, int y, int& X, int& Y)
{
    const int xb = (x / a.bw0) * a.bw0, x1 = x - xb;
    const double X0 = dadd(dadd(dmul(a.M[0], (double)xb), dmul(a.M[1], (double)y)), a.M[2]);
    const double Y0 = dadd(dadd(dmul(a.M[3], (double)xb), dmul(a.M[4], (double)y)), a.M[5]);
    const double W0 = dadd(dadd(dmul(a.M[6], (double)xb), dmul(a.M[7], (double)y)), a.M[8]);
    double W = dadd(W0, dmul(a.M[6], (double)x1));
    W = W != 0 ? ddiv(32.0, W) : 0;
    double fX = dmul(dadd(X0, dmul(a.M[0], (double)x1)), W);
    double fY = dmul(dadd(Y0, dmul(a.M[3], (double)x1)), W);
    fX = fX < -2147483648.0 ? -2147483648.0 : fX > 2147483647.0 ? 2147483647.0 : fX;
    fY = fY < -2147483648.0 ? -2147483648.0 : fY > 2147483647.0 ? 2147483647.0 : fY;
    X = satIntD(fX); Y = satIntD(fY);
}

// LDS layout of a workgroup (bytes): column terms (colX[128], colY[128]) | row terms (rowX[32], rowY[32]) | box terms | source tile
enum { OFF_COL = 0, OFF_ROW = OFF_COL + 2 * TW * 4, OFF_TERMS = OFF_ROW + 2 * MAX_TH * 4, OFF_TILE = OFF_TERMS + 64 };

// the box of source pixels a tile needs, clipped to the image: origin (cx0, cy0), cw x ch pixels (0 = nothing loadable), `all` = every destination pixel of
// the tile has its 2x2 footprint strictly inside the box (no per-pixel test needed), `shift` = byte offset of pixel cx0 inside its aligned dword
struct Box { int cx0, cy0, cw, ch, shift, all; };

// The box is built from a handful of coordinate evaluations (`terms`), one per lane of the first wave, so that no thread walks the double arithmetic alone:
//   affine       terms[0..7]  = rowX(y0), rowX(y1), colX(x0), colX(x1), rowY(y0), rowY(y1), colY(x0), colY(x1)
//   perspective  terms[3i..3i+2] = X, Y (1/32 px) and the sign of the denominator at corner i of the tile (i = 0..3)
template <int KIND> W8_HD int boxTermCount() { return KIND == 0 ? 8 : 12; }

template <int KIND>
W8_HD void boxTerm(const Args& a, int x0, int y0, int k, int* terms)
{
    const int x1 = (x0 + TW < a.dw ? x0 + TW : a.dw) - 1, y1 = (y0 + a.th < a.dh ? y0 + a.th : a.dh) - 1;
    if (KIND == 0) {
        const int v = (k & 2) ? ((k & 1) ? x1 : x0) : ((k & 1) ? y1 : y0);
        terms[k] = k < 4 ? ((k & 2) ? tcolX(a, v) : trowX(a, v)) : ((k & 2) ? tcolY(a, v) : trowY(a, v));
    } else if (k < 4) {
        const int x = (k & 1) ? x1 : x0, y = (k & 2) ? y1 : y0;
        perspXY(a, x, y, terms[3 * k], terms[3 * k + 1]);
        const double w = dadd(dadd(dmul(a.M[6], (double)x), dmul(a.M[7], (double)y)), a.M[8]);
        terms[3 * k + 2] = w > 0 ? 1 : w < 0 ? -1 : 0;
    }
}

template <int CN, int KIND>
W8_HD Box boxFromTerms(const Args& a, const int* t)
{
    Box b = {0, 0, 0, 0, 0, 0};                                       // (perspective early-outs below: cw = ch = 0, the generic sampler takes the tile)
    int bx0, bx1, by0, by1, exact = 1;
    if (KIND == 0) {
        // X5(x, y) = (rowX(y) + colX(x)) >> 5 with rowX, colX monotone (a rounded linear function each): the extremes are sums of the terms' extremes
        bx0 = ((t[0] < t[1] ? t[0] : t[1]) + (t[2] < t[3] ? t[2] : t[3])) >> 10; bx1 = (((t[0] > t[1] ? t[0] : t[1]) + (t[2] > t[3] ? t[2] : t[3])) >> 10) + 1;
        by0 = ((t[4] < t[5] ? t[4] : t[5]) + (t[6] < t[7] ? t[6] : t[7])) >> 10; by1 = (((t[4] > t[5] ? t[4] : t[5]) + (t[6] > t[7] ? t[6] : t[7])) >> 10) + 1;
    } else {
        // a projective map is monotone in x and in y on a rectangle its denominator does not vanish on, so the corners bound it; the reference restarts
        // its row terms every bw0 columns, which moves a coordinate by rounding only: one pixel of margin, and the per-pixel test stays on
        exact = 0;
        const int sgn = t[2] + t[5] + t[8] + t[11];
        if (sgn != 4 && sgn != -4) return b;
        int mnx = t[0], mxx = t[0], mny = t[1], mxy = t[1];
        for (int i = 1; i < 4; i++) {
            mnx = t[3 * i] < mnx ? t[3 * i] : mnx; mxx = t[3 * i] > mxx ? t[3 * i] : mxx;
            mny = t[3 * i + 1] < mny ? t[3 * i + 1] : mny; mxy = t[3 * i + 1] > mxy ? t[3 * i + 1] : mxy;
        }
        if (mnx < -(1 << 28) || mxx > (1 << 28) || mny < -(1 << 28) || mxy > (1 << 28)) return b;
        bx0 = (mnx >> 5) - 1; bx1 = (mxx >> 5) + 2; by0 = (mny >> 5) - 1; by1 = (mxy >> 5) + 2;
    }
    b.cx0 = bx0 < 0 ? 0 : bx0; b.cy0 = by0 < 0 ? 0 : by0;
    const int ex = bx1 > a.sw - 1 ? a.sw - 1 : bx1, ey = by1 > a.sh - 1 ? a.sh - 1 : by1;
    b.cw = ex - b.cx0 + 1; b.ch = ey - b.cy0 + 1;
    b.shift = (b.cx0 * CN) & 3;
    // less than a 2x2 footprint of the source under the box: nothing to stage, but every pixel can still be classified (ch = -1 marks this case: with
    // BORDER_CONSTANT the tile is mostly border value); a box beyond the LDS allotment: the generic sampler takes the whole tile (cw = ch = 0)
    if (b.cw < 2 || b.ch < 2) { b.cw = 0; b.ch = -1; b.shift = 0; b.all = 0; return b; }
    if (b.ch > a.ldsRows || ((b.shift + b.cw * CN + 3) & ~3) + 8 > a.ldsPitch) { b.cw = 0; b.ch = 0; b.all = 0; return b; }
    b.all = exact && bx0 >= 0 && by0 >= 0 && bx1 <= a.sw - 1 && by1 <= a.sh - 1;
    return b;
}

// staging: thread `tid` of 256 copies its share of the box into the LDS tile, whole aligned dwords.  Row r of the tile starts at the aligned dword that
// holds pixel (cx0, cy0 + r); sstep % 4 == 0 makes the in-dword shift the same for every row.  The last dword of the image's last row is assembled from
// bytes (a whole-dword load could read up to 3 bytes past the last pixel).
W8_HD void stage(const Args& a, const Box& b, int cn, const unsigned char* src, unsigned char* tile, int tid)
{
    if (b.cw == 0) return;
    const uint32_t nd = (uint32_t)(b.shift + b.cw * cn + 3) >> 2, pd = (uint32_t)a.ldsPitch >> 2;       // dwords per row to load / per LDS row
    const uint32_t total = pd * (uint32_t)b.ch;
    const uint32_t base = (uint32_t)b.cy0 * a.sstep + (((uint32_t)b.cx0 * (uint32_t)cn) & ~3u);          // both images are below 4 GB (host check)
    const uint32_t last = (uint32_t)(a.sh - 1) * a.sstep + (uint32_t)a.sw * (uint32_t)cn;                // one past the image's last pixel byte
    // loads are issued in groups of NB before the first of them is stored to LDS: a thread that waits for every load on its own pays one memory
    // round trip per dword (8 - 30 of them per tile), and nothing else in the workgroup can run until the tile is there.  (Handing whole rows to waves
    // -- one multiply-add per address -- was measured slower: rows of 38 dwords leave 26 of 64 lanes idle and double the slots per thread.)
    enum { NB = 8 };
    for (uint32_t i0 = (uint32_t)tid; i0 < total; i0 += 256 * NB) {
        uint32_t v[NB];
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int u = 0; u < NB; u++) {
            const uint32_t i = i0 + 256u * (uint32_t)u;
            const uint32_t r = (uint32_t)(((unsigned long long)i * a.pitchMagic) >> 32), c = i - r * pd;
            // branch-free (a divergent byte-wise fallback made the compiler drain the load queue after every load): a dword that would cross the
            // image's last byte is fetched from the last four bytes instead and shifted down; out-of-range slots re-read dword 0 of the box
            const bool on = i < total && c < nd;
            const uint32_t go = on ? base + r * a.sstep + 4 * c : base;
            const uint32_t over = go + 4 > last ? go + 4 - last : 0u;            // 0..3
            typedef uint32_t u32u __attribute__((aligned(1)));
            v[u] = *reinterpret_cast<const u32u*>(src + (go - over)) >> (8 * over);
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int u = 0; u < NB; u++) {
            const uint32_t i = i0 + 256u * (uint32_t)u;
            if (i < total) reinterpret_cast<uint32_t*>(tile)[i] = v[u];
        }
    }
}

// One destination pixel from the LDS tile in two steps, so that the LDS reads of a lane's four pixels can all be issued before the first result is needed:
//   fetchTaps   `p` points at the pixel's upper-left tap; returns the 2 * CN tap bytes of the upper row in (t[0], t[1]) and of the lower row in (t[2], t[3])
//   bilinearOf  the four Q15 weights (byte-split: wh, wl) applied to them; CN result bytes in the low bits
// LDS reads wider than a dword must sit on their natural alignment (a misaligned ds_read_b64 is replayed at 64 cycles per wave instruction) and byte-misaligned
// 16-bit reads measured slow as well (4K 8UC1, 7 degrees: 30.9 us per frame against 17.1 with aligned dwords + a byte funnel shift), so FETCH 1 -- the
// default -- reads the ALIGNED dwords around the taps and shifts; FETCH 0 keeps the reads at the taps' own addresses (A/B runs).
template <int CN, int FETCH>
W8_HD void fetchTaps(const unsigned char* p, uint32_t pitch, uint32_t (&t)[4])
{
    const unsigned char* p1 = p + pitch;
    if (CN == 4) {                                                   // a pixel is a dword: always aligned
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p); const uint32_t* q1 = reinterpret_cast<const uint32_t*>(p1);
        t[0] = q[0]; t[1] = q[1]; t[2] = q1[0]; t[3] = q1[1];
    } else if (FETCH == 0) {
        if (CN == 1) { t[0] = ld16(p); t[1] = 0; t[2] = ld16(p1); t[3] = 0; }
        else { t[0] = ld32(p); t[1] = ld16(p + 4); t[2] = ld32(p1); t[3] = ld16(p1 + 4); }
    } else {
        const uint32_t sh = (uint32_t)(uintptr_t)p & 3u;                 // pitch is a multiple of 4: the same shift on the next row
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p - sh); const uint32_t* q1 = reinterpret_cast<const uint32_t*>(p1 - sh);
        if (CN == 1) { t[0] = alignbyte(q[1], q[0], sh); t[1] = 0; t[2] = alignbyte(q1[1], q1[0], sh); t[3] = 0; }
        else {
            const uint32_t a0 = q[0], a1 = q[1], a2 = q[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
            t[0] = alignbyte(a1, a0, sh); t[1] = alignbyte(a2, a1, sh); t[2] = alignbyte(b1, b0, sh); t[3] = alignbyte(b2, b1, sh);
        }
    }
}

// wx = (32 - ax) in both 16-bit lanes, wx1 = ax in both, wy = (32 - ay) | ay << 16
template <int CN>
W8_HD uint32_t bilinearOf(const uint32_t (&t)[4], uint32_t wx0, uint32_t wx1, uint32_t wy)
{
    const unsigned long long q0 = t[0] | ((unsigned long long)t[1] << 32), q1 = t[2] | ((unsigned long long)t[3] << 32);
    uint32_t out = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c = 0; c < CN; c++) {
        // the left taps of both rows in one register (upper | lower << 16), the right taps in another: H = (h_upper | h_lower << 16), h <= 255 * 32
        const uint32_t L = (uint32_t)((q0 >> (8 * c)) & 255) | ((uint32_t)((q1 >> (8 * c)) & 255) << 16);
        const uint32_t R = (uint32_t)((q0 >> (8 * (CN + c))) & 255) | ((uint32_t)((q1 >> (8 * (CN + c))) & 255) << 16);
        const uint32_t H = pkMad(R, wx1, pkMul(L, wx0));
        out |= (dot2(H, wy, 512u) >> 10) << (8 * c);                               // <= 255: the weights sum to 1024 exactly
    }
    return out;
}

// ---- the three phases of a workgroup (256 threads), separated by barriers in the kernel; tests/hostemu runs them thread by thread -------------------------
// A: box terms (the first lanes)
template <int KIND>
W8_HD void phaseA(const Args& a, int x0, int y0, unsigned char* lds, int tid)
{
    if (tid < (KIND == 0 ? 8 : 4)) boxTerm<KIND>(a, x0, y0, tid, reinterpret_cast<int*>(lds + OFF_TERMS));
}

// B: the source box into the tile; affine: column terms of the tile's 128 columns and row terms of its rows (box origin folded into the row terms)
template <int CN, int KIND>
W8_HD void phaseB(const Args& a, const Box& b, int x0, int y0, const unsigned char* src, unsigned char* lds, int tid)
{
    stage(a, b, CN, src, lds + OFF_TILE, tid);
    if (KIND == 0) {
        int* col = reinterpret_cast<int*>(lds + OFF_COL); int* row = reinterpret_cast<int*>(lds + OFF_ROW);
        if (tid < TW) { col[tid] = tcolX(a, x0 + tid); col[TW + tid] = tcolY(a, x0 + tid); }
        else if (tid < TW + a.th) { const int r = tid - TW; row[r] = trowX(a, y0 + r) - (b.cx0 << 10); row[MAX_TH + r] = trowY(a, y0 + r) - (b.cy0 << 10); }
    }
}

// C: a lane = four horizontally adjacent destination pixels of one row per step, straight-line code: the four pixels are always evaluated (a pixel whose
// footprint is not strictly inside the loaded box reads LDS offset 0 instead) and stored as one dword / three / four when all four are good; otherwise the
// group's bit is set in the returned mask and the kernel redoes the whole group with the generic sampler afterwards (same arithmetic for interior pixels,
// borders and BORDER_TRANSPARENT handled there).  ALL = the tile's box is known to contain every footprint: no per-pixel test at all.
template <int CN> W8_HD constexpr int tileRows() { return CN == 1 ? 32 : 16; }

W8_HD uint32_t mad24(uint32_t x, uint32_t y, uint32_t z)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(x, y) + z;
#else
    return x * y + z;
#endif
}

template <int CN, int KIND, bool ALL, int FETCH>
W8_HD unsigned rowsC(const Args& a, const Box& b, int x0, int y0, const unsigned char* lds, unsigned char* dst, int tid)
{
    const int* col = reinterpret_cast<const int*>(lds + OFF_COL); const int* row = reinterpret_cast<const int*>(lds + OFF_ROW);
    const unsigned char* tile = lds + OFF_TILE;
    const int wave = tid >> 6, lane = tid & 63, lx = lane & (LX - 1), ly = lane >> 5;
    const int x = x0 + lx * PX;
    if (x >= a.dw) return 0;
    const bool fullLane = x + PX <= a.dw;
    int cX[PX], cY[PX];
    if (KIND == 0) for (int p = 0; p < PX; p++) { cX[p] = col[lx * PX + p]; cY[p] = col[TW + lx * PX + p]; }
    const uint32_t pitch = (uint32_t)a.ldsPitch, cwm = b.cw > 0 ? (uint32_t)(b.cw - 1) : 0u, chm = b.ch > 0 ? (uint32_t)(b.ch - 1) : 0u;
    unsigned redo = 0;
    constexpr int NSTEPS = tileRows<CN>() / ROWS_PER_STEP;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int st = 0; st < NSTEPS; st++) {
        const int yi = st * ROWS_PER_STEP + wave * 2 + ly, y = y0 + yi;
        const int rX = KIND == 0 ? row[yi] : 0, rY = KIND == 0 ? row[MAX_TH + yi] : 0;
        uint32_t px[PX]; bool ok = fullLane;
        uint32_t off[PX], wxa[PX], wya[PX], tap[PX][4]; bool outp[PX];
        // (1) addresses of all four pixels, (2) every LDS read, (3) the arithmetic: a pixel's reads are in flight while its neighbours' are issued
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) {
            int rx, ry, ax, ay;
            if (KIND == 0) {
                const int tX = rX + cX[p], tY = rY + cY[p];                 // 1/1024 px relative to the box origin
                rx = tX >> 10; ry = tY >> 10; ax = (tX >> 5) & 31; ay = (tY >> 5) & 31;
            } else {
                int X, Y;
                perspXY(a, x + p < a.dw ? x + p : a.dw - 1, y < a.dh ? y : a.dh - 1, X, Y);
                rx = (X >> 5) - b.cx0; ry = (Y >> 5) - b.cy0; ax = X & 31; ay = Y & 31;
            }
            off[p] = mad24((uint32_t)ry, pitch, (uint32_t)(rx * CN + b.shift));
            outp[p] = false;
            if (!ALL) {
                // inside the staged box: sampled; the whole 2x2 footprint outside the source under BORDER_CONSTANT: the border value; the rest (partial
                // footprints on the source's rim, the other border rules) leaves the group to the generic sampler
                const bool in = (uint32_t)rx < cwm && (uint32_t)ry < chm;
                const int sx = rx + b.cx0, sy = ry + b.cy0;
                outp[p] = a.constBorder && (sx >= a.sw || sx + 1 < 0 || sy >= a.sh || sy + 1 < 0);
                off[p] = in ? off[p] : 0u; ok = ok && (in || outp[p]);
            }
            wxa[p] = (uint32_t)ax | ((uint32_t)ax << 16); wya[p] = (uint32_t)(32 - ay) | ((uint32_t)ay << 16);
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) fetchTaps<CN, FETCH>(tile + off[p], pitch, tap[p]);
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) {
            px[p] = bilinearOf<CN>(tap[p], 0x00200020u - wxa[p], wxa[p], wya[p]);
            if (!ALL) px[p] = outp[p] ? a.cval : px[p];
        }
        if (y < a.dh) {
            if (ok) {
                unsigned char* D = dst + (size_t)y * a.dstep + (size_t)x * CN;
                uint32_t* d = reinterpret_cast<uint32_t*>(D);
                if (CN == 1) d[0] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
                else if (CN == 3) { d[0] = px[0] | (px[1] << 24); d[1] = (px[1] >> 8) | (px[2] << 16); d[2] = (px[2] >> 16) | (px[3] << 8); }
                else { d[0] = px[0]; d[1] = px[1]; d[2] = px[2]; d[3] = px[3]; }
            } else redo |= 1u << st;
        }
    }
    return redo;
}

template <int CN, int KIND, int FETCH = 0>
W8_HD unsigned phaseC(const Args& a, const Box& b, int x0, int y0, const unsigned char* lds, unsigned char* dst, int tid)
{
    if (b.cw == 0 && !(a.constBorder && b.ch == -1)) {                // nothing staged (box too large for the LDS allotment): every group is redone
        const int lane = tid & 63, x = x0 + (lane & (LX - 1)) * PX;
        return x < a.dw ? (1u << (tileRows<CN>() / ROWS_PER_STEP)) - 1 : 0;
    }
    return b.all ? rowsC<CN, KIND, true, FETCH>(a, b, x0, y0, lds, dst, tid) : rowsC<CN, KIND, false, FETCH>(a, b, x0, y0, lds, dst, tid);
}

// the groups phaseC left: slow(x, y, X, Y) for every destination pixel of them, (X, Y) = its source coordinates in 1/32 px (affine: from the row / column
// terms still in LDS, box origin added back)
template <int CN, int KIND, class Slow>
W8_HD void redoGroups(const Args& a, const Box& b, unsigned redo, int x0, int y0, const unsigned char* lds, int tid, Slow slow)
{
    const int* col = reinterpret_cast<const int*>(lds + OFF_COL); const int* row = reinterpret_cast<const int*>(lds + OFF_ROW);
    const int wave = tid >> 6, lane = tid & 63, lx = lane & (LX - 1), x = x0 + lx * PX;
    for (int st = 0; redo >> st; st++) {
        if (!((redo >> st) & 1)) continue;
        const int yi = st * ROWS_PER_STEP + wave * 2 + (lane >> 5), y = y0 + yi;
        if (y >= a.dh) continue;
        for (int p = 0; p < PX && x + p < a.dw; p++) {
            int X, Y;
            if (KIND == 0) { X = ((row[yi] + col[lx * PX + p]) >> 5) + (b.cx0 << 5); Y = ((row[MAX_TH + yi] + col[TW + lx * PX + p]) >> 5) + (b.cy0 << 5); }
            else perspXY(a, x + p, y, X, Y);
            slow(x + p, y, X, Y);
        }
    }
}

// ---- bicubic sampling on the tile path (affine maps, 1 / 3 channels; round 4, not yet the default: MI355CV_WARP_TAPS_TILE=1) --------------------------------
// remapBicubic (imgwarp.cpp:905-1010) reads 4 x 4 taps from one pixel left of / above the bilinear pair's upper-left tap: the tile's box grows by that margin and
// every destination pixel takes its taps from LDS -- per tap row the aligned dwords around its 4 * CN bytes and byte funnel shifts --, pairs them into 16-bit
// lanes (perm) and multiplies them with the Q15 weight pairs of initInterTab2D's table (dot2; the table sits in LDS beside the tile, `wq`: 8 dwords per
// (ay * 32 + ax) entry, ESTR dwords apart).  Exact integers: (sum + 2^14) >> 15, saturated.  Pixels whose 4 x 4 footprint is not inside the staged box are left
// to the generic sampler (redo mask), as on the bilinear path; BORDER_CONSTANT pixels wholly outside the source are the border value.
W8_HD uint32_t permBytes(uint32_t hi, uint32_t lo, uint32_t sel)               // v_perm_b32: selector bytes 0-3 pick from lo, 4-7 from hi, 0x0c = zero
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const unsigned long long v = ((unsigned long long)hi << 32) | lo; uint32_t o = 0;
    for (int i = 0; i < 4; i++) { const uint32_t q = (sel >> (8 * i)) & 0xffu; o |= (q == 0x0cu ? 0u : (uint32_t)((v >> (8 * q)) & 0xffu)) << (8 * i); }
    return o;
#endif
}
W8_HD int sdot2(uint32_t a, uint32_t b, int c)                                 // v_dot2_i32_i16
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
#else
    return (int)(short)(a & 0xffffu) * (int)(short)(b & 0xffffu) + (int)(short)(a >> 16) * (int)(short)(b >> 16) + c;
#endif
}

// the affine box of boxFromTerms<CN, 0> grown by the bicubic margins; same conventions (cw == 0: nothing staged; ch == -1: nothing to stage but classifiable)
template <int CN>
W8_HD Box boxFromTerms4(const Args& a, const int* t)
{
    Box b = {0, 0, 0, 0, 0, 0};
    const int bx0 = (((t[0] < t[1] ? t[0] : t[1]) + (t[2] < t[3] ? t[2] : t[3])) >> 10) - 1, bx1 = (((t[0] > t[1] ? t[0] : t[1]) + (t[2] > t[3] ? t[2] : t[3])) >> 10) + 2;
    const int by0 = (((t[4] < t[5] ? t[4] : t[5]) + (t[6] < t[7] ? t[6] : t[7])) >> 10) - 1, by1 = (((t[4] > t[5] ? t[4] : t[5]) + (t[6] > t[7] ? t[6] : t[7])) >> 10) + 2;
    b.cx0 = bx0 < 0 ? 0 : bx0; b.cy0 = by0 < 0 ? 0 : by0;
    const int ex = bx1 > a.sw - 1 ? a.sw - 1 : bx1, ey = by1 > a.sh - 1 ? a.sh - 1 : by1;
    b.cw = ex - b.cx0 + 1; b.ch = ey - b.cy0 + 1;
    b.shift = (b.cx0 * CN) & 3;
    if (b.cw < 4 || b.ch < 4) { b.cw = 0; b.ch = -1; b.shift = 0; b.all = 0; return b; }
    if (b.ch > a.ldsRows || ((b.shift + b.cw * CN + 3) & ~3) + 8 > a.ldsPitch) { b.cw = 0; b.ch = 0; b.all = 0; return b; }
    b.all = bx0 >= 0 && by0 >= 0 && bx1 <= a.sw - 1 && by1 <= a.sh - 1;
    return b;
}

// CN result bytes (low bits) of one pixel: `p` = its first tap in the tile, `w` = the 8 weight-pair dwords of its table entry
template <int CN>
W8_HD uint32_t bicubicAt(const unsigned char* p, uint32_t pitch, const uint32_t* w)
{
    constexpr int ND = CN == 1 ? 1 : 3;                                          // dwords of a tap row (4 * CN bytes)
    const uint32_t sh = (uint32_t)(uintptr_t)p & 3u;                             // the pitch is a multiple of 4: the same shift on every row
    uint32_t d[4][ND];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 4; r++) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p + (size_t)r * pitch - sh);
        uint32_t v[ND + 1];
        for (int i = 0; i <= ND; i++) v[i] = q[i];
        for (int i = 0; i < ND; i++) d[r][i] = alignbyte(v[i + 1], v[i], sh);
    }
    uint32_t out = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < CN; k++) {
        int sum = 1 << 14;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int r = 0; r < 4; r++)
            for (int j = 0; j < 2; j++) {
                const int b0 = 2 * j * CN + k, b1 = (2 * j + 1) * CN + k, d0 = b0 >> 2, d1 = b1 >> 2;
                const uint32_t sel = (uint32_t)(b0 & 3) | (0x0cu << 8) | ((uint32_t)((d1 == d0 ? 0 : 4) + (b1 & 3)) << 16) | (0x0cu << 24);
                sum = sdot2(permBytes(d[r][d1], d[r][d0], sel), w[r * 2 + j], sum);
            }
        int v = sum >> 15;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(v));                                          // (keeps shift + clamp + pack away from v_ashr_pk_u8_i32)
#endif
        out |= (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v) << (8 * k);
    }
    return out;
}

// phase C of the bicubic path: rowsC's structure (a lane = four neighbouring destination pixels of a row per step; one dword / three stored when all four are good)
template <int CN, bool ALL>
W8_HD unsigned rowsC4(const Args& a, const Box& b, int x0, int y0, const unsigned char* lds, const uint32_t* wq, int estr, unsigned char* dst, int tid)
{
    const int* col = reinterpret_cast<const int*>(lds + OFF_COL); const int* row = reinterpret_cast<const int*>(lds + OFF_ROW);
    const unsigned char* tile = lds + OFF_TILE;
    const int wave = tid >> 6, lane = tid & 63, lx = lane & (LX - 1), ly = lane >> 5;
    const int x = x0 + lx * PX;
    if (x >= a.dw) return 0;
    const bool fullLane = x + PX <= a.dw;
    int cX[PX], cY[PX];
    for (int p = 0; p < PX; p++) { cX[p] = col[lx * PX + p]; cY[p] = col[TW + lx * PX + p]; }
    const uint32_t pitch = (uint32_t)a.ldsPitch, cwm = b.cw > 3 ? (uint32_t)(b.cw - 3) : 0u, chm = b.ch > 3 ? (uint32_t)(b.ch - 3) : 0u;
    unsigned redo = 0;
    constexpr int NSTEPS = tileRows<CN>() / ROWS_PER_STEP;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int st = 0; st < NSTEPS; st++) {
        const int yi = st * ROWS_PER_STEP + wave * 2 + ly, y = y0 + yi;
        const int rX = row[yi], rY = row[MAX_TH + yi];
        uint32_t px[PX]; bool ok = fullLane;
        uint32_t off[PX], ent[PX]; bool outp[PX];
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) {
            const int tX = rX + cX[p], tY = rY + cY[p];                     // 1/1024 px relative to the box origin
            const int rx = (tX >> 10) - 1, ry = (tY >> 10) - 1;             // the first of the 4 x 4 taps
            ent[p] = (uint32_t)((((tY >> 5) & 31) * 32 + ((tX >> 5) & 31)) * estr);
            off[p] = mad24((uint32_t)ry, pitch, (uint32_t)(rx * CN + b.shift));
            outp[p] = false;
            if (!ALL) {
                const bool in = (uint32_t)rx < cwm && (uint32_t)ry < chm;
                const int sx = rx + b.cx0, sy = ry + b.cy0;
                outp[p] = a.constBorder && (sx >= a.sw || sx + 4 <= 0 || sy >= a.sh || sy + 4 <= 0);
                off[p] = in ? off[p] : 0u; ok = ok && (in || outp[p]);
            }
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) {
            px[p] = bicubicAt<CN>(tile + off[p], pitch, wq + ent[p]);
            if (!ALL) px[p] = outp[p] ? a.cval : px[p];
        }
        if (y < a.dh) {
            if (ok) {
                uint32_t* d = reinterpret_cast<uint32_t*>(dst + (size_t)y * a.dstep + (size_t)x * CN);
                if (CN == 1) d[0] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
                else { d[0] = px[0] | (px[1] << 24); d[1] = (px[1] >> 8) | (px[2] << 16); d[2] = (px[2] >> 16) | (px[3] << 8); }
            } else redo |= 1u << st;
        }
    }
    return redo;
}

template <int CN>
W8_HD unsigned phaseC4(const Args& a, const Box& b, int x0, int y0, const unsigned char* lds, const uint32_t* wq, int estr, unsigned char* dst, int tid)
{
    if (b.cw == 0 && !(a.constBorder && b.ch == -1)) {                // nothing staged: every group is redone
        const int lane = tid & 63, x = x0 + (lane & (LX - 1)) * PX;
        return x < a.dw ? (1u << (tileRows<CN>() / ROWS_PER_STEP)) - 1 : 0;
    }
    return b.all ? rowsC4<CN, true>(a, b, x0, y0, lds, wq, estr, dst, tid) : rowsC4<CN, false>(a, b, x0, y0, lds, wq, estr, dst, tid);
}

// ---- the lean path: one channel, affine map, tiles whose box lies wholly inside the source (Box::all) --------------------------------------------------
// Most tiles of a call are of this kind, and for them nothing of the general machinery is needed: no per-pixel inside test, no border rule, no term tables in
// LDS.  The tile kernel above spends ~63 VALU instructions per pixel (profiles/r03_warp8.txt: VALU-issue bound at 92 % busy); this path is written against
// an instruction budget instead -- per pixel 2 adds, 2 shifts, 2 bit-field extracts, 1 multiply-add for the LDS offset, 3 address ops, two ds_read2_b32 +
// two v_alignbyte for the 2 x 2 taps, v_dot4 per tap row with the weights (32 - ax, ax) packed as bytes, and d * ay + 32 h0 + 512 for the column: ~21.
// LDS holds the source box only (offset 0, so that the reads' immediate offsets stay in range); rows are staged wave by wave with a scalar row base.
#if defined(__HIP_DEVICE_COMPILE__)
#define W8_UNI(x) __builtin_amdgcn_readfirstlane(x)
#else
#define W8_UNI(x) (x)
#endif

W8_HD int imad24(int x, int y, int z)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(x, y) + z;
#else
    return x * y + z;
#endif
}

// LDS address of the tile as an integer (folded into the row terms, so that the taps' addresses need no base add), and a dword pair read from such an address
W8_HD uint32_t ldsBaseOf(const unsigned char* tile)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)tile;
#else
    (void)tile; return 0u;
#endif
}
W8_HD void ldsPair(const unsigned char* tile, uint32_t addr, uint32_t& lo, uint32_t& hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(3))) uint32_t* q = (const __attribute__((address_space(3))) uint32_t*)(uintptr_t)addr;
    (void)tile; lo = q[0]; hi = q[1];
#else
    const uint32_t* q = reinterpret_cast<const uint32_t*>(tile + addr); lo = q[0]; hi = q[1];
#endif
}
// acc with byte B replaced by (v >> SH) & 255: one SDWA shift on the device instead of shift + mask + or
template <int B, int SH> W8_HD uint32_t shrIntoByte(uint32_t acc, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t ten = (uint32_t)SH;
    if (B == 1) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "s"(ten), "v"(v));
    if (B == 2) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "s"(ten), "v"(v));
    if (B == 3) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "s"(ten), "v"(v));
    return acc;
#else
    return (acc & ~(255u << (8 * B))) | (((v >> SH) & 255u) << (8 * B));
#endif
}
template <int B> W8_HD uint32_t shr10IntoByte(uint32_t acc, uint32_t v) { return shrIntoByte<B, 10>(acc, v); }

// A tile as the lean kernel sees it (uniform over the workgroup: scalar loads, scalar arithmetic):
//   LEAN_INSIDE   every 2 x 2 footprint inside the source: plain staging, no per-pixel test
//   LEAN_RIM      BORDER_CONSTANT and footprints that reach outside: the staged box is cut to columns -1 .. sw and rows -1 .. sh of the source with the border value
//                 in the positions outside the image (an apron one pixel wide is all bilinear taps can see), and a pixel's coordinates are clamped to
//                 (-1, fraction 0) / (sw, fraction 0): such a pixel's taps are border value either way.  Also the inside tiles whose last staged dword
//                 would reach past the image's last byte, or whose box has fewer rows than a staging round (the predicated loads are safe there)
//   LEAN_OUTSIDE  BORDER_CONSTANT and the whole tile outside: the border value, no staging
//   LEAN_NO       the general kernel's (other border rules on the rim, boxes beyond the staging geometry)
enum { LEAN_NO = 0, LEAN_INSIDE = 1, LEAN_RIM = 2, LEAN_OUTSIDE = 3 };
struct LBox { int cx0, cy0, cw, ch, shift, kind; };

template <int CN>
W8_HD LBox leanClassify(const Args& a, int x0, int y0)
{
    LBox L = {0, 0, 0, 0, 0, LEAN_NO};
    if (!a.colT || !a.rowT || !a.leanLW) return L;
    int t[8];
    for (int k = 0; k < 8; k++) { boxTerm<0>(a, x0, y0, k, t); t[k] = W8_UNI(t[k]); }
    // as boxFromTerms: X(x, y) = (rowX(y) + colX(x)) >> 10 with monotone terms, so the extremes are sums of the terms' extremes
    const int bx0 = ((t[0] < t[1] ? t[0] : t[1]) + (t[2] < t[3] ? t[2] : t[3])) >> 10, bx1 = (((t[0] > t[1] ? t[0] : t[1]) + (t[2] > t[3] ? t[2] : t[3])) >> 10) + 1;
    const int by0 = ((t[4] < t[5] ? t[4] : t[5]) + (t[6] < t[7] ? t[6] : t[7])) >> 10, by1 = (((t[4] > t[5] ? t[4] : t[5]) + (t[6] > t[7] ? t[6] : t[7])) >> 10) + 1;
    const bool inside = bx0 >= 0 && by0 >= 0 && bx1 <= a.sw - 1 && by1 <= a.sh - 1;
    if (!inside) {
        if (!a.constBorder) return L;
        if (bx1 < 0 || bx0 >= a.sw || by1 < 0 || by0 >= a.sh) { L.kind = LEAN_OUTSIDE; return L; }
    }
    const int lo = inside ? 0 : -1, hx = inside ? a.sw - 1 : a.sw, hy = inside ? a.sh - 1 : a.sh;
    L.cx0 = bx0 > lo ? bx0 : lo; L.cy0 = by0 > lo ? by0 : lo;
    L.cw = (bx1 < hx ? bx1 : hx) - L.cx0 + 1; L.ch = (by1 < hy ? by1 : hy) - L.cy0 + 1;
    L.shift = (L.cx0 * CN) & 3;
    const int rpr = 256 / a.leanLW, nd4 = (L.shift + L.cw * CN + 3) & ~3;               // rows per staging round; staged bytes per row
    if (L.ch > a.ldsRows || L.ch > a.leanNR * rpr || nd4 + 8 > a.ldsPitch || nd4 > 4 * a.leanLW) return L;
    const int endB = ((L.cx0 * CN) & ~3) + nd4;                                         // one past the last staged byte of a row
    L.kind = inside && L.ch >= rpr && !(L.cy0 + L.ch == a.sh && endB > a.sw * CN) ? LEAN_INSIDE : LEAN_RIM;
    return L;
}

// Staging in two halves, so that a workgroup can have the NEXT tile's box in flight (in registers) while it samples the current one out of LDS:
//   leanLoad   NR dword loads per thread: thread = (row sub-index, dword) of a round of 256 / LW box rows; rounds past the box's last row are moved up to end
//              on it and dwords right of the box read its last dword (duplicates land on the same LDS address with the same value).  INSIDE: no predication
//              at all, the row base of a round is scalar.  RIM: a dword / row outside the image is the border value (per channel), a dword across the
//              image's right edge is fetched from the row's last four bytes and shifted down (nothing is read past a row's last pixel)
//   leanStore  the same indices into the tile
template <int CN, int LW, int NR, bool RIM>
W8_HD void leanLoad(const Args& a, const LBox& b, const unsigned char* src, int tid, uint32_t (&v)[NR])
{
    constexpr int RPR = 256 / LW;
    const int sub = LW >= 64 ? W8_UNI(tid / LW) : tid / LW, c = tid % LW;
    const int nd = (b.shift + b.cw * CN + 3) >> 2, cc = c < nd ? c : nd - 1;
    if (!RIM) {
        const uint32_t laneOff = (uint32_t)sub * a.sstep + 4u * (uint32_t)cc;
        const uint32_t base = (uint32_t)b.cy0 * a.sstep + ((uint32_t)(b.cx0 * CN) & ~3u);
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < NR; j++) {
            int start = j * RPR;
            start = start < b.ch - RPR ? start : b.ch - RPR;
            v[j] = *reinterpret_cast<const uint32_t*>(src + (base + (uint32_t)start * a.sstep) + laneOff);
        }
    } else {
        const int rowB = a.sw * CN;                                                     // bytes of pixels in a source row
        const int x4 = ((b.cx0 * CN) & ~3) + 4 * cc;                                    // the dword's first byte in its row: -4 (all apron) or >= 0
        const int nvalid = x4 < 0 ? 0 : rowB - x4 >= 4 ? 4 : rowB - x4 > 0 ? rowB - x4 : 0;      // bytes of it inside the image
        uint32_t cv4 = 0;                                                               // the border value as this dword's bytes: byte i is channel (x4 + i) mod CN
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i < 4; i++) { const int ch = CN == 1 ? 0 : ((x4 + i) % CN + CN) % CN; cv4 |= ((a.cval >> (8 * ch)) & 255u) << (8 * i); }
        const uint32_t mask = nvalid >= 4 ? 0xffffffffu : (1u << (8 * nvalid)) - 1u, shr = nvalid > 0 && nvalid < 4 ? 8u * (uint32_t)(4 - nvalid) : 0u;
        const int xl = nvalid > 0 && nvalid < 4 ? rowB - 4 : x4;                        // rowB >= 4 (plan)
        typedef uint32_t u32u __attribute__((aligned(1)));
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < NR; j++) {
            int start = j * RPR;
            start = start < b.ch - RPR ? start : b.ch - RPR;
            start = start > 0 ? start : 0;
            const int y = b.cy0 + start + sub;
            const bool in = nvalid > 0 && (unsigned)y < (unsigned)a.sh;
            const uint32_t off = in ? (uint32_t)y * a.sstep + (uint32_t)xl : 0u;
            const uint32_t w = *reinterpret_cast<const u32u*>(src + off) >> shr;
            v[j] = in ? (w & mask) | (cv4 & ~mask) : cv4;
        }
    }
}
template <int CN, int LW, int NR>
W8_HD void leanStore(const Args& a, const LBox& b, unsigned char* tile, int tid, const uint32_t (&v)[NR])
{
    constexpr int RPR = 256 / LW;
    const int sub = LW >= 64 ? W8_UNI(tid / LW) : tid / LW, c = tid % LW;
    const int nd = (b.shift + b.cw * CN + 3) >> 2, cc = c < nd ? c : nd - 1;
    const uint32_t laneOff = (uint32_t)sub * (uint32_t)a.ldsPitch + 4u * (uint32_t)cc;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < NR; j++) {
        int start = j * RPR;
        start = start < b.ch - RPR ? start : b.ch - RPR;
        start = start > 0 ? start : 0;
        *reinterpret_cast<uint32_t*>(tile + (uint32_t)start * (uint32_t)a.ldsPitch + laneOff) = v[j];
    }
}

// the row terms of a lane's rows (the same for every tile of a tile row: loaded once per workgroup)
struct LeanRowT { int rX[MAX_TH / ROWS_PER_STEP], rY[MAX_TH / ROWS_PER_STEP]; };
template <int CN>
W8_HD void leanRowTerms(const Args& a, int y0, int tid, LeanRowT& rt)
{
    const int wave = W8_UNI(tid >> 6), ly = (tid & 63) >> 5;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int st = 0; st < tileRows<CN>() / ROWS_PER_STEP; st++) {
        int y = y0 + st * ROWS_PER_STEP + wave * 2 + ly;
        y = y < a.dh ? y : a.dh - 1;
        rt.rX[st] = a.rowT[y]; rt.rY[st] = a.rowT[a.dh + y];
    }
}

// a tile wholly outside the source under BORDER_CONSTANT
template <int CN>
W8_HD void leanFill(const Args& a, int x0, int y0, unsigned char* dst, int tid)
{
    const int wave = W8_UNI(tid >> 6), lane = tid & 63, lx = lane & (LX - 1), ly = lane >> 5;
    const int x = x0 + lx * PX;
    if (x >= a.dw) return;
    uint32_t cv[CN];                                                                    // the 4 * CN bytes of four border pixels
    for (int k = 0; k < CN; k++) { cv[k] = 0; for (int i = 0; i < 4; i++) cv[k] |= ((a.cval >> (8 * ((4 * k + i) % CN))) & 255u) << (8 * i); }
    for (int st = 0; st < tileRows<CN>() / ROWS_PER_STEP; st++) {
        const int y = y0 + st * ROWS_PER_STEP + wave * 2 + ly;
        if (y >= a.dh) break;
        unsigned char* d = dst + (uint32_t)y * a.dstep + (uint32_t)x * CN;
        if (x + PX <= a.dw) for (int k = 0; k < CN; k++) reinterpret_cast<uint32_t*>(d)[k] = cv[k];
        else for (int i = 0; i < (a.dw - x) * CN; i++) d[i] = (unsigned char)(cv[i >> 2] >> (8 * (i & 3)));
    }
}

W8_HD uint32_t ldsOne(const unsigned char* tile, uint32_t addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)tile; return *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)addr;
#else
    return *reinterpret_cast<const uint32_t*>(tile + addr);
#endif
}

// a lane = four horizontally adjacent destination pixels of one row per step, two rows per wave, eight rows per step of the workgroup
template <int CN, bool RIM>
W8_HD void leanRows(const Args& a, const LBox& b, int x0, int y0, const unsigned char* tile, unsigned char* dst, int tid, const LeanRowT& rt)
{
    const int wave = W8_UNI(tid >> 6), lane = tid & 63, lx = lane & (LX - 1), ly = lane >> 5;
    const int x = x0 + lx * PX;
    if (x >= a.dw) return;
    const bool full = x + PX <= a.dw;
    int cX[PX], cY[PX];
    if (full) { for (int p = 0; p < PX; p++) { cX[p] = a.colT[x + p]; cY[p] = a.colT[a.dw + x + p]; } }
    else      { for (int p = 0; p < PX; p++) { cX[p] = tcolX(a, x + p); cY[p] = tcolY(a, x + p); } }
    constexpr int NSTEPS = tileRows<CN>() / ROWS_PER_STEP;
    // box origin folded into the column terms; one channel: also the in-dword shift and the tile's LDS address (a pixel is a byte, so they are pixels too)
    const int sb = b.shift + (int)ldsBaseOf(tile);
    const int fx = (CN == 1 ? sb - b.cx0 : -b.cx0) * 1024, fy = -b.cy0 * 1024;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int p = 0; p < PX; p++) { cX[p] += fx; cY[p] += fy; }
    const int loX = fx - 1024, hiX = fx + a.sw * 1024, loY = fy - 1024, hiY = fy + a.sh * 1024;     // RIM: source column -1 / sw and row -1 / sh, fraction 0
    const uint32_t pitch = (uint32_t)a.ldsPitch;
    uint32_t doff = (uint32_t)(y0 + wave * 2 + ly) * a.dstep + (uint32_t)x * CN;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int st = 0; st < NSTEPS; st++) {
        const int y = y0 + st * ROWS_PER_STEP + wave * 2 + ly;
        uint32_t off[PX], a0[PX], a1[PX], a2[PX], b0[PX], b1[PX], b2[PX], px[PX][CN]; int tXs[PX], tYs[PX];
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) {
            int tX = rt.rX[st] + cX[p], tY = rt.rY[st] + cY[p];                        // 1/1024 px relative to the box's origin
            if (RIM) { tX = tX < loX ? loX : tX > hiX ? hiX : tX; tY = tY < loY ? loY : tY > hiY ? hiY : tY; }
            if (CN == 1) off[p] = mad24((uint32_t)(tY >> 10), pitch, (uint32_t)(tX >> 10));
            else off[p] = mad24((uint32_t)(tY >> 10), pitch, mad24((uint32_t)(tX >> 10), (uint32_t)CN, (uint32_t)sb));
            tXs[p] = tX; tYs[p] = tY;
            ldsPair(tile, off[p] & ~3u, a0[p], a1[p]); ldsPair(tile, (off[p] & ~3u) + pitch, b0[p], b1[p]);
            if (CN == 3) { a2[p] = ldsOne(tile, (off[p] & ~3u) + 8u); b2[p] = ldsOne(tile, (off[p] & ~3u) + pitch + 8u); }
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) {
            // v_alignbyte_b32 shifts by its operand's two low bits (measured on gfx950, tools/probes/alignbyte.hip): the byte offset goes in unmasked
            const uint32_t ax = ((uint32_t)tXs[p] >> 5) & 31u, ay = ((uint32_t)tYs[p] >> 5) & 31u;
            if (CN == 1) {
                const uint32_t wx = mad24(ax, 255u, 32u);                              // (32 - ax) | ax << 8
                // both row sums carry + 16 (the dot product's free addend): 32 (h0 + 16) = 32 h0 + 512 is the rounding term, and the difference is unchanged
                const uint32_t h0 = dot4(alignbyte(a1[p], a0[p], off[p]), wx, 16u), h1 = dot4(alignbyte(b1[p], b0[p], off[p]), wx, 16u);
                px[p][0] = (h0 << 5) + (uint32_t)imad24((int)h1 - (int)h0, (int)ay, 0);                // h0 (32 - ay) + h1 ay + 512: the pixel is bits 10..17
            } else {
                // the taps' 8 bytes (left pixel b0..b2, right pixel b3..b5) as two dwords per row; channel c's pair is bytes c and c + 3: the weights sit on
                // bytes 0 and 3 and the data is shifted under them
                const uint32_t wx = mad24(ax, 0xffffffu, 32u);                         // (32 - ax) | ax << 24
                const uint32_t eA0 = alignbyte(a1[p], a0[p], off[p]), eA1 = alignbyte(a2[p], a1[p], off[p]);
                const uint32_t eB0 = alignbyte(b1[p], b0[p], off[p]), eB1 = alignbyte(b2[p], b1[p], off[p]);
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (int c = 0; c < CN; c++) {
                    const uint32_t h0 = dot4(c ? alignbyte(eA1, eA0, (uint32_t)c) : eA0, wx, 16u), h1 = dot4(c ? alignbyte(eB1, eB0, (uint32_t)c) : eB0, wx, 16u);
                    px[p][c] = (h0 << 5) + (uint32_t)imad24((int)h1 - (int)h0, (int)ay, 0);
                }
            }
        }
        if (y < a.dh) {
            if (full) {
                if (CN == 1) *reinterpret_cast<uint32_t*>(dst + doff) = shr10IntoByte<3>(shr10IntoByte<2>(shr10IntoByte<1>(px[0][0] >> 10, px[1][0]), px[2][0]), px[3][0]);
                else {
                    uint32_t* d = reinterpret_cast<uint32_t*>(dst + doff);               // 12 bytes: (p0: c0 c1 c2, p1: c0) (p1: c1 c2, p2: c0 c1) (p2: c2, p3: c0 c1 c2)
                    d[0] = shr10IntoByte<3>(shr10IntoByte<2>(shr10IntoByte<1>(px[0][0] >> 10, px[0][1 % CN]), px[0][2 % CN]), px[1][0]);
                    d[1] = shr10IntoByte<3>(shr10IntoByte<2>(shr10IntoByte<1>(px[1][1 % CN] >> 10, px[1][2 % CN]), px[2][0]), px[2][1 % CN]);
                    d[2] = shr10IntoByte<3>(shr10IntoByte<2>(shr10IntoByte<1>(px[2][2 % CN] >> 10, px[3][0]), px[3][1 % CN]), px[3][2 % CN]);
                }
            } else for (int p = 0; p < PX && x + p < a.dw; p++) for (int c = 0; c < CN; c++) dst[doff + p * CN + c] = (unsigned char)(px[p][c] >> 10);
        }
        doff += (uint32_t)ROWS_PER_STEP * a.dstep;
    }
}

inline void leanGeometry(Args& a, int cn);

// ---- bilinear resize of 8-bit images on the lean kernel's machinery (k_resize8_lean) -----------------------------------------------------------------------
// cv::resize INTER_LINEAR on CV_8U (resize.cpp: HResizeLinear + VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>): coefficients of 11 bits,
//   t = p0 a0 + p1 a1 per tap row,  pixel = (((b0 (t0 >> 4)) >> 16) + ((b1 (t1 >> 4)) >> 16) + 2) >> 2.
// An axis-aligned map needs no per-pixel coordinate arithmetic at all: a pixel's LDS offset is (row part) + (column part), both from per-call tables
// (k_resize8_terms) -- colT: sx[dw] (clamped like the reference's xofs), a0 | a1 << 16 [dw]; rowT: y0[dh], y1[dh] (clamped rows), b0 << 12 [dh], b1 << 12 [dh].
// (b (t >> 4)) >> 16 is one v_mul_hi_u32_u24 of (t & ~15) and (b << 12); the tap pair of a row is one v_perm_b32 into two 16-bit lanes + one v_dot2_u32_u16.
W8_HD int floorF(float v) { const int i = (int)v; return i - ((float)i > v); }
W8_HD int floorD(double v) { const int i = (int)v; return i - ((double)i > v); }
W8_HD int rintF(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rn(v);
#else
    return (int)__builtin_rintf(v);
#endif
}
W8_HD void rzCoef(int d, double scale, double inv, int areaMode, int& s, float& f)       // resize.cpp:3897-3925 (linear) / :3870-3895 (INTER_AREA upscale)
{
    if (!areaMode) { f = (float)(dadd(dmul((double)d + 0.5, scale), -0.5)); s = floorF(f); f -= (float)s; }
    else { s = floorD(dmul((double)d, scale)); f = (float)dadd((double)(d + 1), -dmul((double)(s + 1), inv)); f = f <= 0 ? 0.f : f - (float)floorF(f); }
}
W8_HD int sat16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
W8_HD void rzColTerm(const Args& a, int dx, int& sx, uint32_t& a01)
{
    float fx; rzCoef(dx, a.rzScaleX, a.rzInvX, a.rzArea, sx, fx);
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= a.sw - 1) { fx = 0; sx = a.sw - 1; }
    const int a0 = sat16(rintF((1.f - fx) * 2048)), a1 = sat16(rintF(fx * 2048));
    a01 = (uint32_t)a0 | ((uint32_t)a1 << 16);
}
W8_HD void rzRowTerm(const Args& a, int dy, int& y0, int& y1, uint32_t& b0s, uint32_t& b1s)
{
    int sy; float fy; rzCoef(dy, a.rzScaleY, a.rzInvY, a.rzArea, sy, fy);
    y0 = sy >= 0 ? (sy < a.sh ? sy : a.sh - 1) : 0; y1 = sy + 1 >= 0 ? (sy + 1 < a.sh ? sy + 1 : a.sh - 1) : 0;
    b0s = (uint32_t)sat16(rintF((1.f - fy) * 2048)) << 12; b1s = (uint32_t)sat16(rintF(fy * 2048)) << 12;
}
W8_HD uint32_t mulhi24(uint32_t x, uint32_t y)                                           // bits 47:32 of the product of two 24-bit values
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r;
#else
    return (uint32_t)(((unsigned long long)x * y) >> 32);
#endif
}
// bytes i0 and i1 (0..7) of the 8 bytes (hi:lo) as two 16-bit lanes
template <int I0, int I1> W8_HD uint32_t bytePair(uint32_t hi, uint32_t lo)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)I0 | ((uint32_t)I1 << 16));
#else
    const unsigned long long q = ((unsigned long long)hi << 32) | lo;
    return (uint32_t)((q >> (8 * I0)) & 255u) | ((uint32_t)((q >> (8 * I1)) & 255u) << 16);
#endif
}

// the source box of a destination tile, from the tables (the clamped indices are monotone in the destination index); every position is inside the image
template <int CN>
W8_HD LBox rzClassify(const Args& a, int x0, int y0)
{
    LBox L = {0, 0, 0, 0, 0, LEAN_NO};
    if (!a.colT || !a.rowT || !a.leanLW) return L;
    const int x1 = (x0 + TW < a.dw ? x0 + TW : a.dw) - 1, y1 = (y0 + a.th < a.dh ? y0 + a.th : a.dh) - 1;
    const int sa = W8_UNI(a.colT[x0]), sb = W8_UNI(a.colT[x1]), ya = W8_UNI(a.rowT[y0]), yb = W8_UNI(a.rowT[a.dh + y1]);
    L.cx0 = sa; L.cy0 = ya;
    L.cw = (sb + 1 < a.sw ? sb + 1 : a.sw - 1) - sa + 1; L.ch = yb - ya + 1;
    L.shift = (L.cx0 * CN) & 3;
    const int rpr = 256 / a.leanLW, nd4 = (L.shift + L.cw * CN + 3) & ~3;
    if (L.cw < 1 || L.ch < 1 || L.ch > a.ldsRows || L.ch > a.leanNR * rpr || nd4 + 8 > a.ldsPitch || nd4 > 4 * a.leanLW) return L;
    const int endB = ((L.cx0 * CN) & ~3) + nd4;
    L.kind = L.ch >= rpr && !(L.cy0 + L.ch == a.sh && endB > a.sw * CN) ? LEAN_INSIDE : LEAN_RIM;       // RIM = the predicated loader (no apron is ever read)
    return L;
}

struct RzRowT { int r0[MAX_TH / ROWS_PER_STEP], dy[MAX_TH / ROWS_PER_STEP]; uint32_t b0[MAX_TH / ROWS_PER_STEP], b1[MAX_TH / ROWS_PER_STEP]; };
template <int CN>
W8_HD void rzRowTerms(const Args& a, int y0, int tid, RzRowT& rt)
{
    const int wave = W8_UNI(tid >> 6), ly = (tid & 63) >> 5;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int st = 0; st < tileRows<CN>() / ROWS_PER_STEP; st++) {
        int y = y0 + st * ROWS_PER_STEP + wave * 2 + ly;
        y = y < a.dh ? y : a.dh - 1;
        rt.r0[st] = a.rowT[y]; rt.dy[st] = a.rowT[a.dh + y] - rt.r0[st];
        rt.b0[st] = (uint32_t)a.rowT[2 * a.dh + y]; rt.b1[st] = (uint32_t)a.rowT[3 * a.dh + y];
    }
}

template <int CN>
W8_HD void rzRows(const Args& a, const LBox& b, int x0, int y0, const unsigned char* tile, unsigned char* dst, int tid, const RzRowT& rt)
{
    const int wave = W8_UNI(tid >> 6), lane = tid & 63, lx = lane & (LX - 1), ly = lane >> 5;
    const int x = x0 + lx * PX;
    if (x >= a.dw) return;
    const bool full = x + PX <= a.dw;
    uint32_t colOff[PX], a01[PX];
    const int sb = b.shift + (int)ldsBaseOf(tile) - b.cx0 * CN;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int p = 0; p < PX; p++) {
        const int xx = x + p < a.dw ? x + p : a.dw - 1;
        colOff[p] = (uint32_t)(a.colT[xx] * CN + sb); a01[p] = (uint32_t)a.colT[a.dw + xx];
    }
    constexpr int NSTEPS = tileRows<CN>() / ROWS_PER_STEP;
    const uint32_t pitch = (uint32_t)a.ldsPitch;
    uint32_t doff = (uint32_t)(y0 + wave * 2 + ly) * a.dstep + (uint32_t)x * CN;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int st = 0; st < NSTEPS; st++) {
        const int y = y0 + st * ROWS_PER_STEP + wave * 2 + ly;
        const uint32_t rowOff = (uint32_t)(rt.r0[st] - b.cy0) * pitch, pitch2 = rt.dy[st] ? pitch : 0u, b0s = rt.b0[st], b1s = rt.b1[st];
        uint32_t off[PX], a0[PX], a1[PX], a2[PX], b0[PX], b1[PX], b2[PX], px[PX][CN];
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) {
            off[p] = rowOff + colOff[p];
            ldsPair(tile, off[p] & ~3u, a0[p], a1[p]); ldsPair(tile, (off[p] & ~3u) + pitch2, b0[p], b1[p]);
            if (CN == 3) { a2[p] = ldsOne(tile, (off[p] & ~3u) + 8u); b2[p] = ldsOne(tile, (off[p] & ~3u) + pitch2 + 8u); }
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int p = 0; p < PX; p++) {
            const uint32_t eA0 = alignbyte(a1[p], a0[p], off[p]), eB0 = alignbyte(b1[p], b0[p], off[p]);
            uint32_t eA1 = 0, eB1 = 0;
            if (CN == 3) { eA1 = alignbyte(a2[p], a1[p], off[p]); eB1 = alignbyte(b2[p], b1[p], off[p]); }
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int c = 0; c < CN; c++) {
                const uint32_t pa = c == 0 ? bytePair<0, CN>(eA1, eA0) : c == 1 ? bytePair<1, 1 + CN>(eA1, eA0) : bytePair<2, 2 + CN>(eA1, eA0);
                const uint32_t pb = c == 0 ? bytePair<0, CN>(eB1, eB0) : c == 1 ? bytePair<1, 1 + CN>(eB1, eB0) : bytePair<2, 2 + CN>(eB1, eB0);
                const uint32_t t0 = dot2(pa, a01[p], 0u), t1 = dot2(pb, a01[p], 0u);
                px[p][c] = mulhi24(t0 & ~15u, b0s) + mulhi24(t1 & ~15u, b1s) + 2u;            // the pixel is bits 2..9
            }
        }
        if (y < a.dh) {
            if (full) {
                if (CN == 1) *reinterpret_cast<uint32_t*>(dst + doff) = shrIntoByte<3, 2>(shrIntoByte<2, 2>(shrIntoByte<1, 2>((px[0][0] >> 2) & 255u, px[1][0]), px[2][0]), px[3][0]);
                else {
                    uint32_t* d = reinterpret_cast<uint32_t*>(dst + doff);
                    d[0] = shrIntoByte<3, 2>(shrIntoByte<2, 2>(shrIntoByte<1, 2>((px[0][0] >> 2) & 255u, px[0][1 % CN]), px[0][2 % CN]), px[1][0]);
                    d[1] = shrIntoByte<3, 2>(shrIntoByte<2, 2>(shrIntoByte<1, 2>((px[1][1 % CN] >> 2) & 255u, px[1][2 % CN]), px[2][0]), px[2][1 % CN]);
                    d[2] = shrIntoByte<3, 2>(shrIntoByte<2, 2>(shrIntoByte<1, 2>((px[2][2 % CN] >> 2) & 255u, px[3][0]), px[3][1 % CN]), px[3][2 % CN]);
                }
            } else for (int p = 0; p < PX && x + p < a.dw; p++) for (int c = 0; c < CN; c++) dst[doff + p * CN + c] = (unsigned char)(px[p][c] >> 2);
        }
        doff += (uint32_t)ROWS_PER_STEP * a.dstep;
    }
}

// host side: can k_resize8_lean take this bilinear resize?  (the box of a tile is TW x th destination pixels times the scale factors, + the second tap)
inline bool planResize(Args& a, int cn, int sw, int sh, int dw, int dh, size_t sstep, size_t dstep, const void* src, const void* dst,
                       double scale_x, double inv_x, double scale_y, double inv_y, int areaMode, size_t* ldsBytes)
{
    if (((uintptr_t)src | (uintptr_t)dst | sstep | dstep) & 3) return false;
    if (sw < 2 || sh < 1 || dw < 1 || dh < 1 || (cn != 1 && cn != 3) || sw * cn < 4) return false;
    if ((unsigned long long)sh * sstep >= (1ull << 32) || (unsigned long long)dh * dstep >= (1ull << 32)) return false;
    if (!(scale_x > 0 && scale_x < 64 && scale_y > 0 && scale_y < 64)) return false;
    a = Args();
    a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh; a.sstep = (uint32_t)sstep; a.dstep = (uint32_t)dstep;
    a.rzScaleX = scale_x; a.rzInvX = inv_x; a.rzScaleY = scale_y; a.rzInvY = inv_y; a.rzArea = areaMode;
    a.th = cn == 1 ? tileRows<1>() : tileRows<3>();
    a.gx = (dw + TW - 1) / TW; a.gy = (dh + a.th - 1) / a.th;
    const int ibw = (int)(TW * scale_x) + 4, ibh = (int)(a.th * scale_y) + 4;
    a.ldsPitch = ((ibw * cn + 3 + 3) & ~3) + 8;
    if (!((a.ldsPitch >> 2) & 1)) a.ldsPitch += 4;
    a.ldsRows = ibh;
    leanGeometry(a, cn);
    *ldsBytes = 2 * (size_t)a.leanBuf;
    return a.leanLW != 0;
}

// the lean kernels' staging geometry for a call whose LDS box is ldsPitch x ldsRows: LW threads per box row (the row's dwords, a power of two), 256 / LW rows
// per round, NR rounds in registers; two LDS buffers within the 64 KB a workgroup may ask for
inline void leanGeometry(Args& a, int cn)
{
    a.leanLW = a.leanNR = 0;
    const int ndMax = a.ldsPitch / 4 - 2;
    int lw = 16; while (lw < ndMax) lw *= 2;
    if (lw > 256) return;
    const int rounds = (a.ldsRows + 256 / lw - 1) / (256 / lw);
    const int nr = rounds <= 6 ? 6 : rounds <= 10 ? 10 : rounds <= 14 ? 14 : rounds <= 20 ? 20 : 0;
    if (!nr || (cn == 1 && lw > 64) || (cn == 3 && lw < 32)) return;                   // (the instantiated combinations)
    a.leanBuf = ((uint32_t)a.ldsPitch * (uint32_t)a.ldsRows + 15u) & ~15u;
    if (2 * (size_t)a.leanBuf > 64 * 1024) return;
    a.leanLW = lw; a.leanNR = nr;
}

// host side: can the tile kernel take this call, and with how much LDS?  Affine: the box of a tile has the same size everywhere (up to rounding);
// perspective: the boxes of the tiles at the image's corners, edge centres and centre are measured with the kernel's own code.  Tiles whose box exceeds
// what was allotted take the generic sampler inside the kernel, so the estimate bounds speed, never correctness.
inline bool plan(Args& a, int cn, int kind, const double* M, int sw, int sh, int dw, int dh, size_t sstep, size_t dstep, const void* src, const void* dst, int bw0,
                 size_t* ldsBytes, int margin = 0 /* extra box pixels per axis: 3 for the bicubic path (affine only) */)
{
    if (((uintptr_t)src | (uintptr_t)dst | sstep | dstep) & 3) return false;
    if (sw < 2 || sh < 2 || dw < 1 || dh < 1 || (cn != 1 && cn != 3 && cn != 4)) return false;
    if ((unsigned long long)sh * sstep >= (1ull << 32) || (unsigned long long)dh * dstep >= (1ull << 32)) return false;
    a = Args();
    for (int i = 0; i < (kind == 0 ? 6 : 9); i++) a.M[i] = M[i];
    a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh; a.sstep = (uint32_t)sstep; a.dstep = (uint32_t)dstep; a.bw0 = bw0 > 0 ? bw0 : 1;
    a.th = cn == 1 ? tileRows<1>() : tileRows<3>();
    a.gx = (dw + TW - 1) / TW; a.gy = (dh + a.th - 1) / a.th;
    auto ab = [](double v) { return v < 0 ? -v : v; };
    double bw, bh;
    if (kind == 0) {
        // the fixed-point sums must stay far from the int range (the kernel's box relies on them being monotone)
        const double lim = 1048576.0;                                                      // 2^20 pixels
        if (!(ab(M[0]) * dw + ab(M[1]) * dh + ab(M[2]) < lim && ab(M[3]) * dw + ab(M[4]) * dh + ab(M[5]) < lim)) return false;
        bw = ab(M[0]) * (TW - 1) + ab(M[1]) * (a.th - 1) + 4 + margin; bh = ab(M[3]) * (TW - 1) + ab(M[4]) * (a.th - 1) + 4 + margin;
    } else {
        a.ldsPitch = 1 << 20; a.ldsRows = 1 << 20;
        bw = bh = 0;
        const int txs[3] = {0, a.gx / 2, a.gx - 1}, tys[3] = {0, a.gy / 2, a.gy - 1};
        for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) {
            int terms[12];
            for (int k = 0; k < 4; k++) boxTerm<1>(a, txs[i] * TW, tys[j] * a.th, k, terms);
            const Box b = cn == 1 ? boxFromTerms<1, 1>(a, terms) : cn == 3 ? boxFromTerms<3, 1>(a, terms) : boxFromTerms<4, 1>(a, terms);
            bw = bw > b.cw + 2 ? bw : b.cw + 2; bh = bh > b.ch + 2 ? bh : b.ch + 2;
        }
        if (bw < 4 || bh < 4) return false;                                                // every probed tile looks outside the source: nothing to stage
    }
    if (!(bw < 4096 && bh < 4096)) return false;
    const int ibw = (int)bw + 1, ibh = (int)bh + 1;
    a.ldsPitch = ((ibw * cn + 3 + 3) & ~3) + 8;
    if (!((a.ldsPitch >> 2) & 1)) a.ldsPitch += 4;           // an odd number of dwords per row: the taps of a slanted line step through the LDS banks instead of revisiting them
#if !defined(__HIP_DEVICE_COMPILE__)
    if (const char* e = std::getenv("MI355CV_WARP8_PITCHMOD")) {      // tuning experiments: force (dwords per LDS row) mod 64
        const int r = atoi(e) & 63;
        while (((a.ldsPitch >> 2) & 63) != r) a.ldsPitch += 4;
    }
#endif
    a.ldsRows = ibh;
    a.pitchMagic = (uint32_t)((1ull << 32) / (uint32_t)(a.ldsPitch / 4)) + 1;
    *ldsBytes = (size_t)OFF_TILE + (size_t)a.ldsPitch * a.ldsRows;
    if ((cn == 1 || cn == 3) && kind == 0 && sw * cn >= 4) leanGeometry(a, cn);
    return *ldsBytes <= 40 * 1024 && (size_t)a.ldsPitch / 4 * a.ldsRows < 65536;
}

} // namespace warp8
