// sepmx_body.h -- the host-visible half of sepmx.hip (the 8-bit Q8.8 separable smoothing on the matrix cores): geometry plan, the two Toeplitz operand tables, the
// staging of one 16-byte chunk and the lane <-> element maps of v_mfma_i32_32x32x32_i8.  Everything here is __host__ __device__ and is what the kernel itself runs;
// tests/hostemu/sepmx_emu.cpp replays a whole workgroup on the CPU with these functions (the matrix instruction emulated from its operand maps) against the
// restatement of fixedSmoothInvoker (smooth.simd.hpp:1926), so the index arithmetic is checked without a GPU.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#if defined(__HIPCC__)
#define MX_HD __host__ __device__ __forceinline__
#else
#define MX_HD inline
#endif

namespace sepmx {

constexpr int TW = 256;          // elements (bytes) of a row one workgroup owns
constexpr int NWAVE = 8;         // wave w owns elements [32 w, 32 w + 32) of the strip
constexpr int TR = 32;           // rows per step (one matrix tile)
constexpr int MAXKS = 5;         // 32-byte K steps per pass: 32 + delta + (nx - 1) cn <= 32 KSX, 32 + ny - 1 <= 32 KSY

struct Geom {
    int W, H, cn, WE;                          // the ROI; WE = W * cn elements per row
    int fullW, fullH, offX, offY, border;      // the image around it (real pixels outside the ROI are read), cv border code 0 .. 4
    int nx, ny, ax, ay;
    int ksx, ksy;
    int delta;                                 // the staged row starts at ROI element X0 - ax * cn - delta (delta makes the 16-byte loads aligned)
    int nchunk, P;                             // 16-byte chunks staged per row, LDS pitch in bytes (16 * odd: conflict-free ds_read_b128 down a column)
    int seg;                                   // output rows per segment (a multiple of TR)
    int accR0, accL0;                          // accumulator seeds, see below
};

// v_mfma_i32_32x32x32_i8 operand maps (checked against the hardware by k_ccorr_ring_i8, templmatch.hip): lane (idx = lane & 31, h = lane >> 5), byte i of the 16-byte
// operand <-> k = 16 h + i for A (idx = m) and B (idx = n) alike; result register i of lane (n, h) <-> row regRow(h, i), column n.
MX_HD int regRow(int h, int i) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

MX_HD int borderIdx(int p, int len, int type)              // borderInterpolate (core/src/copy.cpp:748-793); -1 for BORDER_CONSTANT
{
    if ((unsigned)p < (unsigned)len) return p;
    if (type == 1) return p < 0 ? 0 : len - 1;
    if (type == 2 || type == 4) {
        const int d = type == 4;
        if (len == 1) return 0;
        do { if (p < 0) p = -p - 1 + d; else p = len - 1 - (p - len) - d; } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (type == 3) {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1;
}

// The arithmetic.  fixedSmoothInvoker with taps that sum to <= 256 per axis (no ufixedpoint saturation): dst = (sum_j ky[j] * R[y + j] + 2^15) >> 16 with the Q8.8 row sums
// R = sum_i kx[i] * s <= 255 * 256.  On signed 8-bit operands:
//   row pass     acc = accR0 + sum kx (s - 128),  accR0 = 128 sum(kx) + 128 - 32768           =>  acc = R + 128 - 32768 in [-32640, 32640]
//                acc = 256 Hh + L with Hh = (signed) byte 1, L = (unsigned) byte 0; l = L - 128 =>  R = 256 Hh + l + 32768, both Hh and l in int8
//   column pass  accH = sum ky Hh,  accL = accL0 + sum ky l,  accL0 = 32768 sum(ky) + 32768     =>  (accH << 8) + accL = sum ky R + 2^15, < 2^24: the result is its byte 2
MX_HD void seeds(Geom& g, const uint16_t* kx, const uint16_t* ky)
{
    int sx = 0, sy = 0;
    for (int i = 0; i < g.nx; i++) sx += kx[i];
    for (int i = 0; i < g.ny; i++) sy += ky[i];
    g.accR0 = 128 * sx + 128 - 32768;
    g.accL0 = 32768 * sy + 32768;
}

// row pass B operand (Toeplitz of kx): table[ks][lane][16], lane (n, h), byte i: staged column k = 32 ks + 16 h + i (relative to the wave's first column) holds ROI element
// x + (k - n) - ax cn - delta for output element x = first + n: tap (k - n - delta) / cn when that is a whole number in [0, nx)
inline void buildRowB(const Geom& g, const uint16_t* kx, int8_t* tab)
{
    for (int ks = 0; ks < g.ksx; ks++)
        for (int lane = 0; lane < 64; lane++)
            for (int i = 0; i < 16; i++) {
                const int n = lane & 31, h = lane >> 5, d = 32 * ks + 16 * h + i - n - g.delta;
                tab[(ks * 64 + lane) * 16 + i] = (d >= 0 && d % g.cn == 0 && d / g.cn < g.nx) ? (int8_t)kx[d / g.cn] : (int8_t)0;
            }
}
// column pass A operand (Toeplitz of ky): table[s][lane][16], lane (m, h), byte i: row regRow(h, i) of row-sum tile u + s contributes to output row m of tile u with tap
// 32 s + regRow(h, i) - m
inline void buildColA(const Geom& g, const uint16_t* ky, int8_t* tab)
{
    for (int s = 0; s < g.ksy; s++)
        for (int lane = 0; lane < 64; lane++)
            for (int i = 0; i < 16; i++) {
                const int m = lane & 31, h = lane >> 5, j = 32 * s + regRow(h, i) - m;
                tab[(s * 64 + lane) * 16 + i] = (j >= 0 && j < g.ny) ? (int8_t)ky[j] : (int8_t)0;
            }
}

// false: outside what the kernel covers (a tap beyond int8, more K steps than MAXKS)
inline bool plan(Geom& g, const uint16_t* kx, const uint16_t* ky, uintptr_t srcAddr, size_t sstep, int nframes, int segOverride = 0)
{
    if (g.cn < 1 || g.cn > 4 || g.nx < 1 || g.ny < 1 || g.W < 1 || g.H < 1) return false;
    int sx = 0, sy = 0;
    for (int i = 0; i < g.nx; i++) { if (kx[i] > 127) return false; sx += kx[i]; }
    for (int i = 0; i < g.ny; i++) { if (ky[i] > 127) return false; sy += ky[i]; }
    if (sx > 256 || sy > 256) return false;
    g.WE = g.W * g.cn;
    const int spanX = (g.nx - 1) * g.cn;
    // aligned 16-byte loads when the row pitch allows them and the shift does not cost a K step
    int delta = (sstep % 16 == 0) ? (int)((srcAddr + (uintptr_t)(16 * 1024 * 1024) - (uintptr_t)(g.ax * g.cn)) & 15) : 0;
    if ((32 + delta + spanX + 31) / 32 != (32 + spanX + 31) / 32) delta = 0;
    g.delta = delta;
    g.ksx = (32 + delta + spanX + 31) / 32;
    g.ksy = (32 + g.ny - 1 + 31) / 32;
    if (g.ksx < 2) g.ksx = 2;
    if (g.ksy < 2) g.ksy = 2;
    if (g.ksx > MAXKS || g.ksy > MAXKS) return false;
    g.nchunk = (TW - 32 + 32 * g.ksx) / 16;
    g.P = 16 * (g.nchunk | 1);
    seeds(g, kx, ky);
    // segments: enough workgroups to fill the chip (256 CUs x 2), each segment repeats (KSY - 1) tiles of row sums at its top
    const int nstrips = (g.WE + TW - 1) / TW;
    int nseg = (1024 + nstrips * nframes - 1) / (nstrips * nframes);
    const int maxseg = (g.H + 4 * TR - 1) / (4 * TR);
    if (nseg > maxseg) nseg = maxseg;
    if (nseg < 1) nseg = 1;
    int seg = ((g.H + nseg - 1) / nseg + TR - 1) / TR * TR;
    if (segOverride > 0) seg = (segOverride + TR - 1) / TR * TR;
    g.seg = seg;
    return true;
}

// One 16-byte chunk of the staged block: row r (0 .. 31) of step t of the segment starting at output row y0, chunk c of the row.  Returns false when the chunk is a plain
// aligned-or-not 16-byte load (*ptr set), true when `out` was filled here (rows outside the image under BORDER_CONSTANT, chunks that touch the left / right rim).  The bytes
// are unsigned pixels; the caller flips them to signed.
MX_HD bool stageChunk(const Geom& g, const unsigned char* src, size_t sstep, int X0, int y0, int t, int r, int c, const unsigned char** ptr, unsigned char* out)
{
    const int sy = y0 - g.ay + TR * t + r;                             // ROI row of this row-sum row's source
    const int yy = borderIdx(sy + g.offY, g.fullH, g.border);
    if (yy < 0) { for (int b = 0; b < 16; b++) out[b] = 0; return true; }
    const unsigned char* row = src + (ptrdiff_t)(yy - g.offY) * (ptrdiff_t)sstep;
    const int e0 = X0 - g.ax * g.cn - g.delta + 16 * c;                // ROI element of the chunk's first byte
    const int f0 = e0 + g.offX * g.cn;                                 // the same in the full image
    if (f0 >= 0 && f0 + 16 <= g.fullW * g.cn) { *ptr = row + e0; return false; }
    for (int b = 0; b < 16; b++) {
        const int f = f0 + b;
        const int p = f >= 0 ? f / g.cn : -((-f + g.cn - 1) / g.cn), ch = f - p * g.cn;
        const int q = borderIdx(p, g.fullW, g.border);
        out[b] = q < 0 ? (unsigned char)0 : row[(ptrdiff_t)(q - g.offX) * g.cn + ch];
    }
    return true;
}

} // namespace sepmx
