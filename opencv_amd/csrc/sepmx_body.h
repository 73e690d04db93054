// sepmx_body.h -- the host-visible half of sepmx.hip (the 8-bit Q8.8 separable smoothing on the matrix cores): geometry plan, the two Toeplitz operand tables, the
// staging of one 16-byte chunk and the lane <-> element maps of v_mfma_i32_32x32x32_i8.  Everything here is __host__ __device__ and is what the kernel itself runs;
// the host builds the operand tables with them (and decides what the matrix form does not take), the kernel stages with them.
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
constexpr int MAXKS = 9;         // 32-byte K steps of the column pass: 32 + ny - 1 <= 32 KSY (255 taps: the largest box filter); the kernel exists for KSY in {2, 3, 4, 5, 7, 9}
constexpr int MAXKSX = 13;       // ... of the row pass: 32 + delta + (nx - 1) cn <= 32 KSX -- channels are interleaved elements, so three channels x 129 taps need 13;
                                 // the kernel exists for KSX in {2, 3, 4, 5, 7, 9, 13}, a count in between runs on the next one (the extra steps carry zero weights)
MX_HD int ksxClass(int k) { return k <= 5 ? (k < 2 ? 2 : k) : k <= 7 ? 7 : k <= 9 ? 9 : 13; }
MX_HD int ksyClass(int k) { return k <= 5 ? (k < 2 ? 2 : k) : k <= 7 ? 7 : 9; }

struct Geom {
    int W, H, cn, WE;                          // the ROI; WE = W * cn elements per row
    int fullW, fullH, offX, offY, border;      // the image around it (real pixels outside the ROI are read), cv border code 0 .. 4
    int nx, ny, ax, ay;
    int ksx, ksy;
    int delta;                                 // the staged row starts at ROI element X0 - ax * cn - delta (delta makes the 16-byte loads aligned)
    int shift;                                 // strips start at element 256 s - shift (0 / 32 / 64 / 96): the position against the 128-byte lines at which a staged row piece touches the fewest
    int dma;                                   // 1: rows arrive by asynchronous global -> LDS loads (16-byte aligned chunks), 0: through registers
    int seg;                                   // output rows per segment (a multiple of TR)
    int box, divScale, divDelta, tailStart;    // box filter finish (0: the Gaussian's byte 2; 1: ((s + dd) * ds) >> 23; 2: cvRound(float(s) * scaleF), the row's last (WE % 8) elements in double; 3: saturate(s))
    float scaleF; double scaleD;
    int fast;                                  // the row pitch is >= 512 bytes: a chunk of a row that is neither the parent's first nor its last lies inside the parent's memory
    int sumKy;                                 // the column taps' sum (the seeds of the column pass depend on it)
    int xcd;                                   // 1: workgroup ids are dealt so that the strips of one XCD (linear id mod 8) are neighbours and share their halo columns in its L2
    int ncls;                                  // row-pass operand classes (0 = interior; one per wave whose columns see a left / right border)
    long long span;                            // bytes from the parent image's first byte to one past its last: (fullH - 1) * step + fullW * cn
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
//   row pass     acc = sum kx (s - 128) in [-32768, 32512] (it starts at the inline constant 0: no register moves)  =>  R = acc + 128 P, P = the sum of the taps present
//                acc = 256 Hh + L with Hh = (signed) byte 1, L = (unsigned) byte 0; l = L - 128, both in int8           =>  R = 256 Hh + l + 128 + 128 P
//   column pass  accH = sum ky Hh,  accL = C + sum ky l  =>  (accH << 8) + accL = sum ky R + 2^15 with C = (128 + 128 P) sum(ky) + 2^15: a constant of the COLUMN (P differs
//                where BORDER_CONSTANT drops taps), i.e. of the lane; the value is < 2^24 and the result is its byte 2
MX_HD int colSeed(int sumPresentTapsX, int sumTapsY, int box) { return (128 + 128 * sumPresentTapsX) * sumTapsY + (box ? 0 : 32768); }      // (a box filter wants the plain sum)

// Row-pass B operand of one wave (strip X0, wave w: output elements X0 + 32 w + n): tab[ks][lane][16], lane (n, h), byte i <-> staged column k = 32 ks + 16 h + i of the
// wave's window, which starts at ROI element X0 + 32 w - ax cn - delta.  The LEFT / RIGHT BORDER IS FOLDED INTO THE MATRIX: tap i of output element x reads pixel
// borderInterpolate(x / cn + i - ax) of the full image -- its weight is added at THAT pixel's column (two taps can land on one column under the reflecting rules), taps that
// fall outside under BORDER_CONSTANT are dropped (and leave the column's seed: seed[n] = colSeed(sum of the taps present, sum of ky)).  The staged bytes of columns outside the image
// then never matter (weight 0): staging needs no border logic along x.  false: a folded weight beyond 2 x 127, or a border pixel outside the wave's window (BORDER_WRAP on
// an image wider than the window, reflections in an image narrower than the kernel): the caller hands the call to the vector kernel.  *interior: no tap was moved; *twice: tab2 is not empty.
inline bool buildRowB(const Geom& g, const uint16_t* kx, int sumKy, int X0, int w, int8_t* tab, int8_t* tab2, bool* twice, int* seed, bool* interior)
{
    static thread_local int wt[32 * MAXKSX][32];
    memset(wt, 0, sizeof(int) * 32 * 32 * g.ksx);
    *interior = true; *twice = false;
    const int win0 = X0 + 32 * w - g.ax * g.cn - g.delta;
    for (int n = 0; n < 32; n++) {
        const int xe = X0 + 32 * w + n;
        int present = 0;
        if (xe < 0 || xe >= g.WE) { seed[n] = colSeed(0, sumKy, g.box); *interior = false; continue; }   // not an output: an empty column
        const int px = xe / g.cn, ch = xe - px * g.cn;
        for (int i = 0; i < g.nx; i++) {
            const int pf = px + g.offX + i - g.ax, q = borderIdx(pf, g.fullW, g.border);
            if ((unsigned)pf >= (unsigned)g.fullW) *interior = false;
            if (q < 0) continue;
            const int k = (q - g.offX) * g.cn + ch - win0;
            if (k < 0 || k >= 32 * g.ksx) return false;
            wt[k][n] += kx[i]; present += kx[i];
            if (wt[k][n] > 254) return false;
        }
        seed[n] = colSeed(present, sumKy, g.box);
    }
    // a weight beyond int8 (BORDER_REPLICATE piles up to half the kernel on the rim pixel) is applied in two products: min(w, 127) here, the rest in tab2
    for (int ks = 0; ks < g.ksx; ks++)
        for (int lane = 0; lane < 64; lane++)
            for (int i = 0; i < 16; i++) {
                const int v = wt[32 * ks + 16 * (lane >> 5) + i][lane & 31], a = v > 127 ? 127 : v;
                tab[(ks * 64 + lane) * 16 + i] = (int8_t)a; tab2[(ks * 64 + lane) * 16 + i] = (int8_t)(v - a);
                if (v != a) *twice = true;
            }
    return true;
}
// column pass A operand (Toeplitz of ky): table[s][lane][16], lane (m, h), byte i: row regRow(h, i) of row-sum tile u + s contributes to output row m of tile u with tap
// 32 s + regRow(h, i) - m.  (The top / bottom border needs no folding: a staged row IS the source row borderInterpolate names, or zeros.)
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
inline bool plan(Geom& g, const uint16_t* kx, const uint16_t* ky, uintptr_t srcAddr, size_t sstep, size_t sframe, int nframes, int segOverride = 0, int dmaOverride = -1)
{
    if (g.cn < 1 || g.cn > 4 || g.nx < 1 || g.ny < 1 || g.W < 1 || g.H < 1) return false;
    int sx = 0, sy = 0;
    for (int i = 0; i < g.nx; i++) { if (kx[i] > 127) return false; sx += kx[i]; }
    for (int i = 0; i < g.ny; i++) { if (ky[i] > 127) return false; sy += ky[i]; }
    if (sx > 256 || sy > 256) return false;
    g.WE = g.W * g.cn;
    g.span = (long long)(g.fullH - 1) * (long long)sstep + (long long)g.fullW * g.cn;
    const int spanX = (g.nx - 1) * g.cn;
    // asynchronous staging wants 16-byte aligned chunks: the staged row starts delta elements early (delta < 16) -- when every row start moves by multiples of 16 and
    // the shift does not push the row of taps beyond MAXKS K steps
    const bool alignable = sstep % 16 == 0 && sframe % 16 == 0;
    const int dAl = (int)((srcAddr + (uintptr_t)(16 * 1024 * 1024) - (uintptr_t)(g.ax * g.cn)) & 15);
    g.dma = alignable && (32 + dAl + spanX + 31) / 32 <= MAXKSX;
    if (dmaOverride == 0) g.dma = 0;
    g.delta = g.dma ? dAl : 0;
    if (dmaOverride == 2 && !g.dma) { g.dma = 1; g.delta = 0; }       // (experiment: asynchronous loads from rows that are not 16-byte aligned)
    g.ksx = (32 + g.delta + spanX + 31) / 32;
    g.ksy = (32 + g.ny - 1 + 31) / 32;
    if (g.ksx > MAXKSX) return false;
    g.ksx = ksxClass(g.ksx);
    if (g.ksy > MAXKS) return false;
    g.ksy = ksyClass(g.ksy);
    if (g.ksx == 13 && g.ksy == 9) return false;                     // (the one pair the kernel is not built for: it would spill registers)
    g.sumKy = sy;
    g.fast = sstep >= 512;
    // A staged row piece is 224 + 32 KSX bytes from element X0 - ax cn - delta.  With strips at multiples of 256 it starts 16 .. 64 bytes before a 128-byte line and touches
    // one line more than it has to (19 taps: four lines, 512 bytes, for 256 bytes of output: profiles/r06_sepmx.txt); strips moved left by whole 32-column blocks put it
    // where it touches the fewest (the first strip's leading blocks then have no outputs).
    {
        const int piece = TW - 32 + 32 * g.ksx;
        int best = 0, bestLines = 1 << 30;
        for (int sh = 0; sh < 128; sh += 32) {
            const int a0 = (int)((srcAddr + (uintptr_t)(16 * 1024 * 1024) - (uintptr_t)(g.ax * g.cn + g.delta + sh)) & 127), lines = (a0 + piece + 127) / 128;
            if (lines < bestLines) { bestLines = lines; best = sh; }
        }
        g.shift = best;
    }
    // segments: enough workgroups to fill the chip (256 CUs x 2 x 2), each segment repeats (KSY - 1) tiles of row sums at its top
    const int nstrips = (g.WE + g.shift + TW - 1) / TW;
    int nseg = (1024 + nstrips * nframes - 1) / (nstrips * nframes);
    const int maxseg = (g.H + 4 * TR - 1) / (4 * TR);
    if (nseg > maxseg) nseg = maxseg;
    if (nseg < 1) nseg = 1;
    int seg = ((g.H + nseg - 1) / nseg + TR - 1) / TR * TR;
    if (segOverride > 0) seg = (segOverride + TR - 1) / TR * TR;
    g.seg = seg;
    return true;
}

// One 16-byte chunk of a staged row: source row (borderInterpolate of ROI row sy; -1 = zeros) and whether all 16 bytes lie inside the parent image's memory (then ONE
// load; what the bytes of columns outside the image hold does not matter).  Only chunks that would reach before the first / past the last byte of the parent -- first and
// last rows -- are assembled byte by byte.
enum { CH_ZERO = 0, CH_LOAD = 1, CH_BYTES = 2 };
MX_HD int chunkKind(const Geom& g, const unsigned char* src, size_t sstep, int sy, int e0, const unsigned char** ptr, long long* rel)
{
    const int yy = borderIdx(sy + g.offY, g.fullH, g.border);
    if (yy < 0) return CH_ZERO;
    const long long a = (long long)yy * (long long)sstep + e0 + g.offX * g.cn;            // from the parent's first byte
    *ptr = src + (ptrdiff_t)(yy - g.offY) * (ptrdiff_t)sstep + e0;
    *rel = a;
    return (a >= 0 && a + 16 <= g.span) ? CH_LOAD : CH_BYTES;
}

} // namespace sepmx
