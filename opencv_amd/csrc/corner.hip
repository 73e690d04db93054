// corner.hip -- rows a10/a11/a12 of SURVEY.md §8: cv::cornerHarris / cornerMinEigenVal, cv::goodFeaturesToTrack,
// cv::pyrDown / cv::buildPyramid.
//
// Reference semantics:
//   cornerEigenValsVecs (corner.cpp:237-322): Dx, Dy = Sobel(src, CV_32F, scale = 1/(2^(ks-1) * bs * [255 if 8U])),
//     cov = (dx^2, dx*dy, dy^2) in float, boxFilter(cov, bs x bs, normalize=false) with double sums and the border
//     rule applied to COV (i.e. to the derivative positions), then calcHarris :104-155 R = (a*c - b*b) - (k*(a+c))*(a+c)
//     or calcMinEigenVal :52-100 (a+c)/2 - sqrt(((a-c)/2)^2 + b^2), all float.
//   The reference makes ~5 full-image passes (70 B/pixel of traffic); here everything between the 8-bit source and
//   the float response stays inside one workgroup: source tile (+halo) -> LDS, both derivative planes -> LDS, box
//   sums + response from LDS.  HBM traffic = 1 B read + 4 B written per pixel (1080p: 10.4 MB/frame).
//   goodFeaturesToTrack (featureselect.cpp:382-548): max -> THRESH_TOZERO -> 3x3 dilate equality test on the GPU,
//     candidates compacted to a list; the sort (value desc, address desc) and the min-distance grid stay on the host
//     as in the reference's own OpenCL path (:75-360).
//   pyrDown (pyramids.cpp:883-1037): 5x5 [1 4 6 4 1]^2 at even pixels, (S + 128) >> 8 for integers, * 1/256 for float,
//     including the tabR column stepping of :897-910 for non-default dsize.
#include "rt.h"
#include "roll.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <type_traits>

using namespace mi355;

namespace mi355 {
size_t sortKeysDescTemp(unsigned n);                                                                             // gftt_sort.hip (rocPRIM)
bool sortKeysDesc(void* temp, size_t bytes, const unsigned long long* in, unsigned long long* out, unsigned n, hipStream_t st);
}

namespace {

enum { D8U = MI355CV_8U, D16U = MI355CV_16U, D16S = MI355CV_16S, D32F = MI355CV_32F };

// ---------------------------------------------------------------------------------- pyrDown
__global__ __launch_bounds__(256) void k_pyrdown(const uchar* __restrict__ src, size_t sstep, size_t sframe, int sw, int sh,
                                                 uchar* __restrict__ dst, size_t dstep, size_t dframe, int dw, int dh,
                                                 int depth, int cn, int mL, int mT, int mR, int mB, int border)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= dw * cn || y >= dh) return;
    src += (size_t)blockIdx.z * sframe; dst += (size_t)blockIdx.z * dframe;
    const int x = e / cn, c = e - x * cn;
    const int fullW = mL + sw + mR, fullH = mT + sh + mB;
    int width0 = (sw - 3) / 2 + 1; if (width0 > dw) width0 = dw;
    const int cx = x < width0 ? 2 * x : 2 * width0 + (x - width0);      // tabR stepping, pyramids.cpp:897-910
    int xs[5];
#pragma unroll
    for (int i = 0; i < 5; i++) xs[i] = (mi355_borderInterpolate(cx + i - 2 + mL, fullW, border) - mL) * cn + c;
    uchar* drow = dst + (size_t)y * dstep;
    if (depth == D32F) {
        float rows[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int yy = mi355_borderInterpolate(2 * y + j - 2 + mT, fullH, border) - mT;
            const float* r = reinterpret_cast<const float*>(src + (ptrdiff_t)yy * (ptrdiff_t)sstep);
            const float p0 = r[xs[0]], p1 = r[xs[1]], p2 = r[xs[2]], p3 = r[xs[3]], p4 = r[xs[4]];
            float t = __fmul_rn(p2, 6.f);
            t = __fadd_rn(t, __fmul_rn(__fadd_rn(p1, p3), 4.f)); t = __fadd_rn(t, p0); t = __fadd_rn(t, p4);
            rows[j] = t;
        }
        float t = __fmul_rn(rows[2], 6.f);
        t = __fadd_rn(t, __fmul_rn(__fadd_rn(rows[1], rows[3]), 4.f)); t = __fadd_rn(t, rows[0]); t = __fadd_rn(t, rows[4]);
        reinterpret_cast<float*>(drow)[e] = __fmul_rn(t, 1.f / 256);
        return;
    }
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int yy = mi355_borderInterpolate(2 * y + j - 2 + mT, fullH, border) - mT;
        const uchar* r = src + (ptrdiff_t)yy * (ptrdiff_t)sstep;
        int v[5];
#pragma unroll
        for (int i = 0; i < 5; i++)
            v[i] = depth == D8U ? (int)r[xs[i]] : depth == D16U ? (int)reinterpret_cast<const unsigned short*>(r)[xs[i]]
                                                               : (int)reinterpret_cast<const short*>(r)[xs[i]];
        const int rs = v[2] * 6 + (v[1] + v[3]) * 4 + v[0] + v[4];
        const int wj = j == 2 ? 6 : (j == 1 || j == 3) ? 4 : 1;
        acc += wj * rs;
    }
    const int o = (acc + 128) >> 8;
    if (depth == D8U) drow[e] = (uchar)o;
    else if (depth == D16U) reinterpret_cast<unsigned short*>(drow)[e] = (unsigned short)o;
    else reinterpret_cast<short*>(drow)[e] = (short)o;
}


// pyrDown, rolling path: CV_8UC1, default destination size, even 16-aligned width.  The roll.h skeleton with two source rows
// per output row.  Horizontal pass on packed even/odd byte planes (two 16-bit sums per dword, h <= 16*255), vertical pass on
// the last five h rows (v <= 16*4080 fits 16 bits), (v + 128) >> 8 -- the integer arithmetic of PyrDownInvoker
// (pyramids.cpp:873-1040), so bit-exact.  5 bytes of traffic per output pixel (4 read, 1 written).
template <int D>
__global__ __launch_bounds__(256) void k_pyrdown_roll(const uchar* __restrict__ src, size_t sstep, size_t sframe,
                                                      uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                      int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes, int border, int alt)
{
    typedef roll::Ctx<2, 2, 1, 16> Cx;
    typedef typename Cx::RawT RawT;
    Cx cx;
    if (!cx.init(src, sstep, sframe, W, H, nchunks, nstrips, segRows, nseg, nframes, border, alt)) return;
    dst += (size_t)cx.frame * dframe;
    struct HRow { uint32_t h[4]; };                    // h[i] = (hsum of output 2i, hsum of output 2i+1) as 2 x u16
    auto hpass = [&](HRow& o, const RawT& raw) {
        uint32_t X[Cx::NW];                            // X[0] = columns x0-4..x0-1, X[1..4] own, X[5] = x0+16..x0+19
        cx.window(X, raw);
        uint32_t ev[6], od[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { ev[i] = X[i] & 0x00ff00ffu; od[i] = (X[i] >> 8) & 0x00ff00ffu; }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t em = __builtin_amdgcn_alignbit(ev[i + 1], ev[i], 16);      // (E[2i-1], E[2i])
            const uint32_t ep = __builtin_amdgcn_alignbit(ev[i + 2], ev[i + 1], 16);  // (E[2i+1], E[2i+2])
            const uint32_t om = __builtin_amdgcn_alignbit(od[i + 1], od[i], 16);      // (O[2i-1], O[2i])
            o.h[i] = sum146(em + ep, om + od[i + 1], ev[i + 1]);
        }
    };
    // The segment's output rows oy0 .. oy0 + nout - 1 in walking order: downwards from source row B = y0, or (alt: every other segment, so that two segments meet at
    // their common boundary at the same time and the four rows they both read are in L2 for the second one) upwards from B = the centre row of the last output.
    // The taps are symmetric, so the upward walk is the same recurrence on rows B, B - 1, B - 2, ...
    const int oy0 = cx.y0 >> 1, nout = (cx.nrows + 1) >> 1;
    const int dir = cx.up ? -1 : 1, B = cx.up ? 2 * (oy0 + nout - 1) : cx.y0;
    auto rowAt = [&](int k) { return min(max(B + dir * k, -2), H + 1); };            // k-th source row from B in walking direction, kept inside what rowIdx resolves
    HRow h0, h1, h2;
    {
        RawT r0, r1, r2; int v;
        cx.issueImg(r0, rowAt(-2), v); cx.issueImg(r1, rowAt(-1), v); cx.issueImg(r2, rowAt(0), v);
        hpass(h0, r0); hpass(h1, r1); hpass(h2, r2);
    }
    // ring of D source rows in flight per wave (D / 2 output rows ahead): small levels are a few thousand waves in all, so the bytes in flight
    // per wave, not the wave count, is what covers the memory latency there
    RawT raw[D]; int rv;
#pragma unroll
    for (int u = 0; u < D; u++) cx.issueImg(raw[u], rowAt(1 + u), rv);
    for (int t = 0; t < nout; t += D / 2) {
#pragma unroll
        for (int u = 0; u < D / 2; u++) {
            if (t + u < nout) {
                HRow h3, h4;
                hpass(h3, raw[2 * u]); hpass(h4, raw[2 * u + 1]);
                const int nxt = 2 * (t + u) + 1 + D;
                cx.issueImg(raw[2 * u], rowAt(nxt), rv);
                cx.issueImg(raw[2 * u + 1], rowAt(nxt + 1), rv);
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t v = sum146(h0.h[i] + h4.h[i], h1.h[i] + h3.h[i], h2.h[i]);
                    w[i] = ((v + 0x00800080u) >> 8) & 0x00ff00ffu;
                }
                if (cx.active) {
                    uint2 ov;
                    ov.x = __builtin_amdgcn_perm(w[1], w[0], 0x06040200u);
                    ov.y = __builtin_amdgcn_perm(w[3], w[2], 0x06040200u);
                    *reinterpret_cast<uint2*>(dst + (size_t)(cx.up ? oy0 + nout - 1 - (t + u) : oy0 + t + u) * dstep + 8 * (size_t)cx.c) = ov;
                }
                h0 = h2; h1 = h3; h2 = h4;
            }
        }
    }
}

// Three pyramid levels in one launch (CV_8UC1, default sizes): buildPyramid's small levels are a few thousand waves each, so three dependent
// launches cost three launch + load latencies.  Here a workgroup owns a T x T tile of the LAST level and walks back what it needs: (2T+3)^2 of
// the level before, (4T+9)^2 of the first produced level, (8T+21)^2 source pixels (T = 16: 149 x 156 bytes of LDS, 1.4x redundant source
// reads, served by L2 / Infinity Cache where the previous launch just left that level).  All tiles live in LDS in UNCLAMPED coordinates:
// the loader fills source positions outside the image through the border rule, and after each level the entries outside that level's image
// are filled from their border-mapped twins, so the 5x5 passes themselves never see a border.  A level is produced by column-pair threads
// walking down the tile rows (two aligned LDS dwords per source row give the 7 bytes of an output pair -- the tile origin is 2*o-2, which
// puts byte 4j at the left tap of pair j), keeping the last five horizontal sums in registers: the integer arithmetic of PyrDownInvoker
// (pyramids.cpp:873-1040) in the packed 2 x u16 form of k_pyrdown_roll, hence bit-exact.  Every workgroup stores the part of each level its
// last-level tile is the parent of: each output pixel is written exactly once.
constexpr int P3_T = 16;
constexpr int P3_NC = P3_T, P3_RC = P3_T;                              // tile columns (even) x rows per level: C = last, B, A = first produced, S = source
constexpr int P3_NB = 2 * P3_NC + 4, P3_RB = 2 * P3_RC + 3;
constexpr int P3_NA = 2 * P3_NB + 4, P3_RA = 2 * P3_RB + 3;
constexpr int P3_NS = 2 * P3_NA + 4, P3_RS = 2 * P3_RA + 3;
static_assert(P3_NS % 4 == 0 && P3_NA % 4 == 0 && P3_NB % 4 == 0 && P3_NC % 4 == 0, "tile rows are read as dwords");

struct Pyr3Args {
    const uchar* src; size_t sstep, sframe; int sw, sh;
    uchar* d[3]; size_t dstep[3], dframe[3]; int dw[3], dh[3];
    int border, tilesX, tilesY;
};

// the border rules k_pyr3 serves (REPLICATE, REFLECT, REFLECT_101), without the generic function's WRAP division
__device__ __forceinline__ int pyr3Map(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (border == B_REPLICATE) return p < 0 ? 0 : len - 1;
    const int d = border == B_REFLECT_101;
    do { p = p < 0 ? -p - 1 + d : 2 * len - 1 - p - d; } while ((unsigned)p >= (unsigned)len);
    return p;
}

// one level inside LDS: O (ROUT x NOUT bytes) from S (pitch SP bytes, 2*ROUT+3 rows): pair p of segment sg walks rows [r0, r1)
template <int NOUT, int ROUT, int SP, int OP>
__device__ __forceinline__ void pyr3Level(const uchar* S, uchar* O, int tid)
{
    constexpr int NP = NOUT / 2;
    constexpr int NSEG = 256 / NP < ROUT ? 256 / NP : ROUT;
    constexpr int RSEG = (ROUT + NSEG - 1) / NSEG;
    if (tid >= NP * NSEG) return;
    const int p = tid % NP, sg = tid / NP;
    const int r0 = sg * RSEG, r1 = min(ROUT, r0 + RSEG);
    auto hsum = [&](int srow) -> uint32_t {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(S + srow * SP + 4 * p);   // bytes b0..b7 = the source columns from the left tap of output 2p on
        uint2 d; d.x = q[0]; d.y = q[1];
        const uint32_t ev0 = d.x & 0x00ff00ffu, od0 = (d.x >> 8) & 0x00ff00ffu;        // (b0, b2), (b1, b3)
        const uint32_t ev1 = d.y & 0x00ff00ffu, od1 = (d.y >> 8) & 0x00ff00ffu;        // (b4, b6), (b5, b7)
        const uint32_t mid = __builtin_amdgcn_alignbit(ev1, ev0, 16);                  // (b2, b4)
        const uint32_t odm = __builtin_amdgcn_alignbit(od1, od0, 16);                  // (b3, b5)
        return sum146(ev0 + ev1, od0 + odm, mid);                                        // (b0+4b1+6b2+4b3+b4, b2+4b3+6b4+4b5+b6)
    };
    if (r0 >= r1) return;
    uint32_t h0 = hsum(2 * r0), h1 = hsum(2 * r0 + 1), h2 = hsum(2 * r0 + 2);
    for (int r = r0; r < r1; r++) {
        const uint32_t h3 = hsum(2 * r + 3), h4 = hsum(2 * r + 4);
        const uint32_t v = sum146(h0 + h4, h1 + h3, h2);
        const uint32_t w = ((v + 0x00800080u) >> 8) & 0x00ff00ffu;
        *reinterpret_cast<unsigned short*>(O + r * OP + 2 * p) = (unsigned short)((w & 0xffu) | ((w >> 8) & 0xff00u));
        h0 = h2; h1 = h3; h2 = h4;
    }
}

// a 5-tap pass over a level only ever asks for the two positions next to each image side: columns -2, -1, W, W+1 and rows -2, -1, H, H+1 of
// the tile (where the tile holds them) take the values of their border-mapped twins -- columns first, then whole rows, so the corners follow.
// The twins lie inside the tile for every position a later level uses; the index is clamped into the tile for the others.
template <int N, int R, int PITCH>
__device__ __forceinline__ void pyr3Border(uchar* Tl, int ox, int oy, int W, int H, int border, int tid, bool rows)
{
    if (ox >= 0 && oy >= 0 && ox + N <= W && oy + R <= H) return;                     // uniform per workgroup
    for (int i = tid; i < 4 * R; i += 256) {
        const int r = i >> 2, xi = i & 3;
        const int x = xi < 2 ? xi - 2 : W + xi - 2, c = x - ox;
        if ((unsigned)c >= (unsigned)N) continue;
        const int xm = min(max(pyr3Map(x, W, border) - ox, 0), N - 1);
        Tl[r * PITCH + c] = Tl[r * PITCH + xm];
    }
    if (!rows) return;
    __syncthreads();
    for (int i = tid; i < 4 * N; i += 256) {
        const int c = i >> 2, yi = i & 3;
        const int y = yi < 2 ? yi - 2 : H + yi - 2, r = y - oy;
        if ((unsigned)r >= (unsigned)R) continue;
        const int ym = min(max(pyr3Map(y, H, border) - oy, 0), R - 1);
        Tl[r * PITCH + c] = Tl[ym * PITCH + c];
    }
}

// the workgroup's own part of a level (the children of its last-level tile) from the LDS tile to the image, two bytes per lane
template <int PITCH>
__device__ __forceinline__ void pyr3Store(const uchar* Tl, int ox, int oy, int x0, int y0, int nx, int ny, int W, int H, uchar* dst, size_t dstep, int tid)
{
    const int npair = nx / 2;
    for (int i = tid; i < npair * ny; i += 256) {
        const int r = i / npair, c = 2 * (i - r * npair);
        const int x = x0 + c, y = y0 + r;
        if (y >= H || x >= W) continue;
        const unsigned short v = *reinterpret_cast<const unsigned short*>(Tl + (y - oy) * PITCH + (x - ox));
        uchar* d = dst + (size_t)y * dstep + x;
        if (x + 1 < W) *reinterpret_cast<unsigned short*>(d) = v;
        else *d = (uchar)v;
    }
}

__global__ __launch_bounds__(256) void k_pyr3(Pyr3Args a)
{
    __shared__ __attribute__((aligned(16))) uchar tS[P3_RS * P3_NS];
    __shared__ __attribute__((aligned(16))) uchar tA[P3_RA * P3_NA];
    __shared__ __attribute__((aligned(16))) uchar tB[P3_RB * P3_NB];
    __shared__ __attribute__((aligned(16))) uchar tC[P3_RC * P3_NC];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tx = tile % a.tilesX, ty = tile / a.tilesX;
    const uchar* src = a.src + (size_t)blockIdx.y * a.sframe;
    const int xC = tx * P3_T, yC = ty * P3_T;
    const int oxB = 2 * xC - 2, oyB = 2 * yC - 2, oxA = 2 * oxB - 2, oyA = 2 * oyB - 2, oxS = 2 * oxA - 2, oyS = 2 * oyA - 2;
    // source tile: dword k of tile row r holds source columns oxS + 4k .. +3 of row oyS + r.  Rows go through the border rule here; a dword
    // takes the bytes that lie inside the image from one unaligned 8-byte load clamped into the row, and the columns outside the image are
    // then filled inside LDS from their mapped twins (pyr3Border).  All loads of a thread are issued before its first LDS write: a rolled
    // loop would pay one cache latency per pass, 25 of them.
    constexpr int NDW = P3_NS / 4, RPP = 256 / NDW, ITER = (P3_RS + RPP - 1) / RPP;       // RPP tile rows per pass
    {
        typedef unsigned long long u64u __attribute__((aligned(1)));
        const int k = tid % NDW, rsub = tid / NDW;                                       // this thread's dword column; threads >= RPP * NDW idle
        const int x = oxS + 4 * k;
        const int xc = min(max(x, 0), a.sw - 8);                                          // sw >= 8
        const int d = x - xc;                                                             // > 0: the dword starts right of the loaded bytes
        const uchar* colp = src + xc;
        const bool rowsInside = oyS >= 0 && oyS + P3_RS <= a.sh;                         // uniform per workgroup
        unsigned long long q[ITER];
#pragma unroll
        for (int it = 0; it < ITER; it++) {
            const int r = min(it * RPP + rsub, P3_RS - 1);
            const int y = rowsInside ? oyS + r : pyr3Map(oyS + r, a.sh, a.border);
            q[it] = *reinterpret_cast<const u64u*>(colp + (size_t)y * a.sstep);
        }
#pragma unroll
        for (int it = 0; it < ITER; it++) {
            const int r = it * RPP + rsub;
            const uint32_t v = d >= 8 || d <= -8 ? 0u : d >= 0 ? (uint32_t)(q[it] >> (8 * d)) : (uint32_t)(q[it] << (-8 * d));
            if (rsub < RPP && r < P3_RS) reinterpret_cast<uint32_t*>(tS)[r * NDW + k] = v;
        }
    }
    __syncthreads();
    pyr3Border<P3_NS, P3_RS, P3_NS>(tS, oxS, 0, a.sw, P3_RS, a.border, tid, false);      // columns only: the rows are already mapped
    __syncthreads();
    pyr3Level<P3_NA, P3_RA, P3_NS, P3_NA>(tS, tA, tid);
    __syncthreads();
    pyr3Border<P3_NA, P3_RA, P3_NA>(tA, oxA, oyA, a.dw[0], a.dh[0], a.border, tid, true);
    __syncthreads();
    pyr3Level<P3_NB, P3_RB, P3_NA, P3_NB>(tA, tB, tid);
    __syncthreads();
    pyr3Border<P3_NB, P3_RB, P3_NB>(tB, oxB, oyB, a.dw[1], a.dh[1], a.border, tid, true);
    __syncthreads();
    pyr3Level<P3_NC, P3_RC, P3_NB, P3_NC>(tB, tC, tid);
    __syncthreads();
    pyr3Store<P3_NA>(tA, oxA, oyA, 4 * xC, 4 * yC, 4 * P3_T, 4 * P3_T, a.dw[0], a.dh[0], a.d[0] + (size_t)blockIdx.y * a.dframe[0], a.dstep[0], tid);
    pyr3Store<P3_NB>(tB, oxB, oyB, 2 * xC, 2 * yC, 2 * P3_T, 2 * P3_T, a.dw[1], a.dh[1], a.d[1] + (size_t)blockIdx.y * a.dframe[1], a.dstep[1], tid);
    pyr3Store<P3_NC>(tC, xC, yC, xC, yC, P3_T, P3_T, a.dw[2], a.dh[2], a.d[2] + (size_t)blockIdx.y * a.dframe[2], a.dstep[2], tid);
}

// levels l+1..l+3 from level l in one launch where k_pyr3's geometry applies (CV_8UC1, default sizes, even destination addresses, a last level
// of at least 4 x 4, borders that map into the neighbourhood); MI355CV_PYR_FUSE=0 keeps the level-by-level launches
bool launchPyr3(const uchar* s, size_t ss, size_t sf, int w, int h, uchar* const* d, const size_t* dstep, const size_t* dframe, int nframes, int depth, int cn,
                int border, hipStream_t st)
{
    // measured (profiles/r03_pyramid.txt, 256 x 1080p): the three small levels take 118 us fused against 71 us as three launches of the rolling kernel --
    // a batch has waves enough per level, what the fusion buys is launch + load latency, which only a call on one or two frames is short of
    const char* e = getenv("MI355CV_PYR_FUSE");
    if (e ? !atoi(e) : nframes >= 4) return false;
    if (depth != D8U || cn != 1 || !(border == B_REPLICATE || border == B_REFLECT || border == B_REFLECT_101)) return false;
    Pyr3Args a; a.src = s; a.sstep = ss; a.sframe = sf; a.sw = w; a.sh = h; a.border = border;
    int pw = w, ph = h;
    for (int l = 0; l < 3; l++) {
        a.d[l] = d[l]; a.dstep[l] = dstep[l]; a.dframe[l] = nframes == 1 ? 0 : dframe[l];
        a.dw[l] = (pw + 1) / 2; a.dh[l] = (ph + 1) / 2; pw = a.dw[l]; ph = a.dh[l];
        if ((((uintptr_t)d[l]) | dstep[l] | a.dframe[l]) & 1) return false;
    }
    if (a.dw[2] < 4 || a.dh[2] < 4 || w < 8 || h < 8) return false;
    a.tilesX = divUp(a.dw[2], P3_T); a.tilesY = divUp(a.dh[2], P3_T);
    hipLaunchKernelGGL(k_pyr3, dim3(a.tilesX * a.tilesY, nframes), dim3(256), 0, st, a);
    return true;
}

// one level on device-resident images: the rolling kernel where its geometry applies, the per-output kernel otherwise
void launchPyrDown(const uchar* ds, size_t dss, size_t sframe, int sw, int sh, uchar* dd, size_t dds, size_t dframe, int dw, int dh, int nframes,
                   int depth, int cn, int mL, int mT, int mR, int mB, int border, hipStream_t st)
{
    if (depth == D8U && cn == 1 && !mL && !mT && !mR && !mB && dw * 2 == sw && dh == (sh + 1) / 2 && sh >= 2 && border != B_WRAP &&
        (((uintptr_t)dd | dds | dframe) & 7) == 0 && sw % 16 == 0 && (((uintptr_t)ds | dss | sframe) & 15) == 0 &&
        roll::eligible(ds, dss, sframe, ds, dss, sframe, sw, 1, 2, border)) {
        // two source rows per output row; short segments when the batch is small so that every SIMD still gets several waves (the 4 halo
        // rows a segment re-reads come from L2: its vertical neighbours run at the same time)
        roll::Geom g = roll::geometry(sw, sh, 1, nframes, 32, 8, 16, 4096);
        if (g.seg & 1) { g.seg++; g.nseg = divUp(sh, g.seg); g.blocks = (unsigned)(((long long)g.nstrips * g.nseg * nframes + 3) / 4); }
        // (8 source rows in flight per wave: 4 and 12 measured the same or slower, profiles/r03_pyramid.txt)
        // neighbouring segments walk towards / away from each other (roll.h `alt`), partners one turn of the XCD round-robin apart; MI355CV_PYR_ALT=0: all downwards
        static const int altEnv = [] { const char* v = getenv("MI355CV_PYR_ALT"); return v ? atoi(v) : -1; }();
        const int alt = altEnv == 0 ? 0 : altEnv > 0 ? altEnv : (32 % g.nstrips == 0 ? std::max(32 / g.nstrips, 2) : 8);
        hipLaunchKernelGGL(k_pyrdown_roll<8>, dim3(g.blocks), dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, sw, sh, g.nchunks, g.nstrips, g.seg, g.nseg, nframes, border, alt);
        return;
    }
    dim3 grid(divUp(dw * cn, 64), divUp(dh, 4), nframes);
    hipLaunchKernelGGL(k_pyrdown, grid, dim3(256), 0, st, ds, dss, sframe, sw, sh, dd, dds, dframe, dw, dh, depth, cn, mL, mT, mR, mB, border);
}

int runPyrDown(const char* entry, const uchar* src, size_t sstep, size_t sframe, int sw, int sh, uchar* dst, size_t dstep, size_t dframe,
               int dw, int dh, int nframes, int depth, int cn, int mL, int mT, int mR, int mB, int border)
{
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    border &= ~MI355CV_BORDER_ISOLATED;
    if (border == B_CONSTANT || border < 0 || border > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "border == B_CONSTANT || border < 0 || border > B_REFLECT_101");   // pyramids.cpp:1352 forbids CONSTANT
    if (!(depth == D8U || depth == D16U || depth == D16S || depth == D32F) || cn < 1 || cn > 4) return mi355::declined(__func__, __LINE__, "!(depth == D8U || depth == D16U || depth == D16S || depth == D32F) || cn < 1 || cn > 4");
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || abs(dw * 2 - sw) > 2 || abs(dh * 2 - sh) > 2) return mi355::declined(__func__, __LINE__, "sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || abs(dw * 2 - sw) > 2 || abs(dh * 2 - sh) > 2");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src, (size_t)sw * sh, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src, (size_t)sw * sh, minPixels())");
    const int e = depth == D8U ? 1 : depth == D32F ? 4 : 2;
    size_t dss = sstep, dds = dstep;
    const uchar* ds = src; uchar* dd = dst;
    if (nframes == 1) {
        const uchar* top = src - (ptrdiff_t)mT * (ptrdiff_t)sstep - (ptrdiff_t)mL * cn * e;
        const uchar* dtop = stg.in(top, sstep, (size_t)(mL + sw + mR) * cn * e, mT + sh + mB, &dss);
        dd = stg.out(dst, dstep, (size_t)dw * cn * e, dh, &dds);
        if (!dtop || !dd) return mi355::declined(__func__, __LINE__, "!dtop || !dd");
        ds = dtop + (size_t)mT * dss + (size_t)mL * cn * e;
    } else if (!isDevicePtr(src) || !isDevicePtr(dst)) return mi355::declined(__func__, __LINE__, "!isDevicePtr(src) || !isDevicePtr(dst)");
    launchPyrDown(ds, dss, sframe, sw, sh, dd, dds, dframe, dw, dh, nframes, depth, cn, mL, mT, mR, mB, border, stream());
    return stg.finish(entry);
}

// ---------------------------------------------------------------------------------- fused corner response
constexpr int CT_X = 64, CT_Y = 16;     // outputs per workgroup

struct CornerArgs {
    int W, H, sdepth, bs, ax, ay, border, harris;
    float kf;
    // separable derivative taps as cv::Sobel / cv::Scharr generate them (deriv.cpp:55-162, :432-439)
    float dxRow[8], dxCol[8], dyRow[8], dyCol[8];
    int nRow, nCol;                     // tap counts of the row / column kernels (same for Dx and Dy except ksize == 1)
    int dxNRow, dxNCol, dyNRow, dyNCol;
    int rx, ry;                         // halo radii of the source tile
};

// Separable derivative evaluation inside the tile, in the reference's association order:
//   row pass   (RowFilter, filter.simd.hpp:2386):  r = k0*v0; r = fma(ki, vi, r)
//   column pass (SymmColumnFilter :2679-2751):      symmetric  s = fma(kc, r_c, 0); s = fma(k_{c+j}, r_{c+j} + r_{c-j}, s)
//                                                   antisymm.  s = 0;              s = fma(k_{c+j}, r_{c+j} - r_{c-j}, s)
__device__ __forceinline__ float rowPass(const float* row, const float* k, int n)
{
    float s = k[0] * row[0];
    for (int i = 1; i < n; i++) s = __builtin_fmaf(k[i], row[i], s);
    return s;
}
__device__ __forceinline__ float colPass(const float* col, int stride, const float* k, int n, bool asym)
{
    const int c = n / 2;
    float s = asym ? 0.f : __builtin_fmaf(k[c], col[c * stride], 0.f);
    for (int j = 1; j <= c; j++)
        s = __builtin_fmaf(k[c + j], asym ? col[(c + j) * stride] - col[(c - j) * stride] : col[(c + j) * stride] + col[(c - j) * stride], s);
    return s;
}

__global__ __launch_bounds__(256) void k_corner_fused(const uchar* __restrict__ src, size_t sstep, size_t sframe,
                                                      uchar* __restrict__ dst, size_t dstep, size_t dframe, CornerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    src += (size_t)blockIdx.z * sframe; dst += (size_t)blockIdx.z * dframe;
    const int X0 = blockIdx.x * CT_X, Y0 = blockIdx.y * CT_Y;
    const int PW = CT_X + a.bs - 1, PH = CT_Y + a.bs - 1;          // derivative planes: positions [X0-ax, X0+CT_X-1+bx]
    const int SW = PW + 2 * a.rx, SH = PH + 2 * a.ry;              // source tile: plane region +- (rx, ry)
    const int rxd = a.dxNRow / 2, rxs = a.dyNRow / 2;              // radii of the two row kernels
    float* S = lds;                                                // SH x SW   source, border-extended
    float* Rd = S + SW * SH;                                       // SH x PW   row pass with the x-derivative taps  (for Dx)
    float* Rs = Rd + PW * SH;                                      // SH x PW   row pass with the x-smoothing taps   (for Dy)
    float* Pdx = Rs + PW * SH;                                     // PH x PW
    float* Pdy = Pdx + PW * PH;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    // 1. source tile: S(p) = src(borderInterpolate(p)); BORDER_CONSTANT contributes 0
    const int sx0 = X0 - a.ax - a.rx, sy0 = Y0 - a.ay - a.ry;
    for (int lx = tx; lx < SW; lx += 64) {
        const int xx = mi355_borderInterpolate(sx0 + lx, a.W, a.border);
        for (int ly = ty; ly < SH; ly += 4) {
            const int yy = mi355_borderInterpolate(sy0 + ly, a.H, a.border);
            float v = 0.f;
            if (yy >= 0 && xx >= 0) {
                const uchar* row = src + (size_t)yy * sstep;
                v = a.sdepth == D8U ? (float)row[xx] : reinterpret_cast<const float*>(row)[xx];
            }
            S[ly * SW + lx] = v;
        }
    }
    __syncthreads();
    // 2. row passes over every staged row, for the PW plane columns
    for (int lx = tx; lx < PW; lx += 64)
        for (int ly = ty; ly < SH; ly += 4) {
            const float* row = S + ly * SW + lx + a.rx;
            Rd[ly * PW + lx] = rowPass(row - rxd, a.dxRow, a.dxNRow);
            Rs[ly * PW + lx] = rowPass(row - rxs, a.dyRow, a.dyNRow);
        }
    __syncthreads();
    // 3. column passes -> derivative planes (only in-image positions are ever read back)
    const int ryd = a.dxNCol / 2, rys = a.dyNCol / 2;
    for (int lx = tx; lx < PW; lx += 64)
        for (int ly = ty; ly < PH; ly += 4) {
            Pdx[ly * PW + lx] = colPass(Rd + (ly + a.ry - ryd) * PW + lx, PW, a.dxCol, a.dxNCol, false);
            Pdy[ly * PW + lx] = colPass(Rs + (ly + a.ry - rys) * PW + lx, PW, a.dyCol, a.dyNCol, a.dyNCol > 1);
        }
    __syncthreads();
    // 4. box sums of the products (double, as RowSum<float,double>/ColumnSum<double,float>) + response
    const int x = X0 + tx;
    if (x >= a.W) return;
    int lxs[16];
    for (int i = 0; i < a.bs; i++) {
        int q = mi355_borderInterpolate(x - a.ax + i, a.W, a.border);
        lxs[i] = q < 0 ? -1 : min(max(q - (X0 - a.ax), 0), PW - 1);
    }
    for (int oy = ty; oy < CT_Y; oy += 4) {
        const int y = Y0 + oy;
        if (y >= a.H) break;
        double sxx = 0, sxy = 0, syy = 0;
        for (int j = 0; j < a.bs; j++) {
            int q = mi355_borderInterpolate(y - a.ay + j, a.H, a.border);
            if (q < 0) continue;
            const int ly = min(max(q - (Y0 - a.ay), 0), PH - 1);
            double rxx = 0, rxy = 0, ryy = 0;
            for (int i = 0; i < a.bs; i++) {
                if (lxs[i] < 0) continue;
                const float dx = Pdx[ly * PW + lxs[i]], dy = Pdy[ly * PW + lxs[i]];
                rxx += (double)__fmul_rn(dx, dx); rxy += (double)__fmul_rn(dx, dy); ryy += (double)__fmul_rn(dy, dy);
            }
            sxx += rxx; sxy += rxy; syy += ryy;
        }
        const float A = (float)sxx, B = (float)sxy, C = (float)syy;
        float r;
        if (a.harris) {
            const float acbb = __fsub_rn(__fmul_rn(A, C), __fmul_rn(B, B));
            const float ac = __fadd_rn(A, C);
            r = __fsub_rn(acbb, __fmul_rn(__fmul_rn(a.kf, ac), ac));
        } else {
            const float ah = __fmul_rn(A, 0.5f), ch = __fmul_rn(C, 0.5f);
            const float t = __fsub_rn(ah, ch);
            const float u = __fadd_rn(__fmul_rn(B, B), __fmul_rn(t, t));
            r = __fsub_rn(__fadd_rn(ah, ch), sqrtf(u));
        }
        reinterpret_cast<float*>(dst + (size_t)y * dstep)[x] = r;
    }
}


// ---------------------------------------------------------------------------------- cornerHarris / MinEigenVal, rolling path
// CV_8UC1, ksize 3, blockSize 2 (BASELINE config 4): the roll.h skeleton.  A lane owns 16 output pixels of a row and walks a
// segment of rows; per source row it forms the two row passes (Rd = x-derivative taps, exact; Rs = scaled smoothing taps),
// keeps the last two of each, and from three of them the derivatives Dx, Dy of the middle row; the 2x2 box of the products
// is the sum over (previous, current) rows and (x-1, x) columns.  The reference accumulates those four floats in double
// (RowSum<float,double>/ColumnSum<double,float>).  cornerMinEigenVal does the same here (its response cancels to a small
// number on edges, so ulp-level differences in A, B, C would show at the 1e-5 level of the norm): bit-identical to the
// LDS-tiled kernel.  cornerHarris is well conditioned and adds them in float, (p + c)[x-1] + (p + c)[x]: at most 2 ulp away
// in A, B, C, < 1e-5 in the response norm against the 1e-4 bar; the f64 converts/adds were 16 of 40 VALU instructions per
// pixel and made the kernel VALU-bound at several times the HBM time.
// Values are held as pairs (column m, column m+8) so that the float work issues as v_pk_*_f32.
// Borders are two-level like the reference: the source is extrapolated for the Sobel passes, the *covariance image* for the
// box filter: the cov column left of x = 0 and the cov row above y = 0 are copies of an in-image cov column/row (or zero).
// The sign of Dy flips in upward-walking segments; only (sum dx*dy)^2 is used, so nothing needs correcting.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
struct CornerRollArgs { float kr0, kr1, kr2, kc1, kc2, kf; };

template <bool HARRIS, int CB>
__global__ __launch_bounds__(256) void k_corner_roll(const uchar* __restrict__ src, size_t sstep, size_t sframe,
                                                     uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                     int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes, int border, int alt,
                                                     CornerRollArgs a)
{
    typedef roll::Ctx<2, 2, 1, CB> Cx;
    typedef typename Cx::RawT RawT;
    constexpr int NW = Cx::NW;
    constexpr int NP = CB / 2;                         // output pairs (i, i + NP) per lane
    constexpr int NC = NP + 1;                         // cov columns x0-1+m / x0-1+NP+m
    constexpr int NS = NP + 3;                         // source columns x0-2+m / x0-2+NP+m
    Cx cx;
    if (!cx.init(src, sstep, sframe, W, H, nchunks, nstrips, segRows, nseg, nframes, border, alt)) return;
    __shared__ __attribute__((aligned(16))) uchar tscratch[4 * Cx::template tldsBytesPerWave<4>()];      // roll.h: transposed stores
    cx.useLds(tscratch, Cx::template tldsBytesPerWave<4>());
    dst += (size_t)cx.frame * dframe;
    struct RowR { f32x2 d[NC], s[NC]; };               // the two row passes of one source row
    struct RowC { f32x2 a[NC], b[NC], c[NC]; };        // dx*dx, dx*dy, dy*dy of one derivative row
    const int qx = mi355_borderInterpolate(-1, W, border);
    const bool fixL = cx.hasFirst && cx.lane == 0;

    auto rowPasses = [&](RowR& r, RawT raw, int valid) {
        uint32_t X[NW];
        if (!valid) {                                  // BORDER_CONSTANT row (window() only moves bytes)
#pragma unroll
            for (int d = 0; d < Cx::MD; d++) raw.m[d] = 0;
            raw.side[0] = 0;
        }
        cx.window(X, raw);
        f32x2 P[NS];
#pragma unroll
        for (int m = 0; m < NS; m++) {
            const int b0 = 2 + m, b1 = b0 + NP;         // window bytes: X[0] holds columns x0-4 .. x0-1
            P[m].x = (float)((X[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu);
            P[m].y = (float)((X[b1 >> 2] >> (8 * (b1 & 3))) & 0xffu);
        }
#pragma unroll
        for (int m = 0; m < NC; m++) {
            r.d[m] = P[m + 2] - P[m];
            f32x2 t = P[m] * f32x2{a.kr0, a.kr0};
            t = __builtin_elementwise_fma(f32x2{a.kr1, a.kr1}, P[m + 1], t);
            r.s[m] = __builtin_elementwise_fma(f32x2{a.kr2, a.kr2}, P[m + 2], t);
        }
    };
    auto products = [&](RowC& o, const RowR& r0, const RowR& r1, const RowR& r2) {
        f32x2 dx[NC], dy[NC];
#pragma unroll
        for (int m = 0; m < NC; m++) {
            const f32x2 c = r1.d[m] * f32x2{a.kc1, a.kc1};
            dx[m] = __builtin_elementwise_fma(f32x2{a.kc2, a.kc2}, r2.d[m] + r0.d[m], c);
            dy[m] = r2.s[m] - r0.s[m];
        }
        const float sx = qx < 0 ? 0.f : (qx == 0 ? dx[1].x : dx[2].x);
        const float sy = qx < 0 ? 0.f : (qx == 0 ? dy[1].x : dy[2].x);
        dx[0].x = fixL ? sx : dx[0].x;
        dy[0].x = fixL ? sy : dy[0].x;
#pragma unroll
        for (int m = 0; m < NC; m++) { o.a[m] = dx[m] * dx[m]; o.b[m] = dx[m] * dy[m]; o.c[m] = dy[m] * dy[m]; }
    };
    // response row y from the product rows p (kept) and c (new); c replaces p
    auto emit = [&](RowC& p, const RowC& c, int y) {
        typedef typename std::conditional<HARRIS, f32x2, f64x2>::type acc2;      // see the header comment
        acc2 vxx[NC], vxy[NC], vyy[NC];
#pragma unroll
        for (int m = 0; m < NC; m++) {
            vxx[m] = __builtin_convertvector(p.a[m], acc2) + __builtin_convertvector(c.a[m], acc2);
            vxy[m] = __builtin_convertvector(p.b[m], acc2) + __builtin_convertvector(c.b[m], acc2);
            vyy[m] = __builtin_convertvector(p.c[m], acc2) + __builtin_convertvector(c.c[m], acc2);
            p.a[m] = c.a[m]; p.b[m] = c.b[m]; p.c[m] = c.c[m];
        }
        f32x2 o[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const f32x2 A = __builtin_convertvector(vxx[i] + vxx[i + 1], f32x2);
            const f32x2 B = __builtin_convertvector(vxy[i] + vxy[i + 1], f32x2);
            const f32x2 C = __builtin_convertvector(vyy[i] + vyy[i + 1], f32x2);
            if (HARRIS) {
                const f32x2 acbb = A * C - B * B;
                const f32x2 ac = A + C;
                o[i] = acbb - (f32x2{a.kf, a.kf} * ac) * ac;
            } else {
                const f32x2 ah = A * f32x2{0.5f, 0.5f}, ch = C * f32x2{0.5f, 0.5f};
                const f32x2 t = ah - ch;
                const f32x2 u = B * B + t * t;
                o[i] = (ah + ch) - f32x2{sqrtf(u.x), sqrtf(u.y)};
            }
        }
        uint32_t ow[CB];                                   // memory order: pixels 0..NP-1 are the .x halves, NP..2NP-1 the .y halves
#pragma unroll
        for (int i = 0; i < NP; i++) { ow[i] = __float_as_uint(o[i].x); ow[NP + i] = __float_as_uint(o[i].y); }
        cx.template store<4>(dst, dstep, y, ow);
    };

    // cov rows in walking order: c_0 (prologue), c_1 .. c_n; step t emits image row gy(t-1) from (c_{t-1}, c_t) and needs the
    // source rows with logical indices t-2+o .. t+o, o = up
    const int o = cx.up;
    RowR R[3]; RowC Cp;
    {
        RawT r0, r1, r2; int v0, v1, v2;
        const bool virt = !cx.up && cx.y0 == 0;        // c_0 is the cov row above the image
        const int qy = mi355_borderInterpolate(-1, H, border);
        if (virt) { const int q = max(qy, 0); cx.issueImg(r0, q - 1, v0); cx.issueImg(r1, q, v1); cx.issueImg(r2, q + 1, v2); }
        else { cx.issueImg(r0, cx.gy(o - 2), v0); cx.issueImg(r1, cx.gy(o - 1), v1); cx.issueImg(r2, cx.gy(o), v2); }
        rowPasses(R[0], r0, v0); rowPasses(R[1], r1, v1); rowPasses(R[2], r2, v2);
        products(Cp, R[0], R[1], R[2]);
        if (virt) {
            if (qy < 0) {
#pragma unroll
                for (int m = 0; m < NC; m++) { Cp.a[m] = f32x2{0.f, 0.f}; Cp.b[m] = f32x2{0.f, 0.f}; Cp.c[m] = f32x2{0.f, 0.f}; }
            }
            cx.issueImg(r1, -1, v1); cx.issueImg(r2, 0, v2);
            rowPasses(R[1], r1, v1); rowPasses(R[2], r2, v2);
        }
    }
    // now R[1], R[2] hold logical rows o-1, o;  ring of loads for logical rows o+1, o+2, o+3
    RawT raw[3]; int rv[3];
#pragma unroll
    for (int u = 0; u < 3; u++) cx.issue(raw[u], o + 1 + u, rv[u]);
    for (int t = 1; t <= cx.nrows; t += 3) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            if (t + u <= cx.nrows) {
                // step t+u: the new row goes to slot u (the one holding the oldest row); older rows in slots u+1, u+2 (mod 3)
                rowPasses(R[u], raw[u], rv[u]);
                cx.issue(raw[u], t + u + o + 3, rv[u]);
                RowC Cn;
                products(Cn, R[(u + 1) % 3], R[(u + 2) % 3], R[u]);
                emit(Cp, Cn, cx.gy(t + u - 1));
            }
        }
    }
}

// ---------------------------------------------------------------------------------- cornerHarris, packed rolling path (the default for CV_8U, ksize 3, blockSize 2)
// The same walk as k_corner_roll, with the Sobel passes held as PACKED HALVES and the products formed by v_dot2_f32_f16 (round 5; k_corner_roll<true, 8> held its rows
// as float pairs: 169 VGPRs, 2 waves per SIMD, 22 VALU per pixel -- issue-bound at 0.52 of the roofline, profiles/r04_why_slow_call1.txt).  Every intermediate is a small
// integer, so half precision is EXACT here (integers up to 2048 are halves):
//   * a source row becomes pairs of adjacent columns as halves with one v_perm per pair (bytes 0x64 : p = the half 1024 + p) and one packed subtract of 1024;
//   * its row passes d = p[x+1] - p[x-1] (|d| <= 255) and s = p[x-1] + 2 p[x] + p[x+1] (<= 1020): 3 packed instructions per 2 columns;
//   * the column passes dx = d0 + 2 d1 + d2, dy = s2 - s0 (|.| <= 1020): 3 more, still exact, where the reference rounds scaled floats at every tap;
//   * the horizontal half of the 2x2 box is one v_dot2_f32_f16 on a (column x-1, column x) pair: dx.dx, dx.dy, dy.dy <= 2 * 1020^2 < 2^24, exact in float whatever the
//     order of the hardware's internal roundings; the vertical half is a float add of two such integers (<= 2^23) -- so the three box sums A', B', C' are EXACT integers, and
//     the only roundings are the six float operations of the response  R = s^4 (A'C' - B'^2) - k s^4 (A' + C')^2,  s = 1 / (2^(ksize-1) * blockSize * 255)  (corner.cpp:247-252:
//     the reference scales the Sobel taps by s and rounds at every tap, product and -- in double -- box step; its A = s^2 A' (1 + O(1e-7))).
//   Against the reference's own result the response differs by the reference's rounding noise: < 1e-6 of the global norm (bar: 1e-4, tests/test_corner_gpu.py).
// Upward-walking segments see Dy negated: only B'^2 is used.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
struct HarrisPArgs { float s4, ks4; };

// FAST: the launch serves the waves whose row pieces take the transposed store (roll.h), the !FAST launch the others (a ragged last strip, a window edge): one kernel with
// both stores in it needed 194 VGPRs -- the element-wise store of partial chunks is the register hog --, this one 120.
template <int RD, bool FAST>                             // RD: source rows in flight, a multiple of the three-slot row ring
__global__ __launch_bounds__(256) void k_harris_roll_p(const uchar* __restrict__ src, size_t sstep, size_t sframe,
                                                       uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                       int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes, int border, int alt,
                                                       HarrisPArgs a)
{
    typedef roll::Ctx<2, 2, 1, 8> Cx;
    typedef typename Cx::RawT RawT;
    constexpr int NW = Cx::NW;                           // 4 window dwords: columns x0-4 .. x0+11
    constexpr int NJ = 5;                                // packed pairs of columns (2j-1, 2j), j = 0 .. 4: columns x0-1 .. x0+8
    Cx cx;
    if (!cx.init(src, sstep, sframe, W, H, nchunks, nstrips, segRows, nseg, nframes, border, alt)) return;
    __shared__ __attribute__((aligned(16))) uchar tscratch[4 * Cx::template tldsBytesPerWave<4>()];      // roll.h: the wave's 2 KiB row piece is transposed through LDS into 1 KiB stores
    cx.useLds(tscratch, Cx::template tldsBytesPerWave<4>());
    if (cx.transposes() != FAST) return;
    dst += (size_t)cx.frame * dframe;
    struct RowR { h16x2 d[NJ], s[NJ]; };                 // the two row passes of one source row, packed
    struct RowH { f32x2 a[4], b[4], c[4]; };             // (column x-1) + (column x) of dx*dx, dx*dy, dy*dy of one derivative row, pixels (2i, 2i+1)
    const int qx = mi355_borderInterpolate(-1, W, border);
    const bool fixL = cx.hasFirst && cx.lane == 0;
    const h16x2 k1024 = {(_Float16)1024.f, (_Float16)1024.f}, k2 = {(_Float16)2.f, (_Float16)2.f};

    auto rowPasses = [&](RowR& r, RawT raw, int valid) {
        uint32_t X[NW];
        if (!valid) {                                    // BORDER_CONSTANT row (window() only moves bytes)
#pragma unroll
            for (int d = 0; d < Cx::MD; d++) raw.m[d] = 0;
            raw.side[0] = 0;
        }
        cx.window(X, raw);
        h16x2 P[NJ + 1];                                 // columns (2j-2, 2j-1) = window bytes (2j+2, 2j+3), as halves
#pragma unroll
        for (int j = 0; j <= NJ; j++)
            P[j] = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x64646464u, X[(2 * j + 2) >> 2], ((2 * j + 2) & 3) ? 0x04030402u : 0x04010400u)) - k1024;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const h16x2 C = __builtin_bit_cast(h16x2, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, P[j + 1]), __builtin_bit_cast(uint32_t, P[j]), 16));   // columns (2j-1, 2j)
            r.d[j] = P[j + 1] - P[j];
            r.s[j] = __builtin_elementwise_fma(C, k2, P[j] + P[j + 1]);
        }
    };
    auto products = [&](RowH& o, const RowR& r0, const RowR& r1, const RowR& r2) {
        uint32_t dx[NJ], dy[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            dx[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(r1.d[j], k2, r0.d[j] + r2.d[j]));
            dy[j] = __builtin_bit_cast(uint32_t, r2.s[j] - r0.s[j]);
        }
        // the covariance column left of x = 0 is a copy of covariance column borderInterpolate(-1) (or zero)
        const uint32_t sx = qx < 0 ? 0u : (qx == 0 ? dx[0] >> 16 : dx[1] & 0xffffu);
        const uint32_t sy = qx < 0 ? 0u : (qx == 0 ? dy[0] >> 16 : dy[1] & 0xffffu);
        dx[0] = fixL ? ((dx[0] & 0xffff0000u) | sx) : dx[0];
        dy[0] = fixL ? ((dy[0] & 0xffff0000u) | sy) : dy[0];
        // pixel 2i: columns (2i-1, 2i) = pair i; pixel 2i+1: columns (2i, 2i+1) = the high half of pair i and the low half of pair i+1
        uint32_t xo[4], yo[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { xo[i] = __builtin_amdgcn_alignbit(dx[i + 1], dx[i], 16); yo[i] = __builtin_amdgcn_alignbit(dy[i + 1], dy[i], 16); }
        // 24 v_dot2_f32_f16 with an inline-zero accumulator in ONE asm statement.  The builtin selects v_dot2c_f32_f16, whose accumulator is its destination: a v_mov 0 per
        // product, 24 per row.  Inline asm is opaque to the compiler's hazard recognizer, and a DOT result must not be read by another VALU instruction within 3 wait states
        // (one asm per product gave one wrong pixel per turn of the unrolled loop on the MI355X): inside the block no product reads another's result, and the trailing
        // s_nop 2 covers the last ones; the early-clobber outputs keep every result off the inputs still to be read.
        float h[24];
        asm("v_dot2_f32_f16 %0, %24, %24, 0\n\t"
                "v_dot2_f32_f16 %1, %28, %28, 0\n\t"
                "v_dot2_f32_f16 %2, %24, %32, 0\n\t"
                "v_dot2_f32_f16 %3, %28, %36, 0\n\t"
                "v_dot2_f32_f16 %4, %32, %32, 0\n\t"
                "v_dot2_f32_f16 %5, %36, %36, 0\n\t"
                "v_dot2_f32_f16 %6, %25, %25, 0\n\t"
                "v_dot2_f32_f16 %7, %29, %29, 0\n\t"
                "v_dot2_f32_f16 %8, %25, %33, 0\n\t"
                "v_dot2_f32_f16 %9, %29, %37, 0\n\t"
                "v_dot2_f32_f16 %10, %33, %33, 0\n\t"
                "v_dot2_f32_f16 %11, %37, %37, 0\n\t"
                "v_dot2_f32_f16 %12, %26, %26, 0\n\t"
                "v_dot2_f32_f16 %13, %30, %30, 0\n\t"
                "v_dot2_f32_f16 %14, %26, %34, 0\n\t"
                "v_dot2_f32_f16 %15, %30, %38, 0\n\t"
                "v_dot2_f32_f16 %16, %34, %34, 0\n\t"
                "v_dot2_f32_f16 %17, %38, %38, 0\n\t"
                "v_dot2_f32_f16 %18, %27, %27, 0\n\t"
                "v_dot2_f32_f16 %19, %31, %31, 0\n\t"
                "v_dot2_f32_f16 %20, %27, %35, 0\n\t"
                "v_dot2_f32_f16 %21, %31, %39, 0\n\t"
                "v_dot2_f32_f16 %22, %35, %35, 0\n\t"
                "v_dot2_f32_f16 %23, %39, %39, 0\n\t"
            "s_nop 2"
            : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(h[4]), "=&v"(h[5]), "=&v"(h[6]), "=&v"(h[7]), "=&v"(h[8]), "=&v"(h[9]), "=&v"(h[10]), "=&v"(h[11]),
              "=&v"(h[12]), "=&v"(h[13]), "=&v"(h[14]), "=&v"(h[15]), "=&v"(h[16]), "=&v"(h[17]), "=&v"(h[18]), "=&v"(h[19]), "=&v"(h[20]), "=&v"(h[21]), "=&v"(h[22]), "=&v"(h[23])
            : "v"(dx[0]), "v"(dx[1]), "v"(dx[2]), "v"(dx[3]), "v"(xo[0]), "v"(xo[1]), "v"(xo[2]), "v"(xo[3]),
              "v"(dy[0]), "v"(dy[1]), "v"(dy[2]), "v"(dy[3]), "v"(yo[0]), "v"(yo[1]), "v"(yo[2]), "v"(yo[3]));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            o.a[i] = f32x2{h[6 * i], h[6 * i + 1]};
            o.b[i] = f32x2{h[6 * i + 2], h[6 * i + 3]};
            o.c[i] = f32x2{h[6 * i + 4], h[6 * i + 5]};
        }
    };
    // response row y from the box rows p (kept) and c (new); c replaces p
    auto emit = [&](auto fastTag, RowH& p, const RowH& c, int y) {
        uint32_t ow[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const f32x2 A = p.a[i] + c.a[i], B = p.b[i] + c.b[i], C = p.c[i] + c.c[i];
            p.a[i] = c.a[i]; p.b[i] = c.b[i]; p.c[i] = c.c[i];
            const f32x2 det = __builtin_elementwise_fma(-B, B, A * C);
            const f32x2 ac = A + C;
            const f32x2 r = __builtin_elementwise_fma(f32x2{-a.ks4, -a.ks4}, ac * ac, det * f32x2{a.s4, a.s4});
            ow[2 * i] = __float_as_uint(r.x); ow[2 * i + 1] = __float_as_uint(r.y);
        }
        if constexpr (decltype(fastTag)::value) cx.template storeT<4>(dst, dstep, y, ow);
        else cx.template storeLanes<4>(dst, dstep, y, ow);
    };

    // rows in walking order exactly as in k_corner_roll: c_0 (prologue), c_1 .. c_n; step t emits image row gy(t-1) from (c_{t-1}, c_t)
    const int o = cx.up;
    RowR R[3]; RowH Hp;
    {
        RawT r0, r1, r2; int v0, v1, v2;
        const bool virt = !cx.up && cx.y0 == 0;          // c_0 is the covariance row above the image
        const int qy = mi355_borderInterpolate(-1, H, border);
        if (virt) { const int q = max(qy, 0); cx.issueImg(r0, q - 1, v0); cx.issueImg(r1, q, v1); cx.issueImg(r2, q + 1, v2); }
        else { cx.issueImg(r0, cx.gy(o - 2), v0); cx.issueImg(r1, cx.gy(o - 1), v1); cx.issueImg(r2, cx.gy(o), v2); }
        rowPasses(R[0], r0, v0); rowPasses(R[1], r1, v1); rowPasses(R[2], r2, v2);
        products(Hp, R[0], R[1], R[2]);
        if (virt) {
            if (qy < 0) {
#pragma unroll
                for (int i = 0; i < 4; i++) { Hp.a[i] = f32x2{0.f, 0.f}; Hp.b[i] = f32x2{0.f, 0.f}; Hp.c[i] = f32x2{0.f, 0.f}; }
            }
            cx.issueImg(r1, -1, v1); cx.issueImg(r2, 0, v2);
            rowPasses(R[1], r1, v1); rowPasses(R[2], r2, v2);
        }
    }
    RawT raw[RD]; int rv[RD];
#pragma unroll
    for (int u = 0; u < RD; u++) cx.issue(raw[u], o + 1 + u, rv[u]);
    // The walk: whole turns of RD rows run without a condition inside (the compiler then counts the loads and stores in flight instead of draining them at every row);
    // the last, partial turn tests each row.  Which store a wave uses is decided once, here.
    auto walk = [&](auto fastTag) {
        auto step = [&](int t, int u) {
            rowPasses(R[u % 3], raw[u], rv[u]);
            cx.issue(raw[u], t + u + o + RD, rv[u]);
            RowH Hn;
            products(Hn, R[(u + 1) % 3], R[(u + 2) % 3], R[u % 3]);
            emit(fastTag, Hp, Hn, cx.gy(t + u - 1));
            __builtin_amdgcn_sched_barrier(0);               // rows are not interleaved by the scheduler (it otherwise keeps several rows' intermediates live: 190 VGPRs)
        };
        int t = 1;
        for (; t + RD - 1 <= cx.nrows; t += RD) {
#pragma unroll
            for (int u = 0; u < RD; u++) step(t, u);
        }
#pragma unroll
        for (int u = 0; u < RD - 1; u++)
            if (t + u <= cx.nrows) step(t, u);
    };
    walk(std::integral_constant<bool, FAST>{});
}

bool derivTaps(int order, int ksize, bool scharr, std::vector<int>& k)
{
    if (scharr) { if (order == 0) k = {3, 10, 3}; else k = {-1, 0, 1}; return true; }
    if (ksize == 1 && order > 0) ksize = 3;
    if (ksize % 2 == 0 || ksize > 7 || ksize <= order) return false;
    if (ksize == 1) { k = {1}; return true; }
    if (ksize == 3) { if (order == 0) k = {1, 2, 1}; else k = {-1, 0, 1}; return true; }
    std::vector<int> kerI(ksize + 1, 0);
    kerI[0] = 1;
    for (int i = 0; i < ksize - order - 1; i++) { int ov = kerI[0]; for (int j = 1; j <= ksize; j++) { int nv = kerI[j] + kerI[j - 1]; kerI[j - 1] = ov; ov = nv; } }
    for (int i = 0; i < order; i++) { int ov = -kerI[0]; for (int j = 1; j <= ksize; j++) { int nv = kerI[j - 1] - kerI[j]; kerI[j - 1] = ov; ov = nv; } }
    k.assign(kerI.begin(), kerI.begin() + ksize);
    return true;
}

int launchCorner(const uchar* ds, size_t dss, size_t sframe, uchar* dd, size_t dds, size_t dframe, int nframes,
                 int W, int H, int sdepth, int blockSize, int ksize, double k, int border, bool harris, hipStream_t st)
{
    CornerArgs a; memset(&a, 0, sizeof a);
    a.W = W; a.H = H; a.sdepth = sdepth; a.bs = blockSize; a.ax = blockSize / 2; a.ay = blockSize / 2; a.border = border;
    a.harris = harris; a.kf = (float)k;
    double scale = (double)(1 << ((ksize > 0 ? ksize : 3) - 1)) * blockSize;       // corner.cpp:247-252
    if (ksize < 0) scale *= 2.0;
    if (sdepth == D8U) scale *= 255.0;
    scale = 1.0 / scale;
    const bool scharr = ksize <= 0;
    std::vector<int> d1, s0x, s0y;
    // Dx = Sobel(1,0): kx = derivative taps, ky = smoothing taps * scale; Dy = Sobel(0,1): kx = smoothing * scale, ky = derivative
    std::vector<int> dxr, dxc, dyr, dyc;
    if (!derivTaps(1, ksize, scharr, dxr) || !derivTaps(0, ksize, scharr, dxc) || !derivTaps(0, ksize, scharr, dyr) || !derivTaps(1, ksize, scharr, dyc))
        return mi355::declined(__func__, __LINE__, "!derivTaps(1, ksize, scharr, dxr) || !derivTaps(0, ksize, scharr, dxc) || !derivTaps(0, ksize, scharr, dyr) || !derivTaps(1, ksize, scharr, dyc)");
    a.dxNRow = (int)dxr.size(); a.dxNCol = (int)dxc.size(); a.dyNRow = (int)dyr.size(); a.dyNCol = (int)dyc.size();
    for (int i = 0; i < a.dxNRow; i++) a.dxRow[i] = (float)dxr[i];
    for (int i = 0; i < a.dxNCol; i++) a.dxCol[i] = (float)((double)dxc[i] * scale);     // `ky *= scale` (dx != 0)
    for (int i = 0; i < a.dyNRow; i++) a.dyRow[i] = (float)((double)dyr[i] * scale);     // `kx *= scale` (dx == 0)
    for (int i = 0; i < a.dyNCol; i++) a.dyCol[i] = (float)dyc[i];
    if (scale == 1) { /* unreachable for the depths handled; kept for symmetry with cv::Sobel */ }
    a.rx = std::max(a.dxNRow, a.dyNRow) / 2; a.ry = std::max(a.dxNCol, a.dyNCol) / 2;
    if (sdepth == D8U && ksize == 3 && blockSize == 2 && H >= 2 && std::getenv("MI355CV_CORNER_LDS") == nullptr &&
        (((uintptr_t)dd | dds | dframe) & 3) == 0 && roll::eligible(ds, dss, sframe, ds, dss, sframe, W, 1, 2, border, 8)) {
        if (harris && std::getenv("MI355CV_CORNER_FLOATROLL") == nullptr) {          // the packed-half rolling kernel; the float one stays for cornerMinEigenVal and as the A/B partner
            const double s2 = scale * scale;
            HarrisPArgs ia = {(float)(s2 * s2), (float)((double)a.kf * s2 * s2)};
            const char* segEnv = std::getenv("MI355CV_CORNER_SEG");               // tuning experiments
            const roll::Geom g = roll::geometry(W, H, 1, nframes, segEnv ? atoi(segEnv) : 36, 4, 8);
#define HROLL(RD_, FAST_) hipLaunchKernelGGL((k_harris_roll_p<RD_, FAST_>), dim3(g.blocks), dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W, H, g.nchunks, g.nstrips, g.seg, g.nseg, nframes, border, 1, ia)
            HROLL(3, true);                                                       // (6 rows in flight measured the same: profiles/r05_harris.txt)
            if (W % 8 != 0) HROLL(3, false);                                      // the ragged last strip of every row
#undef HROLL
            noteKernel("k_harris_roll_p grid=%u x256 seg=%d rows", g.blocks, g.seg);
            return MI355CV_OK;
        }
        CornerRollArgs ra = {a.dyRow[0], a.dyRow[1], a.dyRow[2], a.dxCol[1], a.dxCol[2], a.kf};
        const char* segEnv = std::getenv("MI355CV_CORNER_SEG");               // tuning experiments
        const roll::Geom g = roll::geometry(W, H, 1, nframes, segEnv ? atoi(segEnv) : 36, 4, 8);
#define CROLL(HR, CB_) hipLaunchKernelGGL((k_corner_roll<HR, CB_>), dim3(g.blocks), dim3(256), 0, st, ds, dss, sframe, dd, dds, dframe, W, H, g.nchunks, g.nstrips, g.seg, g.nseg, nframes, border, 1, ra)
        if (harris) CROLL(true, 8); else CROLL(false, 8);                 // (16 pixels per lane: 256 VGPRs, one wave per SIMD -- removed in round 5)
#undef CROLL
        return MI355CV_OK;
    }
    const int PW = CT_X + a.bs - 1, PH = CT_Y + a.bs - 1, SW = PW + 2 * a.rx, SH = PH + 2 * a.ry;
    const size_t lds = (size_t)(SW * SH + 2 * PW * SH + 2 * PW * PH) * sizeof(float);
    dim3 grid(divUp(W, CT_X), divUp(H, CT_Y), nframes);
    hipLaunchKernelGGL(k_corner_fused, grid, dim3(256), lds, st, ds, dss, sframe, dd, dds, dframe, a);
    return MI355CV_OK;
}

int runCorner(const char* entry, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
              int W, int H, int src_type, int blockSize, int ksize, double k, int borderType, bool harris)
{
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    const int sdepth = MI355CV_MAT_DEPTH(src_type);
    if (MI355CV_MAT_CN(src_type) != 1 || (sdepth != D8U && sdepth != D32F)) return mi355::declined(__func__, __LINE__, "MI355CV_MAT_CN(src_type) != 1 || (sdepth != D8U && sdepth != D32F)");   // corner.cpp:254
    const int border = borderType & ~MI355CV_BORDER_ISOLATED;
    if (border == B_WRAP || border < 0 || border > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "border == B_WRAP || border < 0 || border > B_REFLECT_101");          // FilterEngine rejects WRAP
    if (blockSize < 1 || blockSize > 16 || W <= 0 || H <= 0 || nframes <= 0) return mi355::declined(__func__, __LINE__, "blockSize < 1 || blockSize > 16 || W <= 0 || H <= 0 || nframes <= 0");
    if (!(ksize == -1 || ksize == 1 || ksize == 3 || ksize == 5 || ksize == 7)) return mi355::declined(__func__, __LINE__, "!(ksize == -1 || ksize == 1 || ksize == 3 || ksize == 5 || ksize == 7)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src, (size_t)W * H, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src, (size_t)W * H, minPixels(HOST_HEAVY))");
    size_t dss = sstep, dds = dstep;
    const uchar* ds = src; uchar* dd = dst;
    if (nframes == 1) {
        ds = stg.in(src, sstep, (size_t)W * (sdepth == D8U ? 1 : 4), H, &dss);
        dd = stg.out(dst, dstep, (size_t)W * 4, H, &dds);
        if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    } else if (!isDevicePtr(src) || !isDevicePtr(dst)) return mi355::declined(__func__, __LINE__, "!isDevicePtr(src) || !isDevicePtr(dst)");
    int rc = launchCorner(ds, dss, sframe, dd, dds, dframe, nframes, W, H, sdepth, blockSize, ksize, k, border, harris, stream());
    if (rc != MI355CV_OK) return rc;
    return stg.finish(entry);
}

// ---------------------------------------------------------------------------------- goodFeaturesToTrack
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ inline float ord2f(unsigned o) { unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; memcpy(&f, &u, 4); return f; }

__global__ __launch_bounds__(256) void k_maxval(const float* __restrict__ eig, size_t estep, const uchar* __restrict__ mask, size_t mstep,
                                                int W, int H, unsigned* __restrict__ out)
{
    unsigned best = 0;                                  // below every real float in the ordered encoding
    for (int y = blockIdx.x; y < H; y += gridDim.x) {
        const float* row = reinterpret_cast<const float*>(reinterpret_cast<const uchar*>(eig) + (size_t)y * estep);
        for (int x = threadIdx.x; x < W; x += 256)
            if (!mask || mask[(size_t)y * mstep + x]) best = max(best, f2ord(row[x]));
    }
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));      // wave64 reduction
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}

// a candidate as one sortable word: (order-preserving image of the response) << 32 | pixel index
__device__ __forceinline__ unsigned ordF(float v) { const unsigned b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
inline float unordF(unsigned o) { const unsigned b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float v; memcpy(&v, &b, 4); return v; }

__global__ __launch_bounds__(256) void k_gftt_candidates(const float* __restrict__ eig, size_t estep, const uchar* __restrict__ mask, size_t mstep,
                                                         int W, int H, const unsigned* __restrict__ maxOrd, double quality,
                                                         unsigned long long* __restrict__ out, unsigned* __restrict__ count, unsigned capacity)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
    bool hit = x < W - 1 && y < H - 1;
    float v = 0.f;
    if (hit) {
        unsigned mo = *maxOrd;
        unsigned mu = (mo & 0x80000000u) ? (mo & 0x7fffffffu) : ~mo;
        const double maxVal = mo == 0 ? 0.0 : (double)__uint_as_float(mu);
        const float thr = (float)(maxVal * quality);          // cv::threshold converts the double threshold to the image depth
        auto T = [&](int yy, int xx) { float t = reinterpret_cast<const float*>(reinterpret_cast<const uchar*>(eig) + (size_t)yy * estep)[xx]; return t > thr ? t : 0.f; };
        v = T(y, x);
        hit = !(v == 0.f || (mask && !mask[(size_t)y * mstep + x]));
        if (hit) {
            float m = v;
#pragma unroll
            for (int j = -1; j <= 1; j++)
#pragma unroll
                for (int i = -1; i <= 1; i++) m = fmaxf(m, T(y + j, x + i));
            hit = v == m;
        }
    }
    // one atomic per wavefront (a counter every candidate adds to by itself serialises on a textured frame: 1.7 ms per 4K ORB call in FAST's former collect kernel, profiles/r03_orb_trace.txt)
    const unsigned long long wm = __ballot(hit);
    if (!wm) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)wm) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(wm));
    base = __shfl(base, leader, 64);
    if (!hit) return;
    const unsigned slot = base + (unsigned)__popcll(wm & ((1ull << lane) - 1ull));
    if (slot < capacity) out[slot] = ((unsigned long long)ordF(v) << 32) | (unsigned)(y * W + x);
}

} // namespace

extern "C" {

MI355CV_API int mi355cv_pyrdown(const uchar* src_data, size_t src_step, int src_width, int src_height, uchar* dst_data, size_t dst_step,
                                int dst_width, int dst_height, int depth, int cn, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    return runPyrDown("pyrdown", src_data, src_step, 0, src_width, src_height, dst_data, dst_step, 0, dst_width, dst_height, 1, depth, cn, 0, 0, 0, 0, border_type);
}

MI355CV_API int mi355cv_pyrdown_offset(const uchar* src_data, size_t src_step, int src_width, int src_height, uchar* dst_data, size_t dst_step,
                                       int dst_width, int dst_height, int depth, int cn, int margin_left, int margin_top, int margin_right,
                                       int margin_bottom, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    return runPyrDown("pyrdown_offset", src_data, src_step, 0, src_width, src_height, dst_data, dst_step, 0, dst_width, dst_height, 1, depth, cn,
                      margin_left, margin_top, margin_right, margin_bottom, border_type);
}

MI355CV_API int mi355cv_pyrdownBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride, int src_width, int src_height,
                                     uchar* dst_data, size_t dst_step, size_t dst_frame_stride, int dst_width, int dst_height, int nframes,
                                     int depth, int cn, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (src_width > 0 && src_height > 0 && dst_width > 0 && dst_height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {        // frames in host memory
        const size_t pix = (size_t)cn * depthBytes(depth);
        const HostBatch hb = {src_data, src_step, src_frame_stride, pix * src_width, src_height, dst_data, dst_step, dst_frame_stride, pix * dst_width, dst_height, nframes};
        return runHostBatch("pyrdownBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_pyrdownBatch(s, ss, sf, src_width, src_height, d, ds, df, dst_width, dst_height, nf, depth, cn, border_type); });
    }
    return runPyrDown("pyrdownBatch", src_data, src_step, nframes == 1 ? 0 : src_frame_stride, src_width, src_height, dst_data, dst_step,
                      nframes == 1 ? 0 : dst_frame_stride, dst_width, dst_height, nframes, depth, cn, 0, 0, 0, 0, border_type);
}

// cv::buildPyramid (pyramids.cpp:1616-1643): level 0 is the source itself; levels 1..maxlevel by pyrDown of the previous one.
// dst_data[i], dst_step[i] describe level i+1 ((w+1)/2 x (h+1)/2 of the previous level), pre-allocated by the caller.
MI355CV_API int mi355cv_buildPyramid(const uchar* src_data, size_t src_step, int width, int height, int depth, int cn,
                                     uchar** dst_data, const size_t* dst_step, int maxlevel, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (!dst_data || !dst_step || maxlevel < 0) return mi355::declined(__func__, __LINE__, "!dst_data || !dst_step || maxlevel < 0");
    const uchar* s = src_data; size_t ss = src_step; int w = width, h = height;
    for (int l = 0; l < maxlevel; l++) {
        const int dw = (w + 1) / 2, dh = (h + 1) / 2;
        int rc = runPyrDown("buildPyramid", s, ss, 0, w, h, dst_data[l], dst_step[l], 0, dw, dh, 1, depth, cn, 0, 0, 0, 0, border_type);
        if (rc != MI355CV_OK) return l == 0 ? rc : MI355CV_ERROR_UNKNOWN;
        s = dst_data[l]; ss = dst_step[l]; w = dw; h = dh;
    }
    return MI355CV_OK;
}

// cv::buildPyramid over a batch of device-resident frames, every level of every frame enqueued by ONE call (levels 1..maxlevel into the
// caller's pre-allocated arrays: dst_data[l-1] = level l of frame 0, frames dst_frame_stride[l-1] bytes apart).  Level l+1 only needs level l,
// which the previous launch just wrote and L2 / Infinity Cache still hold, so the chain reads level 0 from HBM once.
MI355CV_API int mi355cv_buildPyramidBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride, int width, int height, int depth, int cn,
                                          uchar* const* dst_data, const size_t* dst_step, const size_t* dst_frame_stride, int maxlevel, int nframes,
                                          int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || !src_data || !dst_data || !dst_step || !dst_frame_stride || maxlevel < 1 || maxlevel > 30 || nframes < 1) return mi355::declined(__func__, __LINE__, "disabled() || !src_data || !dst_data || !dst_step || !dst_frame_stride || maxlevel < 1 || maxlevel > 30 || nframes < 1");
    int border = border_type & ~MI355CV_BORDER_ISOLATED;
    if (border == B_CONSTANT || border < 0 || border > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "border == B_CONSTANT || border < 0 || border > B_REFLECT_101");
    if (!(depth == D8U || depth == D16U || depth == D16S || depth == D32F) || cn < 1 || cn > 4 || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "!(depth == D8U || depth == D16U || depth == D16S || depth == D32F) || cn < 1 || cn > 4 || width <= 0 || height <= 0");
    // every frame and every level in host memory: chunks of frames cross PCIe through two sets of device buffers, the upload of
    // chunk i + 1 under the kernels and the downloads of chunk i; the chunk itself is this entry on device pointers
    bool allHost = hostBatchEligible(src_data, dst_data[0], nframes) && maxlevel <= HOST_BATCH_MAX_OUT;
    for (int l = 1; l < maxlevel && allHost; l++) allHost = dst_data[l] && ptrKind(dst_data[l]) == PTR_HOST;
    if (allHost) {
        const size_t e = (size_t)depthBytes(depth) * cn;
        HostBatchN hb; memset(&hb, 0, sizeof hb);
        hb.src = src_data; hb.sstep = src_step; hb.sframe = src_frame_stride; hb.srowBytes = (size_t)width * e; hb.srows = height; hb.nout = maxlevel; hb.nframes = nframes;
        int w = width, h = height;
        for (int l = 0; l < maxlevel; l++) { w = (w + 1) / 2; h = (h + 1) / 2; hb.out[l] = {dst_data[l], dst_step[l], dst_frame_stride[l], (size_t)w * e, h}; }
        return runHostBatchN("buildPyramidBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* const* d, const size_t* ds, const size_t* df, int nf) {
            return mi355cv_buildPyramidBatch(s, ss, sf, width, height, depth, cn, d, ds, df, maxlevel, nf, border_type); });
    }
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (!isDevicePtr(src_data)) return setError(MI355CV_NOT_IMPLEMENTED, "buildPyramidBatch: frames and levels all in device memory, or all in host memory");
    for (int l = 0; l < maxlevel; l++) if (!isDevicePtr(dst_data[l])) return setError(MI355CV_NOT_IMPLEMENTED, "buildPyramidBatch: frames and levels all in device memory, or all in host memory");
    const uchar* s = src_data; size_t ss = src_step, sf = nframes == 1 ? 0 : src_frame_stride; int w = width, h = height;
    for (int l = 0; l < maxlevel; l++) {
        const int dw = (w + 1) / 2, dh = (h + 1) / 2;
        const size_t df = nframes == 1 ? 0 : dst_frame_stride[l];
        // the last three levels in one launch once the level they start from is small (its redundant tile reads come from cache)
        if (l == maxlevel - 3 && l >= 1 && launchPyr3(s, ss, sf, w, h, dst_data + l, dst_step + l, dst_frame_stride + l, nframes, depth, cn, border, stream())) break;
        launchPyrDown(s, ss, sf, w, h, dst_data[l], dst_step[l], df, dw, dh, nframes, depth, cn, 0, 0, 0, 0, border, stream());
        s = dst_data[l]; ss = dst_step[l]; sf = df; w = dw; h = dh;
    }
    return stg.finish("buildPyramidBatch");
}

MI355CV_API int mi355cv_cornerHarris(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                     int src_type, int blockSize, int ksize, double k, int borderType)
{
    mi355::EntryGuard entry_(__func__);
    return runCorner("cornerHarris", src_data, src_step, 0, dst_data, dst_step, 0, 1, width, height, src_type, blockSize, ksize, k, borderType, true);
}

MI355CV_API int mi355cv_cornerMinEigenVal(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                          int src_type, int blockSize, int ksize, int borderType)
{
    mi355::EntryGuard entry_(__func__);
    return runCorner("cornerMinEigenVal", src_data, src_step, 0, dst_data, dst_step, 0, 1, width, height, src_type, blockSize, ksize, 0.0, borderType, false);
}

MI355CV_API int mi355cv_cornerHarrisBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride, uchar* dst_data, size_t dst_step,
                                          size_t dst_frame_stride, int nframes, int width, int height, int src_type, int blockSize, int ksize,
                                          double k, int borderType)
{
    mi355::EntryGuard entry_(__func__);
    if (width > 0 && height > 0 && hostBatchEligible(src_data, dst_data, nframes)) {            // frames in host memory: chunks through two sets of device buffers
        const HostBatch hb = {src_data, src_step, src_frame_stride, (size_t)width * MI355CV_MAT_CN(src_type) * depthBytes(MI355CV_MAT_DEPTH(src_type)), height, dst_data, dst_step, dst_frame_stride, (size_t)width * 4, height, nframes};
        return runHostBatch("cornerHarrisBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return mi355cv_cornerHarrisBatch(s, ss, sf, d, ds, df, nf, width, height, src_type, blockSize, ksize, k, borderType); });
    }
    return runCorner("cornerHarrisBatch", src_data, src_step, nframes == 1 ? 0 : src_frame_stride, dst_data, dst_step,
                     nframes == 1 ? 0 : dst_frame_stride, nframes, width, height, src_type, blockSize, ksize, k, borderType, true);
}

// cv::goodFeaturesToTrack (featureselect.cpp:382-548).  corners: x0,y0,x1,y1,... (capacity maxCorners pairs, or width*height
// when maxCorners <= 0); quality (optional) receives the response of each returned corner.  Returns the corner count (>= 0),
// -1 when the arguments are not supported (nothing computed), -2 on a device failure.
MI355CV_API int mi355cv_goodFeaturesToTrack(const uchar* src_data, size_t src_step, int width, int height, int src_type,
                                            float* corners, float* quality, int maxCorners, double qualityLevel, double minDistance,
                                            const uchar* mask_data, size_t mask_step, int blockSize, int gradientSize,
                                            int useHarrisDetector, double harrisK)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || !corners || qualityLevel <= 0 || minDistance < 0 || width <= 0 || height <= 0) return -1;
    const int sdepth = MI355CV_MAT_DEPTH(src_type);
    if (MI355CV_MAT_CN(src_type) != 1 || (sdepth != D8U && sdepth != D32F)) return -1;
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return -1;
    size_t dss, dms = mask_step;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * (sdepth == D8U ? 1 : 4), height, &dss);
    const uchar* dm = nullptr;
    if (mask_data) dm = stg.in(mask_data, mask_step, (size_t)width, height, &dms);
    const size_t estep = ((size_t)width * 4 + 255) & ~size_t(255);
    uchar* eig = (uchar*)stg.scratch(estep * height);
    const unsigned capacity = (unsigned)std::min<size_t>((size_t)width * height, (size_t)1 << 26);
    unsigned long long* cand = (unsigned long long*)stg.scratch((size_t)capacity * 8);
    unsigned* ctr = (unsigned*)stg.scratch(256);
    if (!ds || !eig || !cand || !ctr || (mask_data && !dm)) return -1;
    hipStream_t st = stream();
    // cv::cornerHarris / cornerMinEigenVal with the default border (BORDER_DEFAULT, featureselect.cpp:408-411)
    int rc = launchCorner(ds, dss, 0, eig, estep, 0, 1, width, height, sdepth, blockSize, gradientSize, harrisK, B_REFLECT_101, useHarrisDetector != 0, st);
    if (rc != MI355CV_OK) return -1;
    if (hipMemsetAsync(ctr, 0, 8, st) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_maxval, dim3(std::min(height, 1024)), dim3(256), 0, st, (const float*)eig, estep, dm, dms, width, height, ctr);
    if (width > 2 && height > 2) {
        dim3 grid(divUp(width - 2, 64), divUp(height - 2, 4));
        hipLaunchKernelGGL(k_gftt_candidates, grid, dim3(256), 0, st, (const float*)eig, estep, dm, dms, width, height, ctr, qualityLevel, cand, ctr + 1, capacity);
    }
    unsigned hc[2] = {0, 0};
    if (hipMemcpyAsync(hc, ctr, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -2;
    const unsigned total = std::min(hc[1], capacity);
    // order the candidates on the device (response, then address, both descending: std::sort with greaterThanPtr, featureselect.cpp:55-60, :447)
    // and bring them over in chunks: the greedy minimum-distance pass below usually stops long before the end of the list
    const unsigned long long* sorted = cand;
    if (total > 1) {
        const size_t tb = sortKeysDescTemp(total);
        void* temp = stg.scratch(tb ? tb : 16);
        unsigned long long* outKeys = (unsigned long long*)stg.scratch((size_t)total * 8);
        if (!tb || !temp || !outKeys || !sortKeysDesc(temp, tb, cand, outKeys, total, st)) return -2;
        sorted = outKeys;
    }
    std::vector<unsigned long long> c;
    unsigned fetched = 0;
    auto need = [&](unsigned i) -> bool {                                // make candidate i available on the host
        if (i < fetched) return true;
        const unsigned chunk = std::min(total - fetched, std::max(65536u, fetched));        // doubling chunks
        c.resize((size_t)fetched + chunk);
        if (hipMemcpyAsync(c.data() + fetched, sorted + fetched, (size_t)chunk * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return false;
        fetched += chunk;
        return true;
    };
    int n = 0;
    bool failed = false;
    const int cap = maxCorners > 0 ? maxCorners : width * height;
    if (minDistance >= 1) {                                          // :451-525 grid-based rejection
        const int cell = (int)nearbyint(minDistance);
        const int gw = (width + cell - 1) / cell, gh = (height + cell - 1) / cell;
        std::vector<std::vector<int>> grid((size_t)gw * gh);
        const double md2 = minDistance * minDistance;
        for (unsigned i = 0; i < total; i++) {
            if (!need(i)) { failed = true; break; }
            const int idx = (int)(unsigned)(c[i] & 0xffffffffu);
            const int y = idx / width, x = idx % width;
            const int xc = x / cell, yc = y / cell;
            const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            bool good = true;
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (int j : grid[(size_t)yy * gw + xx]) {
                        const float dx = x - corners[2 * j], dy = y - corners[2 * j + 1];
                        if (dx * dx + dy * dy < md2) { good = false; break; }
                    }
            if (good) {
                if (n >= cap) break;
                corners[2 * n] = (float)x; corners[2 * n + 1] = (float)y;
                if (quality) quality[n] = unordF((unsigned)(c[i] >> 32));
                grid[(size_t)yc * gw + xc].push_back(n);
                n++;
                if (maxCorners > 0 && n == maxCorners) break;
            }
        }
    } else {
        for (unsigned i = 0; i < total && n < cap; i++) {
            if (!need(i)) { failed = true; break; }
            const int idx = (int)(unsigned)(c[i] & 0xffffffffu);
            corners[2 * n] = (float)(idx % width); corners[2 * n + 1] = (float)(idx / width);
            if (quality) quality[n] = unordF((unsigned)(c[i] >> 32));
            n++;
            if (maxCorners > 0 && n == maxCorners) break;
        }
    }
    (void)stg.finish("goodFeaturesToTrack");
    if (failed) return -2;
    return n;
}

} // extern "C"
